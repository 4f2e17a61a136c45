"""Where does the hidden-128 weight-gradient launch spend its time?  glass_dual_linear_wgrad_f32 (trans pair, ELU; the comb pair
in effective-weight form) timed alone at N = 12 500 .. 200 000 with the partials-only form (dW = NULL: no reduce launch):
the slab geometry keeps 64 slabs x 4 tiles = 256 workgroups, so rows per wave scale with N — slope = time per 16-row group,
intercept = ramp + tail."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from glass_amd import _lib, ops  # noqa: E402

DEV = "cuda:0"
if os.environ.get("GLASS_PROBE_LIB"):  # A/B against another build of the library (laboratory use)
    _lib.LIB_PATH = os.path.join(ROOT, os.environ["GLASS_PROBE_LIB"])
lib = _lib.load()
H = int(sys.argv[1]) if len(sys.argv) > 1 else 128
NS = tuple(int(v) for v in sys.argv[2].split(",")) if len(sys.argv) > 2 else (12500, 25000, 50000, 100000, 200000)
EAGER_ITERS = int(sys.argv[3]) if len(sys.argv) > 3 else 0  # under a counter collection: a few eager launches, no graph


def timeit(fn, iters=50, warm=10):
    if EAGER_ITERS:
        for _ in range(EAGER_ITERS):
            fn()
        torch.cuda.synchronize()
        return float("nan")
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0.record()
    g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for comb in (False, True):
    res = []
    for N in NS:
        gen = torch.Generator(device=DEV).manual_seed(N)
        dsrc = torch.randn(N, H, generator=gen, device=DEV)
        T = torch.randn(N, 2 * H, generator=gen, device=DEV)
        X = torch.randn(N, H, generator=gen, device=DEV)
        X2 = torch.randn(N, H, generator=gen, device=DEV) if comb else None
        mask = (torch.rand(N, generator=gen, device=DEV) < float(os.environ.get("PROBE_LABELED", "0.02"))).to(torch.uint8)
        I = 2 * H if comb else H
        ws = ops._wgrad_workspace(torch.device(DEV), N, 2 * H, I, slot=("probe", N, comb))
        act = 0 if comb else 1

        def run():
            rc = lib.glass_dual_linear_wgrad_f32(dsrc.data_ptr(), dsrc.stride(0), 0 if comb else T.data_ptr(), 0 if comb else T.stride(0),
                                                 mask.data_ptr(), 0.8, ops.act_word(act), X.data_ptr(), X.stride(0),
                                                 0 if X2 is None else X2.data_ptr(), 0 if X2 is None else X2.stride(0), N, H, 0, 0, 0, 1,
                                                 ws.data_ptr(), torch.cuda.current_stream().cuda_stream)
            assert rc == 0, lib.glass_last_error_string()
        res.append((N, timeit(run)))
    print("comb (effective-weight form)" if comb else "trans (ELU)", "hidden", H, " ".join(f"N={n}: {t:.1f} us" for n, t in res))
    for (n0, t0), (n1, t1) in zip(res, res[1:]):
        print(f"   {n0} -> {n1}: {(t1 - t0) / ((n1 - n0) / 64 / 4 / 16) :.2f} us per extra 16-row group per wave (64 slabs x 4 waves)")
