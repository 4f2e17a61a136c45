"""Micro-benchmarks of single ops on the GPU (HIP events, interleaved A/B in one process).
usage: python tools/bench_ops.py [wgrad|gn|all]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from glass_amd import ops  # noqa: E402

DEV = "cuda:0"


def timeit(fn, iters=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3  # us


def bench_wgrad():
    print("wgrad: dW[O,I] = G^T X   (ours: split-K fp32 MFMA | lib: torch.mm -> hipBLASLt)")
    for N, O, I in [(4998, 128, 64), (17080, 128, 64), (17080, 128, 128), (14587, 128, 128), (50000, 256, 128),
                    (50000, 256, 256), (200000, 128, 128), (1000000, 128, 128), (1000000, 512, 256), (1000000, 512, 512)]:
        G = torch.randn(N, O, device=DEV)
        X = torch.randn(N, I, device=DEV)
        dW = torch.empty(O, I, device=DEV)
        db = torch.empty(O, device=DEV)
        t_ours = timeit(lambda: ops.linear_wgrad(G, X, dW, db, False))
        t_lib = timeit(lambda: (torch.mm(G.t(), X), G.sum(0)))
        fl = 2.0 * N * O * I
        print(f"  N={N:8d} O={O:4d} I={I:4d}  ours {t_ours:9.1f} us ({fl/t_ours/1e6:6.1f} TF)   lib {t_lib:9.1f} us "
              f"({fl/t_lib/1e6:6.1f} TF)")


def bench_gn():
    print("graphnorm fwd / bwd (3 kernels each)")
    for N, C in [(17080, 64), (17080, 128), (50000, 128), (1000000, 256), (1000000, 512)]:
        x = torch.randn(N, C, device=DEV, requires_grad=True)
        ones, zeros = torch.ones(C, device=DEV, requires_grad=True), torch.zeros(C, device=DEV, requires_grad=True)
        a = torch.ones(C, device=DEV, requires_grad=True)
        g = torch.randn(N, C, device=DEV)
        t_f = timeit(lambda: ops.graphnorm(x, ones, zeros, a))
        y = ops.graphnorm(x, ones, zeros, a)
        t_b = timeit(lambda: torch.autograd.grad(y, x, g, retain_graph=True))
        by = N * C * 4
        print(f"  N={N:8d} C={C:4d}  fwd {t_f:9.1f} us ({3*by/t_f/1e6:5.2f} TB/s)   bwd {t_b:9.1f} us ({5*by/t_b/1e6:5.2f} TB/s)")


def bench_dual(shapes=((17080, 64), (50000, 128), (1000000, 256))):
    """The fused Linear-pair kernels (forward with stats, data gradient, weight gradient) as the step program calls
    them, per pair, against the library GEMM of the same FLOPs."""
    from glass_amd import stack
    from glass_amd.arena import ParamArena
    from glass_amd.factory import build_glass
    print("fused Linear pair (glass_dual_linear_{fwd,dgrad,wgrad}_f32) vs library GEMM of the same shape")
    for N, H in shapes:
        model = build_glass(H, 1, 5, 3, "mean", "sum", 0.9).to(DEV).train()
        ParamArena(model)
        conv = model.conv.convs[0]
        # labeled fraction: 5 % by default (every 128-row tile holds a labeled row: the two-weight path of the comb kernels);
        # GLASS_LAB_LABELED=0.002 is config 5's density (most tiles take the effective-weight path)
        mask = (torch.rand(N, device=DEV) < float(os.environ.get("GLASS_LAB_LABELED", "0.05"))).to(torch.uint8)
        f32 = dict(dtype=torch.float32, device=DEV)
        h, a = torch.randn(N, H, **f32), torch.randn(N, H, **f32)
        T, m, c = torch.empty(N, 2 * H, **f32), torch.empty(N, H, **f32), torch.empty(N, H, **f32)
        nblk = -(-N // int(stack._lib.load().glass_dual_linear_stat_rows(H)))
        cstat = torch.empty(nblk, 2, H, dtype=torch.float64, device=DEV)
        dc, dm = torch.randn(N, H, **f32), torch.randn(N, H, **f32)
        din, dh = torch.empty(N, 2 * H, **f32), torch.empty(N, H, **f32)
        rows = []
        for kind, K, n_out in (("trans", H, H), ("comb", 2 * H, 2 * H)):
            st = conv._stack[kind]
            if kind == "trans":
                fwd = lambda: stack._dual_fwd(h, None, st, mask, 0.9, 1, T, m)
                dg = lambda: stack._dual_dgrad(dm, T, st, mask, 0.9, 1, H, None, dh)
                wg = lambda: stack._dual_wgrad(dm, T, st, mask, 0.9, 1, h, None, [])
            else:
                fwd = lambda: stack._dual_fwd(a, h, st, mask, 0.9, 0, None, c, cstat)
                dg = lambda: stack._dual_dgrad(dc, None, st, mask, 0.9, 0, 2 * H, None, din)
                wg = lambda: stack._dual_wgrad(dc, None, st, mask, 0.9, 0, a, h, [])
            A = torch.randn(N, K, **f32)
            W = torch.randn(2 * H, K, **f32)
            lib = lambda: torch.mm(A, W.t())
            fl = 2.0 * N * K * 2 * H
            iters = 5 if N >= 500000 else 30
            t = [timeit(f, iters=iters, warm=2) for f in (fwd, dg, wg, lib)]
            rows.append((kind, fl, t))
            print(f"  N={N:8d} H={H:4d} {kind:5s}: fwd {t[0]:9.1f} us ({fl/t[0]/1e6:6.1f} TF)  dgrad {t[1]:9.1f} us ({fl/t[1]/1e6:6.1f} TF)  "
                  f"wgrad {t[2]:9.1f} us ({fl/t[2]/1e6:6.1f} TF)  | library GEMM {t[3]:9.1f} us ({fl/t[3]/1e6:6.1f} TF)")
        del model


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    if what in ("wgrad", "all"):
        bench_wgrad()
    if what in ("gn", "all"):
        bench_gn()
    if what in ("dual", "all"):
        bench_dual()
    if what == "dualn":  # workgroup-count quantisation of the hidden-64 kernels: N around multiples of 256 CUs x 64 rows
        bench_dual(tuple((n, 64) for n in (8192, 12288, 16384, 16448, 17080, 20480, 24576, 32768, 32832)))
