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


def bench_dual():
    import torch.nn as nn
    print("dual linear (fused Linear pair + mix): fwd / dgrad")
    for N, H in [(17080, 64)]:
        for comb in (False, True):
            K = 2 * H if comb else H
            W = torch.randn(2 * H, K, device=DEV) / K**0.5
            b = torch.randn(2 * H, device=DEV)
            dW, db, WT = torch.zeros_like(W), torch.zeros_like(b), W.t().contiguous()
            lin1, lin0 = nn.Linear(K, H).to(DEV), nn.Linear(K, H).to(DEV)
            xa = torch.randn(N, H, device=DEV, requires_grad=True)
            xb = torch.randn(N, H, device=DEV, requires_grad=True) if comb else None
            mask = (torch.rand(N, device=DEV) < 0.05).to(torch.uint8)
            act = 0 if comb else 1
            stack = (W, b, dW, db, WT)
            t_f = timeit(lambda: ops.dual_linear_mix(xa.detach(), None if xb is None else xb.detach(), lin1, lin0, mask, 0.9, act, stack))
            out = ops.dual_linear_mix(xa, xb, lin1, lin0, mask, 0.9, act, stack)
            g = torch.randn_like(out)
            t_b = timeit(lambda: torch.autograd.grad(out, xa, g, retain_graph=True))
            print(f"  N={N} H={H} comb={comb}: fwd {t_f:.1f} us, bwd (dgrad+wgrad) {t_b:.1f} us")


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    if what in ("wgrad", "all"):
        bench_wgrad()
    if what in ("gn", "all"):
        bench_gn()
    if what in ("dual", "all"):
        bench_dual()
