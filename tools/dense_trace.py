"""Phase timeline of the hidden-64 dense kernels inside one training step (laboratory tool).

Needs a library built with -DGLASS_DENSE_TRACE (dense.hip stamps wall_clock64 — 100 MHz — per wave at the phase boundaries
of ONE selected kernel); run with GLASS_HIP_LIB pointing at it:
    GLASS_HIP_LIB=$PWD/tools/scratch/libglass_trace.so python tools/dense_trace.py [workload]
sel 1 comb forward (effective weight), 2 trans forward, 3 comb backward, 4 trans backward.  Forward slots: 0 entry,
1 GraphNorm coefficients ready, 2 product done, 3 tile stored, 4 statistics added; backward (data-gradient workgroups):
0 entry, 2 product done, 3 tile stored, 4 sums added; weight-gradient workgroups: 5 entry, 6 exit.
The step runs eagerly, so a kernel launched twice per step (two layers) leaves the stamps of its LAST launch."""
import ctypes
import sys

import numpy as np
import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from glass_amd import _lib, losses, stack, synth  # noqa: E402
from glass_amd.arena import ParamArena  # noqa: E402
from glass_amd.factory import build_glass  # noqa: E402


def pct(a, q):
    return float(np.percentile(a, q)) if len(a) else float("nan")


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "ppi_bp"
    dev = "cuda:0"
    w, ei, ew, x, pos, y = synth.make_workload(name, seed=0, n_batches=1)
    ei, ew, x, pos, y = (torch.from_numpy(a).to(dev) for a in (ei, ew, x, pos, y))
    torch.manual_seed(0)
    model = build_glass(w.hidden, w.layers, int(x.max()), w.n_class, w.aggr, w.pool, w.z_ratio, dropout=w.dropout)
    loss_fn = losses.BCEWithLogits() if w.multilabel else losses.CrossEntropy()
    model.to(dev).train()
    ParamArena(model)
    lib = _lib.load()
    lib.glass_dense_trace_set.restype = ctypes.c_int
    lib.glass_dense_trace_set.argtypes = [ctypes.c_void_p, ctypes.c_int]
    for _ in range(3):
        stack.loss_and_grads(model, loss_fn, x, ei, ew, pos, "pos", y, overwrite=True)
    torch.cuda.synchronize()
    buf = torch.zeros((4096, 4, 8), dtype=torch.int64, device=dev)
    for sel, label in ((1, "comb fwd"), (2, "trans fwd"), (3, "comb bwd"), (4, "trans bwd")):
        buf.zero_()
        torch.cuda.synchronize()
        assert lib.glass_dense_trace_set(buf.data_ptr(), sel) == 0
        stack.loss_and_grads(model, loss_fn, x, ei, ew, pos, "pos", y, overwrite=True)
        torch.cuda.synchronize()
        lib.glass_dense_trace_set(None, 0)
        t = buf.cpu().numpy().astype(np.float64) * 10.0  # ns
        live = t[:, :, 0] > 0
        wg = t[:, :, 5] > 0
        if not live.any():
            print(f"{label}: no stamps")
            continue
        t0 = min(t[:, :, 0][live].min(), t[:, :, 5][wg].min() if wg.any() else 1e30)
        ends = [t[:, :, 4][live].max()] + ([t[:, :, 6][wg].max()] if wg.any() else [])
        print(f"== {label}: {int(live.sum())} data waves, {int(wg.sum())} weight-gradient waves; span {max(ends) - t0:.0f} ns")
        st = t[:, :, 0][live] - t0
        print(f"   data waves start  p10/p50/p90/max {pct(st, 10):.0f}/{pct(st, 50):.0f}/{pct(st, 90):.0f}/{st.max():.0f}")
        prev = 0
        for slot, nm in ((1, "coef"), (2, "product"), (3, "stages"), (4, "sums")):
            ok = live & (t[:, :, slot] > 0)
            if not ok.any():
                continue
            d = (t[:, :, slot] - t[:, :, prev])[ok]
            print(f"   phase {nm:8s} (slot {prev}->{slot}) mean {d.mean():.0f}  p50 {pct(d, 50):.0f}  p90 {pct(d, 90):.0f}  max {d.max():.0f}")
            prev = slot
        if sel <= 2:  # forward kernels: inside the product — 5 first weight image committed, 6 first operand chunk ready,
            #               7 second chunk ready (after the first pass's MFMAs were issued)
            # (staged comb forward: 5 stage 1 begins, 6 its rows stored to LDS + next loads issued, 7 barrier passed, 2 MFMAs issued)
            for a, b, nm in ((0, 6, "0->6 setup"), (6, 1, "6->1 fold"), (1, 5, "1->5"), (5, 2, "5->2"), (2, 3, "2->3")):
                ok = live & (t[:, :, a] > 0) & (t[:, :, b] > 0)
                if ok.any():
                    d = (t[:, :, b] - t[:, :, a])[ok]
                    print(f"     inside: {nm:9s} ({a}->{b}) mean {d.mean():.0f}  p50 {pct(d, 50):.0f}  p90 {pct(d, 90):.0f}")
        life = (t[:, :, 4] - t[:, :, 0])[live]
        print(f"   data wave life    mean {life.mean():.0f}  p50 {pct(life, 50):.0f}  p90 {pct(life, 90):.0f}  max {life.max():.0f};"
              f" last end at {t[:, :, 4][live].max() - t0:.0f}")
        if wg.any():
            s5, l5 = t[:, :, 5][wg] - t0, (t[:, :, 6] - t[:, :, 5])[wg]
            print(f"   wgrad waves start p10/p50/p90/max {pct(s5, 10):.0f}/{pct(s5, 50):.0f}/{pct(s5, 90):.0f}/{s5.max():.0f};"
                  f" life mean {l5.mean():.0f} p90 {pct(l5, 90):.0f} max {l5.max():.0f}; last end at {t[:, :, 6][wg].max() - t0:.0f}")
            # split-form staged bodies: 5 entry, 1 first stage in LDS, 2 first stage multiplied, 3 last stage done, 7 tile stored, 6 exit
            prev_s = 5
            for slot, nm in ((1, "prologue"), (2, "stage 0"), (3, "other stages"), (7, "tile store"), (6, "bias + exit")):
                okw = wg & (t[:, :, slot] > 0) & (t[:, :, prev_s] > 0)
                if not okw.any():
                    continue
                dd = (t[:, :, slot] - t[:, :, prev_s])[okw]
                print(f"     wgrad {nm:12s} ({prev_s}->{slot}) mean {dd.mean():.0f}  p50 {pct(dd, 50):.0f}  p90 {pct(dd, 90):.0f}")
                prev_s = slot


if __name__ == "__main__":
    main()
