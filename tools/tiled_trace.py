"""Phase timeline of the LDS-tiled forward (sel 7) / data-gradient (sel 8) kernels inside one training step (laboratory tool;
library built by tools/build_trace.sh, GLASS_HIP_LIB pointing at it).  Slots: 0 entry, 1 first stage published, 2 K loop done,
3 tile stored, 4 exit.  A kernel launched several times per step leaves the stamps of its LAST launch; <= 4096 workgroups.
usage: python tools/tiled_trace.py [workload]"""
import ctypes
import sys

import numpy as np
import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from glass_amd import _lib, losses, stack, synth  # noqa: E402
from glass_amd.arena import ParamArena  # noqa: E402
from glass_amd.factory import build_glass  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "em_user"
    dev = "cuda:0"
    w, ei, ew, x, pos, y = synth.make_workload(name, seed=0, n_batches=1)
    ei, ew, x, pos, y = (torch.from_numpy(a).to(dev) for a in (ei, ew, x, pos, y))
    torch.manual_seed(0)
    model = build_glass(w.hidden, w.layers, int(x.max()), w.n_class, w.aggr, w.pool, w.z_ratio, dropout=w.dropout)
    loss_fn = losses.BCEWithLogits() if w.multilabel else losses.CrossEntropy()
    model.to(dev).train()
    ParamArena(model)
    lib = _lib.load()
    lib.glass_tiled_trace_set.restype = ctypes.c_int
    lib.glass_tiled_trace_set.argtypes = [ctypes.c_void_p, ctypes.c_int]
    for _ in range(3):
        stack.loss_and_grads(model, loss_fn, x, ei, ew, pos, "pos", y, overwrite=True)
    torch.cuda.synchronize()
    buf = torch.zeros((4096, 4, 8), dtype=torch.int64, device=dev)
    for sel, label in ((7, "tiled fwd"), (8, "tiled dgrad")):
        buf.zero_()
        torch.cuda.synchronize()
        assert lib.glass_tiled_trace_set(buf.data_ptr(), sel) == 0
        stack.loss_and_grads(model, loss_fn, x, ei, ew, pos, "pos", y, overwrite=True)
        torch.cuda.synchronize()
        lib.glass_tiled_trace_set(None, 0)
        t = buf.cpu().numpy().astype(np.float64) * 10.0  # ns
        live = t[:, :, 0] > 0
        if not live.any():
            print(f"{label}: no stamps")
            continue
        t0 = t[:, :, 0][live].min()
        last = max(t[:, :, s][live & (t[:, :, s] > 0)].max() for s in (3, 4) if (live & (t[:, :, s] > 0)).any())
        print(f"== {label}: {int(live.sum())} waves in {int(live.any(axis=1).sum())} workgroups; span {last - t0:.0f} ns")
        st = t[:, :, 0][live] - t0
        print(f"   start  p10/p50/p90/max {np.percentile(st, 10):.0f}/{np.percentile(st, 50):.0f}/{np.percentile(st, 90):.0f}/{st.max():.0f}")
        prev = 0
        for slot, nm in ((1, "first stage"), (2, "K loop"), (3, "epilogue stores"), (4, "statistics")):
            ok = live & (t[:, :, slot] > 0)
            if not ok.any():
                continue
            d = (t[:, :, slot] - t[:, :, prev])[ok]
            print(f"   phase {nm:16s} ({prev}->{slot}) mean {d.mean():.0f}  p50 {np.percentile(d, 50):.0f}  p90 {np.percentile(d, 90):.0f}  max {d.max():.0f}")
            prev = slot
        if sel == 7:  # inside K step 2 of the forward: 5 top, 6 MFMAs issued, 7 next stage committed (before the barrier)
            for a, b, nm in ((5, 6, "step 2: loads issued + MFMAs"), (6, 7, "step 2: wait + cut + LDS writes")):
                ok = live & (t[:, :, a] > 0) & (t[:, :, b] > 0)
                if ok.any():
                    d = (t[:, :, b] - t[:, :, a])[ok]
                    print(f"   {nm:34s} mean {d.mean():.0f}  p50 {np.percentile(d, 50):.0f}  p90 {np.percentile(d, 90):.0f}")
        end = t[:, :, prev][live & (t[:, :, prev] > 0)] - t0
        print(f"   end    p10/p50/p90/max {np.percentile(end, 10):.0f}/{np.percentile(end, 50):.0f}/{np.percentile(end, 90):.0f}/{end.max():.0f}")
        # early starters against late starters: the second round of workgroups
        first_round = st < np.percentile(st, 50)
        life = (t[:, :, prev] - t[:, :, 0])[live]
        print(f"   life   early starters mean {life[first_round].mean():.0f}  late starters mean {life[~first_round].mean():.0f}")


if __name__ == "__main__":
    main()
