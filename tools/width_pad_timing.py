"""Step time of a width outside the kernel families: zero-padded at the next family width (the step program) against the
same width on the per-op path with library GEMMs.  usage: python tools/width_pad_timing.py [hidden] [workload]"""
import sys
import time

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from glass_amd import losses, synth  # noqa: E402
from glass_amd.arena import ParamArena  # noqa: E402
from glass_amd.factory import build_glass  # noqa: E402
from glass_amd.optim import FlatAdam  # noqa: E402
from impl import utils  # noqa: E402


def main():
    hidden = int(sys.argv[1]) if len(sys.argv) > 1 else 96
    name = sys.argv[2] if len(sys.argv) > 2 else "ppi_bp"
    dev = "cuda:0"
    w, ei, ew, x, pos, y = synth.make_workload(name, seed=0, n_batches=1)
    ei, ew, x, pos, y = (torch.from_numpy(a).to(dev) for a in (ei, ew, x, pos, y))
    loss_fn = losses.BCEWithLogits() if w.multilabel else losses.CrossEntropy()
    for pad in (True, False):
        torch.manual_seed(0)
        model = build_glass(hidden, w.layers, int(x.max()), w.n_class, w.aggr, w.pool, w.z_ratio, dropout=w.dropout, pad_width=pad)
        model.to(dev).train()
        arena = ParamArena(model)
        opt = FlatAdam(arena, lr=1e-3)
        b_pos, b_y = (pos[0], y[0]) if pos.dim() == 3 else (pos, y)
        from glass_amd.step import TrainStep
        b_pos, b_y = (pos[0], y[0]) if pos.dim() == 3 else (pos, y)
        try:
            stepper = TrainStep(model, opt, loss_fn, x, ei, ew, arena, use_graph=True)
            for _ in range(5):
                stepper(b_pos, b_y)
            mode = "hipGraph replay"
        except Exception as e:  # the per-op path cannot always be captured: time it eagerly behind a GPU spin
            stepper = TrainStep(model, opt, loss_fn, x, ei, ew, arena, use_graph=False)
            for _ in range(5):
                stepper(b_pos, b_y)
            mode = f"eager ({type(e).__name__})"
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(100):
            stepper(b_pos, b_y)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 100
        lw = getattr(model, "_glass_logical_width", (hidden, hidden))
        print(f"{name} hidden {hidden} {'padded to ' + str(lw[1]) + ' (step program)' if pad else 'as it is (per-op path)'}: "
              f"{dt * 1e3:.3f} ms/step ({mode}, 100 steps)")


if __name__ == "__main__":
    main()
