"""Link-prediction pre-training step (GNNEmb.py's inner loop) at a bench workload's graph shape: time per step and, under
rocprofv3 --kernel-trace --stats, the kernel table of the SSL path (EdgeGNN = EmbGConv(MyGCNConv) + pair mean pool + MLP).
    python tools/ssl_step.py [workload] [steps] [conv_layers] [dropout] [pairs] [eager|graph|program|program_eager]
"graph": the step as GNNEmb.py runs it (GraphedPairStep: forward + backward replayed from a hipGraph, optimizer eager).
The pairs are random node pairs (the reference's batch: 131072 edge / non-edge pairs, GNNEmb.py:144)."""
import functools
import os
import sys
import time

import numpy as np
import torch
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glass_amd import models, synth  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "ppi_bp"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    layers = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    dropout = float(sys.argv[4]) if len(sys.argv) > 4 else 0.5
    n_pairs = int(sys.argv[5]) if len(sys.argv) > 5 else 131072
    dev = "cuda:0"
    w, ei, ew, x, _pos, _y = synth.make_workload(name, seed=0, n_batches=1)
    rng = np.random.default_rng(0)
    pairs = torch.from_numpy(rng.integers(0, w.n_node, size=(n_pairs, 2))).to(dev)
    target = torch.from_numpy(rng.integers(0, 2, size=n_pairs).astype(np.float32)).to(dev)
    ei, ew, x = (torch.from_numpy(a).to(dev) for a in (ei, ew, x))
    torch.manual_seed(0)
    h = w.hidden
    conv = models.EmbGConv(h, h, h, layers, max_deg=int(x.max()), activation=nn.ReLU(inplace=True), jk=False, dropout=dropout,
                           conv=functools.partial(models.MyGCNConv, aggr=w.aggr), gn=True)
    head = models.MLP(h, h, 1, 2, dropout=dropout, activation=nn.ReLU(inplace=True))
    model = models.EdgeGNN(conv, nn.ModuleList([head]), nn.ModuleList([models.MeanPool()])).to(dev).train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    loss_fn = nn.BCEWithLogitsLoss()

    mode = sys.argv[6] if len(sys.argv) > 6 else "eager"
    if mode in ("program", "program_eager"):
        # the step as GNNEmb.py runs it now: parameters in one arena, FlatAdam, forward + backward as glass_amd.ssl.PairProgram
        import GNNEmb
        opt = GNNEmb.Pretrain.make_optimizer(model, 1e-3)
        from glass_amd import ssl
        prog = ssl.program_for(model)
        assert prog is not None, "step program not selected"
        if mode == "program_eager":
            os.environ["GLASS_SSL_GRAPH"] = "0"
        graphed = GNNEmb.GraphedPairStep(model, lambda pred, t: loss_fn(pred.flatten(), t), x, ei, ew, bce_mean=True)

        def step():
            loss = graphed(pairs, target)
            opt.step()
            return loss
    elif mode == "graph":
        import GNNEmb
        graphed = GNNEmb.GraphedPairStep(model, lambda pred, t: loss_fn(pred.flatten(), t), x, ei, ew)

        def step():
            loss = graphed(pairs, target)
            opt.step()
            return loss
    else:
        def step():
            opt.zero_grad()
            emb = model.NodeEmb(x, ei, ew)
            loss = loss_fn(model.preds[0](model.Pool(emb, pairs, None)).flatten(), target)
            loss.backward()
            opt.step()
            return loss

    for _ in range(5):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print(f"ssl_step {name}: N={w.n_node} H={h} layers={layers} dropout={dropout} pairs={n_pairs}: {dt * 1e3:.3f} ms/step ({mode}), "
          f"loss {loss.item():.5f}")


if __name__ == "__main__":
    main()
