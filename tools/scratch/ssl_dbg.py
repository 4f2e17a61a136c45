import functools, os, sys
import numpy as np, torch, torch.nn as nn
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from glass_amd import models, synth
import GNNEmb
dev = "cuda:0"
w, ei, ew, x, _p, _y = synth.make_workload("ppi_bp", seed=0, n_batches=1)
rng = np.random.default_rng(0)
ei, ew, x = (torch.from_numpy(a).to(dev) for a in (ei, ew, x))
for n_pairs, dropout, optname in ((131072, 0.5, "adam"), (131072, 0.0, "adam"), (2048, 0.5, "adam"), (131072, 0.0, "sgd")):
    pairs = torch.from_numpy(rng.integers(0, w.n_node, size=(n_pairs, 2))).to(dev)
    target = torch.from_numpy(rng.integers(0, 2, size=n_pairs).astype(np.float32)).to(dev)
    for graph in (False, True):
        torch.manual_seed(0)
        h = 64
        conv = models.EmbGConv(h, h, h, 2, max_deg=int(x.max()), activation=nn.ReLU(inplace=True), jk=False, dropout=dropout,
                               conv=functools.partial(models.MyGCNConv, aggr="mean"), gn=True)
        head = models.MLP(h, h, 1, 2, dropout=dropout, activation=nn.ReLU(inplace=True))
        model = models.EdgeGNN(conv, nn.ModuleList([head]), nn.ModuleList([models.MeanPool()])).to(dev).train()
        opt = torch.optim.Adam(model.parameters(), lr=1e-3) if optname == "adam" else torch.optim.SGD(model.parameters(), lr=0.01)
        lf = nn.BCEWithLogitsLoss()
        st = GNNEmb.GraphedPairStep(model, lambda pred, t: lf(pred.flatten(), t), x, ei, ew)
        st.enabled = graph
        out = []
        for k in range(8):
            loss = st(pairs, target)
            gn = float(sum(p.grad.double().pow(2).sum() for p in model.parameters()).sqrt())
            out.append((round(float(loss), 5), round(gn, 5)))
            opt.step()
        print(n_pairs, dropout, optname, "graph" if graph else "eager", out, flush=True)
