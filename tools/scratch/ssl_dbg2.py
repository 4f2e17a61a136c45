import functools, os, sys
import numpy as np, torch, torch.nn as nn
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from glass_amd import models, synth
import GNNEmb
dev = "cuda:0"
w, ei, ew, x, _p, _y = synth.make_workload("ppi_bp", seed=0, n_batches=1)
rng = np.random.default_rng(0)
ei, ew, x = (torch.from_numpy(a).to(dev) for a in (ei, ew, x))
n_pairs, dropout = 131072, 0.5
pairs = torch.from_numpy(rng.integers(0, w.n_node, size=(n_pairs, 2))).to(dev)
target = torch.from_numpy(rng.integers(0, 2, size=n_pairs).astype(np.float32)).to(dev)
for sync_every, use_opt in ((1, True), (0, True), (0, False)):
    torch.manual_seed(0)
    h = 64
    conv = models.EmbGConv(h, h, h, 2, max_deg=int(x.max()), activation=nn.ReLU(inplace=True), jk=False, dropout=dropout,
                           conv=functools.partial(models.MyGCNConv, aggr="mean"), gn=True)
    head = models.MLP(h, h, 1, 2, dropout=dropout, activation=nn.ReLU(inplace=True))
    model = models.EdgeGNN(conv, nn.ModuleList([head]), nn.ModuleList([models.MeanPool()])).to(dev).train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    lf = nn.BCEWithLogitsLoss()
    st = GNNEmb.GraphedPairStep(model, lambda pred, t: lf(pred.flatten(), t), x, ei, ew)
    out = []
    for k in range(60):
        loss = st(pairs, target)
        if use_opt:
            opt.step()
        if sync_every or k % 10 == 9:
            out.append(round(float(loss), 5))
    print("sync", sync_every, "opt", use_opt, out[-8:], flush=True)
    keep = [loss.clone() for _ in range(1)]
    torch.cuda.synchronize()
    print("  final", float(loss), "grad norm", float(sum(p.grad.double().pow(2).sum() for p in model.parameters()).sqrt()),
          "param finite", all(bool(torch.isfinite(p).all()) for p in model.parameters()))
