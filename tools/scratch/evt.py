import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from glass_amd import synth, ops, stack
from glass_amd.factory import build_glass
from glass_amd.arena import ParamArena
from glass_amd.optim import FlatAdam
from glass_amd.step import TrainStep
from glass_amd import losses
dev = "cuda:0"
nb = 16
w, ei, ew, x, pos, y = synth.make_workload("ppi_bp", seed=0, n_batches=nb)
ei, ew, x, pos, y = (torch.from_numpy(a) for a in (ei, ew, x, pos, y))
torch.manual_seed(0)
model = build_glass(w.hidden, w.layers, int(x.max()), w.n_class, w.aggr, w.pool, w.z_ratio, dropout=w.dropout).to(dev)
model.train()
bucket = ParamArena(model)
opt = FlatAdam(bucket, lr=w.lr)
loss_fn = losses.BCEWithLogits() if w.multilabel else losses.CrossEntropy()
xg, eig, ewg = x.to(dev), ei.to(dev), ew.to(dev)
pos_g = pos.to(dev).reshape(nb, w.batch, -1).contiguous()
y_g = y.to(dev).reshape(nb, w.batch, *y.shape[1:]).contiguous()
ops.rng_seed(1234, dev)
st = TrainStep(model, opt, loss_fn, xg, eig, ewg, bucket, use_graph=True)
for i in range(40):
    st(pos_g[i % nb], y_g[i % nb])
torch.cuda.synchronize()
side = torch.cuda.Stream()
main = torch.cuda.current_stream()
lab2 = stack.BatchLabels(xg.shape[0], pos_g[0].numel(), dev)
pos2 = torch.full_like(pos_g[0], -1)
dummy = torch.zeros(64, device=dev)

def loop(mode, n=400):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        b = i % nb
        if mode == 1:
            with torch.cuda.stream(side):
                dummy.add_(1.0)
                ev = torch.cuda.Event(); ev.record(side)
            main.wait_event(ev)
        elif mode == 2:
            ev0 = torch.cuda.Event(); ev0.record(main)
            side.wait_event(ev0)
            with torch.cuda.stream(side):
                lab2.load(pos_g[(b + 1) % nb].contiguous(), pos2)
                ev = torch.cuda.Event(); ev.record(side)
            main.wait_event(ev)
        st(pos_g[b], y_g[b])
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6

for rep in range(3):
    print("base %.2f us | side dummy+wait %.2f | side labels+wait %.2f" % (loop(0), loop(1), loop(2)))
