import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch
from helpers import build_glass
from glass_amd import synth, ops
from glass_amd.arena import ParamArena
DEV="cuda:0"
w, ei, ew, x, pos, y = synth.make_workload("tiny", seed=6, n_batches=1)
ei, ew, x = (torch.from_numpy(a).to(DEV) for a in (ei, ew, x))
torch.manual_seed(1)
model = build_glass(64, w.layers, int(x.max()), w.n_class, w.aggr, w.pool, w.z_ratio, dropout=0.3).to(DEV).train()
ParamArena(model)
xb = x.flip(0)
x2 = torch.cat([x, xb], dim=1)
z = (torch.arange(x.shape[0], device=DEV) % 7 == 0).to(torch.int64)
ops.rng_seed(5, DEV)
with torch.enable_grad():
    e0 = model.conv(x2[:,0,:].reshape(-1,1), ei, ew, z)
    e1 = model.conv(x2[:,1,:].reshape(-1,1), ei, ew, z)
print("state after two", ops.rng_state(DEV).tolist())
ops.rng_seed(5, DEV)
a0 = model.conv(x.reshape(-1,1), ei, ew, z)
print("state after one", ops.rng_state(DEV).tolist())
ops.rng_seed(5, DEV); ops.rng_state(DEV)[1] = 1
a1 = model.conv(xb.reshape(-1,1), ei, ew, z)
print("fwd equal ch0", torch.equal(e0, a0), "ch1", torch.equal(e1, a1), float((e1-a1).abs().max()))
ops.rng_seed(5, DEV)
b0 = model.conv(x.reshape(-1,1), ei, ew, z)
print("repeat same call equal:", torch.equal(a0, b0))
ops.rng_seed(5, DEV)
c0 = model.conv(x2[:,0,:].reshape(-1,1), ei, ew, z)
print("x2 slice vs x:", torch.equal(c0, a0), torch.equal(x2[:,0,:].reshape(-1), x.reshape(-1)))
model.eval()
with torch.no_grad():
    d0 = model.conv(x.reshape(-1,1), ei, ew, z); d1 = model.conv(x2[:,0,:].reshape(-1,1), ei, ew, z)
print("eval equal:", torch.equal(d0, d1))
