"""Randomised check of the REFERENCE CALLER on the captured step program: for random configurations — hidden width from every
kernel family (8 / 17 / 20 narrow, 64 staged, 128 tiled) and a padded-free off-family width that must fall back (48), depth,
aggregation, pooling, dropout 0, cross-entropy / the reference's binary loss function (single- and multi-label), ragged subgraph
matrices, batch sizes that do and do not divide the set — two epochs of `impl.train.train` with exactly what
/root/reference/GLASSTest.py constructs (buildModel, Adam(gnn.parameters(), lr), ReduceLROnPlateau, its loss callable,
ZGDataloader(shuffle, drop_last)) against an eager twin on the per-op path (plain autograd + torch.optim.Adam) from the same
seeds: epoch losses within 5e-4 relative, parameters within 5e-3 rel-inf, optimizer step counts equal, and the step really is
the captured program wherever the width has a kernel family.

usage (GPU box): python tools/fuzz_reference_caller.py [n_cases] [seed]"""
import copy
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from torch.nn import CrossEntropyLoss
from torch.optim import Adam, lr_scheduler

from helpers import rel_inf, record_parity
from test_gpu_reference_caller import reference_build_model, reference_loader, reference_binary_loss, _eager_epoch
from glass_amd import synth
from impl import SubGDataset, train, config

DEV = "cuda:0"
config.set_device(0)
N_CASES = int(sys.argv[1]) if len(sys.argv) > 1 else 16
SEED = int(sys.argv[2]) if len(sys.argv) > 2 else 2025
rng = np.random.default_rng(SEED)
rng_id = np.random.default_rng(SEED + 77)  # (its own stream: the configurations of earlier rounds stay what they were)
n_nodeid = 0
worst_loss, worst_par, n_prog = 0.0, 0.0, 0
for it in range(N_CASES):
    H = int(rng.choice([8, 17, 20, 64, 64, 128, 48]))
    L = int(rng.integers(1, 4))
    aggr = str(rng.choice(["mean", "sum", "gcn"]))
    pool = str(rng.choice(["sum", "mean", "size", "max"]))
    kind = str(rng.choice(["ce", "binary", "multilabel"]))
    K = int(rng.integers(2, 7)) if kind == "ce" else (1 if kind == "binary" else int(rng.integers(2, 6)))
    n = int(rng.integers(120, 2500))
    n_sub = int(rng.integers(24, 90))
    S = int(rng.integers(2, 20))
    bs = int(rng.integers(3, 17))
    V = int(rng.integers(3, 30))
    zr = float(rng.uniform(0.5, 1.0))
    lr = float(10 ** rng.uniform(-3.3, -2.0))
    # --use_nodeid (GLASSTest.py:152-157, datasets.py:58-61): x = arange(N) and an [N, hidden] table put in place of input_emb by
    # nn.Embedding.from_pretrained(emb, freeze=False); at hidden 64 / 128 large enough for the arena's big bucket (>= 1 MB)
    nodeid = bool(rng_id.random() < 0.3) or it == 1
    if nodeid and H in (64, 128):
        n = max(n, (1 << 18) // H + int(rng_id.integers(1, 400)))
    ei, ew = synth.make_graph(n, min(int(rng.integers(n, 5 * n)), n * (n - 1) // 4), it, float(rng.choice([0.0, 0.7])))
    x = torch.from_numpy(rng.integers(0, V, n)).reshape(n, 1, 1)
    if nodeid:
        x = torch.arange(n, dtype=torch.int64).reshape(n, 1, 1)
    pos = rng.integers(0, n, (n_sub, S))
    pos[rng.random((n_sub, S)) < 0.25] = -1
    pos[:, 0] = rng.integers(0, n, n_sub)
    y = (torch.from_numpy(rng.integers(0, K, n_sub)) if kind == "ce" else
         torch.from_numpy((rng.random(n_sub if kind == "binary" else (n_sub, K)) < 0.45).astype(np.float32)))
    xg, eig, ewg, posg, yg = (t.to(DEV) for t in (x, torch.from_numpy(ei), torch.from_numpy(ew), torch.from_numpy(pos), y))
    loss_fn = CrossEntropyLoss() if kind == "ce" else reference_binary_loss
    torch.manual_seed(it)
    gnn = reference_build_model(H, L, 0.0, True, pool, zr, aggr, torch.max(xg), K)
    if nodeid:
        gnn.conv.input_emb = torch.nn.Embedding.from_pretrained(torch.randn(n, H), freeze=False)
        gnn = gnn.to(config.device)
        n_nodeid += 1
    twin = copy.deepcopy(gnn)
    ds = SubGDataset.GDataset(xg, eig, ewg, posg, yg)
    opt, opt_twin = Adam(gnn.parameters(), lr=lr), Adam(twin.parameters(), lr=lr)
    scd = lr_scheduler.ReduceLROnPlateau(opt, factor=0.7, min_lr=5e-5)
    e_loss = 0.0
    for epoch in range(2):
        torch.manual_seed(1000 * it + epoch)
        got = train.train(opt, gnn, reference_loader(ds, bs), loss_fn)
        scd.step(got)
        torch.manual_seed(1000 * it + epoch)
        want = _eager_epoch(twin, opt_twin, reference_loader(ds, bs), loss_fn)
        e_loss = max(e_loss, abs(got - want) / max(abs(want), 1e-12))
    steps = gnn.__dict__.get("_glass_train_steps") or {}
    step = next(iter(steps.values())) if steps else None
    program = bool(step is not None and step.graphed and step._program_step())
    n_prog += program
    pa = torch.cat([p.detach().reshape(-1) for p in gnn.parameters()]).cpu()
    pb = torch.cat([p.detach().reshape(-1) for p in twin.parameters()]).cpu()
    e_par = rel_inf(pa, pb)
    n_steps = 2 * (n_sub // bs)
    counts = {int(float(s["step"])) for s in opt.state_dict()["state"].values()}
    ok = e_loss < 5e-4 and e_par < 5e-3 and counts == {n_steps} and step is not None and step.graphed
    # widths with a kernel family and a fusable pool must be on the program; max pooling and width 48 (no padding asked for
    # through the reference's own buildModel) legitimately take the captured per-op step
    expect_program = H in (8, 17, 20, 64, 128) and pool != "max"
    ok = ok and (program == expect_program)
    worst_loss, worst_par = max(worst_loss, e_loss), max(worst_par, e_par)
    print(f"{it:2d} H={H:3d} L={L} {aggr:4s} {pool:4s} {kind:10s} K={K} n={n} subs={n_sub}x{S} bs={bs} lr={lr:.1e}{' nodeid' if nodeid else ''}: loss {e_loss:.1e} params "
          f"{e_par:.1e} steps {sorted(counts)} program={program}{'' if ok else '  <-- FAIL'}", flush=True)
    if not ok:
        sys.exit(1)
print(f"worst: epoch loss {worst_loss:.2e}, parameters {worst_par:.2e}; {n_prog} of {N_CASES} cases on the captured step program, "
      f"{n_nodeid} with a from_pretrained node-id table")
record_parity(f"fuzz/reference_caller_{N_CASES}_random_configs_seed{SEED}", cases=N_CASES, worst_epoch_loss_rel=worst_loss,
              worst_param_rel_inf=worst_par, on_step_program=n_prog, nodeid_cases=n_nodeid,
              note="tools/fuzz_reference_caller.py: GLASSTest.py's own objects through impl.train.train vs an eager per-op twin, two epochs")
