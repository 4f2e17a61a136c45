"""The REFERENCE'S OWN CALLER on the measured path.

Everything below is constructed the way /root/reference/GLASSTest.py constructs it — `buildModel` (:129-175), the loaders of
`split()` (:104-126), `Adam(gnn.parameters(), lr=lr)` + `ReduceLROnPlateau(optimizer, factor=resi, min_lr=5e-5)` (:213-216),
the binary loss as a plain function around `BCEWithLogitsLoss` (:57-58), `CrossEntropyLoss()` (:69) — through the `impl.*`
module surface only: no ParamArena, no FlatAdam, no glass_amd.losses class.  `impl.train.train` must put that caller on
the hipGraph step program by itself (optimizer adopted in place, loss recognised by evaluation) and keep every torch-side
contract: scheduler cuts are followed, `optimizer.state_dict()` carries the moments and the step count, eager use of the
same optimizer afterwards continues the same state."""
import copy
import functools

import numpy as np
import pytest
import torch
import torch.nn as nn
from torch.nn import BCEWithLogitsLoss, CrossEntropyLoss
from torch.optim import Adam, lr_scheduler

from helpers import load, sd_from, rel_inf, flat_grads

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def reference_build_model(hidden_dim, conv_layer, dropout, jk, pool, z_ratio, aggr, max_deg, output_channels):
    """GLASSTest.py:129-175, line by line in meaning (module globals `max_deg`, `output_channels` as arguments)."""
    from impl import models, config
    conv = models.EmbZGConv(hidden_dim, hidden_dim, conv_layer, max_deg=max_deg, activation=nn.ELU(inplace=True), jk=jk,
                            dropout=dropout,
                            conv=functools.partial(models.GLASSConv, aggr=aggr, z_ratio=z_ratio, dropout=dropout), gn=True)
    mlp = nn.Linear(hidden_dim * (conv_layer) if jk else hidden_dim, output_channels)
    pool_fn_fn = {"mean": models.MeanPool, "max": models.MaxPool, "sum": models.AddPool, "size": models.SizePool}
    pool_fn1 = pool_fn_fn[pool]()
    gnn = models.GLASS(conv, torch.nn.ModuleList([mlp]), torch.nn.ModuleList([pool_fn1])).to(config.device)
    return gnn


def reference_loader(ds, bs, shuffle=True, drop_last=True):
    """GLASSTest.py:107-113 (`tfunc`)."""
    from impl import SubGDataset, utils
    return SubGDataset.ZGDataloader(ds, bs, z_fn=utils.MaxZOZ, shuffle=shuffle, drop_last=drop_last)


def reference_binary_loss(x, y):
    """GLASSTest.py:57-58."""
    return BCEWithLogitsLoss()(x.flatten(), y.flatten())


def _taken_step(model):
    steps = model.__dict__.get("_glass_train_steps") or {}
    assert len(steps) == 1, "impl.train.train did not build a TrainStep for the reference caller"
    return next(iter(steps.values()))


@pytest.fixture(autouse=True)
def _device():
    from impl import config
    config.set_device(0)


def _g8():
    g = load("g8_adam.npz")
    x = torch.from_numpy(g["x"]).to(DEV)
    ei, ew = torch.from_numpy(g["edge_index"]).to(DEV), torch.from_numpy(g["edge_weight"]).to(DEV)
    pos, y = torch.from_numpy(g["pos"]).to(DEV), torch.from_numpy(g["y"]).to(DEV)
    return g, x, ei, ew, pos, y


def test_reference_caller_reproduces_g8_on_the_graph_path():
    """g8 = the reference's own three Adam steps (losses + final weights).  Three one-batch epochs of impl.train.train with
    the reference's constructions reproduce them, on the captured step program."""
    from impl import SubGDataset, train
    from glass_amd import stack
    g, x, ei, ew, pos, y = _g8()
    gnn = reference_build_model(int(g["hidden"]), int(g["layers"]), 0.0, True, str(g["pool"]), float(g["z_ratio"]), str(g["aggr"]),
                                torch.max(x), 3)
    gnn.load_state_dict(sd_from(g))
    optimizer = Adam(gnn.parameters(), lr=float(g["lr"]))
    scd = lr_scheduler.ReduceLROnPlateau(optimizer, factor=0.7, min_lr=5e-5)
    loss_fn = CrossEntropyLoss()
    params_before = [p for p in gnn.parameters()]
    got = []
    for k in range(3):
        ds = SubGDataset.GDataset(x, ei, ew, pos[4 * k:4 * k + 4], y[4 * k:4 * k + 4])
        loss = train.train(optimizer, gnn, reference_loader(ds, 4, shuffle=False), loss_fn)
        scd.step(loss)
        got.append(loss)
    step = _taken_step(gnn)
    assert step.graphed and step._program_step() and stack.step_supported(gnn, loss_fn), "not the captured step program"
    assert all(a is b for a, b in zip(params_before, gnn.parameters())), "Parameter identities must survive the adoption"
    assert np.allclose(got, g["losses"], rtol=1e-5, atol=0), (got, g["losses"])
    end = sd_from(g, "sd_end/")
    keys = sorted(end)
    mine = {k: v.cpu() for k, v in gnn.state_dict().items()}
    assert rel_inf(flat_grads(mine, keys), flat_grads(end, keys)) < 1e-4
    # the torch optimizer still tells the truth about its state
    sd = optimizer.state_dict()
    assert len(sd["state"]) == len(params_before)
    assert all(float(s["step"]) == 3.0 for s in sd["state"].values())
    assert all(float(s["exp_avg_sq"].abs().sum()) > 0 for s in sd["state"].values() if s["exp_avg_sq"].numel() > 8)


def _binary_task(seed=0, n=300, n_sub=24, smax=6, k_out=1):
    gen = torch.Generator().manual_seed(seed)
    pairs = torch.randint(0, n, (2, 900), generator=gen)
    pairs = pairs[:, pairs[0] != pairs[1]]
    ei = torch.cat([pairs, pairs.flip(0)], dim=1)
    ei = torch.unique(ei, dim=1)
    ew = torch.ones(ei.shape[1])
    deg = torch.bincount(ei[0], minlength=n)
    x = torch.unique(deg, return_inverse=True)[1].reshape(n, 1, 1)
    pos = torch.stack([torch.randperm(n, generator=gen)[:smax] for _ in range(n_sub)])
    pos[::3, -2:] = -1
    y = (torch.rand(n_sub, generator=gen) > 0.5).float() if k_out == 1 else (torch.rand(n_sub, k_out, generator=gen) > 0.5).float()
    return tuple(t.to(DEV) for t in (x, ei, ew, pos, y))


def _eager_epoch(model, opt, loader, loss_fn):
    """The reference's train() body (impl/train.py:4-17) on the per-op path: what the caller got before adoption."""
    model.train()
    out = []
    for batch in loader:
        opt.zero_grad()
        loss = loss_fn(model(*batch[:-1], id=0), batch[-1])
        loss.backward()
        out.append(loss.item())
        opt.step()
    return float(np.mean(out))


@pytest.mark.parametrize("k_out", [1, 3])
def test_reference_binary_lambda_is_fused_and_scheduler_cut_is_followed(k_out):
    """Binary / multi-label sets: the reference's loss is an opaque function (GLASSTest.py:57-58).  It is recognised by
    evaluation, the caller lands on the step program; losses follow an eager twin (same seeds); a ReduceLROnPlateau cut
    is followed by the very next replay: the parameter update shrinks with the learning rate."""
    from impl import SubGDataset, train
    from glass_amd import losses
    x, ei, ew, pos, y = _binary_task(k_out=k_out)
    assert losses.fusable_mode(reference_binary_loss) == 1
    torch.manual_seed(3)
    gnn = reference_build_model(64, 2, 0.0, True, "sum", 0.9, "mean", torch.max(x), k_out)
    twin = copy.deepcopy(gnn)
    ds = SubGDataset.GDataset(x, ei, ew, pos, y)
    optimizer = Adam(gnn.parameters(), lr=1e-2)
    scd = lr_scheduler.ReduceLROnPlateau(optimizer, factor=0.1, patience=0, min_lr=5e-5)
    opt_twin = Adam(twin.parameters(), lr=1e-2)
    for epoch in range(2):
        torch.manual_seed(100 + epoch)
        got = train.train(optimizer, gnn, reference_loader(ds, 8), reference_binary_loss)
        torch.manual_seed(100 + epoch)
        want = _eager_epoch(twin, opt_twin, reference_loader(ds, 8), reference_binary_loss)
        assert abs(got - want) <= 2e-4 * abs(want), (epoch, got, want)
    step = _taken_step(gnn)
    assert step.graphed and step._program_step()
    # force a cut: a "loss" far above the best one
    scd.step(1.0)
    scd.step(1e9)
    lr_new = optimizer.param_groups[0]["lr"]
    assert lr_new == pytest.approx(1e-3)
    before = torch.cat([p.detach().reshape(-1).clone() for p in gnn.parameters()])
    one = SubGDataset.GDataset(x, ei, ew, pos[:8], y[:8])
    train.train(optimizer, gnn, reference_loader(one, 8), reference_binary_loss)
    assert _taken_step(gnn) is step, "a learning-rate change must not need a new capture"
    delta = (torch.cat([p.detach().reshape(-1) for p in gnn.parameters()]) - before).abs().max().item()
    assert 0.05 * lr_new < delta <= 1.5 * lr_new, f"update {delta:.3e} does not follow the cut learning rate {lr_new:.1e}"
    assert float(step.opt.lr_dev) == pytest.approx(lr_new)


def test_adopted_optimizer_state_round_trips_and_eager_use_continues():
    """optimizer.state_dict() -> a fresh Adam on a fresh model -> training continues exactly as without the round trip; and
    an eager step of the adopted optimizer in between (zero_grad(set_to_none=True), backward, step) is part of the same
    trajectory: the next train() call picks its step count up."""
    from impl import SubGDataset, train
    x, ei, ew, pos, y = _binary_task(seed=1)
    torch.manual_seed(5)
    a = reference_build_model(64, 2, 0.0, True, "sum", 0.9, "mean", torch.max(x), 1)
    ds = SubGDataset.GDataset(x, ei, ew, pos, y)
    opt_a = Adam(a.parameters(), lr=5e-3)
    torch.manual_seed(7)
    train.train(opt_a, a, reference_loader(ds, 8), reference_binary_loss)
    # checkpoint
    b = reference_build_model(64, 2, 0.0, True, "sum", 0.9, "mean", torch.max(x), 1)
    b.load_state_dict(copy.deepcopy(a.state_dict()))
    opt_b = Adam(b.parameters(), lr=5e-3)
    opt_b.load_state_dict(copy.deepcopy(opt_a.state_dict()))
    assert all(float(s["step"]) == 3.0 for s in opt_b.state_dict()["state"].values())
    for model, opt in ((a, opt_a), (b, opt_b)):
        torch.manual_seed(8)
        train.train(opt, model, reference_loader(ds, 8), reference_binary_loss)
    pa = torch.cat([p.detach().reshape(-1) for p in a.parameters()])
    pb = torch.cat([p.detach().reshape(-1) for p in b.parameters()])
    assert torch.equal(pa, pb), f"resumed run diverged: {(pa - pb).abs().max().item():.3e}"
    assert all(float(s["step"]) == 6.0 for s in opt_b.state_dict()["state"].values())
    # eager use of the adopted optimizer, reference style
    from impl import utils
    opt_a.zero_grad()
    p8 = pos[:8]
    loss = reference_binary_loss(a(x, ei, ew, p8, utils.MaxZOZ(x, p8), id=0), y[:8])
    loss.backward()
    opt_a.step()
    assert all(float(s["step"]) == 7.0 for s in opt_a.state_dict()["state"].values())
    torch.manual_seed(9)
    train.train(opt_a, a, reference_loader(ds, 8), reference_binary_loss)
    assert all(float(s["step"]) == 10.0 for s in opt_a.state_dict()["state"].values())
    assert int(_taken_step(a).opt.step_dev[0]) == 10


def test_unadoptable_callers_keep_the_eager_loop():
    """amsgrad / another optimizer class / an unknown loss: impl.train.train stays on the plain per-batch loop."""
    from impl import SubGDataset, train
    x, ei, ew, pos, y = _binary_task(seed=2)
    ds = SubGDataset.GDataset(x, ei, ew, pos, y)
    torch.manual_seed(1)
    m = reference_build_model(64, 1, 0.0, True, "mean", 0.8, "gcn", torch.max(x), 1)
    for opt in (Adam(m.parameters(), lr=1e-3, amsgrad=True), torch.optim.SGD(m.parameters(), lr=1e-3)):
        loss = train.train(opt, m, reference_loader(ds, 8), reference_binary_loss)
        assert np.isfinite(loss) and not m.__dict__.get("_glass_train_steps")
    # an unknown loss with an adoptable optimizer: the step is built, but not as the fused program
    opt = Adam(m.parameters(), lr=1e-3)
    loss = train.train(opt, m, reference_loader(ds, 8), lambda p, t: ((p.flatten() - t.flatten()) ** 2).mean())
    assert np.isfinite(loss)
    assert not _taken_step(m)._program_step()


def test_adopted_adam_follows_weight_decay_and_a_beta_change():
    """Hyper-parameters other than the learning rate are launch arguments of the fused update: the adopted engine reads them
    from optimizer.param_groups — weight_decay (torch.optim.Adam's L2 form) from the start, and a later change of `betas`
    triggers a new capture — so the trajectory keeps following an eager twin driven by the same torch optimizer settings."""
    from impl import SubGDataset, train
    x, ei, ew, pos, y = _binary_task(seed=4)
    torch.manual_seed(11)
    gnn = reference_build_model(64, 2, 0.0, True, "sum", 0.9, "mean", torch.max(x), 1)
    twin = copy.deepcopy(gnn)
    ds = SubGDataset.GDataset(x, ei, ew, pos, y)
    opt = Adam(gnn.parameters(), lr=5e-3, weight_decay=1e-2)
    opt_twin = Adam(twin.parameters(), lr=5e-3, weight_decay=1e-2)
    for epoch in range(3):
        if epoch == 2:
            for o in (opt, opt_twin):
                o.param_groups[0]["betas"] = (0.8, 0.99)
        torch.manual_seed(200 + epoch)
        got = train.train(opt, gnn, reference_loader(ds, 8), reference_binary_loss)
        torch.manual_seed(200 + epoch)
        want = _eager_epoch(twin, opt_twin, reference_loader(ds, 8), reference_binary_loss)
        assert abs(got - want) <= 2e-4 * abs(want), (epoch, got, want)
    step = _taken_step(gnn)
    assert step.graphed and step._hyper == (0.8, 0.99, 1e-8, 1e-2)
    pa = torch.cat([p.detach().reshape(-1) for p in gnn.parameters()])
    pb = torch.cat([p.detach().reshape(-1) for p in twin.parameters()])
    assert rel_inf(pa.cpu(), pb.cpu()) < 2e-3
    # weight decay really acted: the parameters of a run without it differ by far more than the twin does
    torch.manual_seed(11)
    free = reference_build_model(64, 2, 0.0, True, "sum", 0.9, "mean", torch.max(x), 1)
    opt_free = Adam(free.parameters(), lr=5e-3)
    for epoch in range(3):
        torch.manual_seed(200 + epoch)
        train.train(opt_free, free, reference_loader(ds, 8), reference_binary_loss)
    pf = torch.cat([p.detach().reshape(-1) for p in free.parameters()])
    assert rel_inf(pa.cpu(), pf.cpu()) > 10 * rel_inf(pa.cpu(), pb.cpu())


def test_reference_caller_randomised_configurations():
    """tools/fuzz_reference_caller.py: 12 random configurations — every kernel family's widths (8 / 17 / 20, 64, 128) and one
    without a family (48), 1-3 layers, all aggregations and pools (max pooling included), cross-entropy and the reference's
    binary / multi-label loss function, ragged subgraphs, batch sizes that do not divide the set — of GLASSTest.py's own
    objects through impl.train.train against an eager per-op twin: epoch losses, parameters, optimizer step counts, and the
    captured step program wherever a family serves the width."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_reference_caller.py"), "12", "7"],
                         capture_output=True, text=True, timeout=900, cwd=root)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert "on the captured step program" in out.stdout


def test_deepcopy_after_training_gives_an_independent_plain_model():
    """A caller that snapshots its best model with copy.deepcopy AFTER impl.train.train has adopted the optimizer (arena,
    captured step, evaluation graphs on the model): the copy is a plain model with the same weights, trains on its own —
    landing on the step program again — and training it leaves the original untouched (a value copy of the arena would have
    kept the original's device pointers in its launch arguments)."""
    from impl import SubGDataset, train, metrics
    from glass_amd import arena
    x, ei, ew, pos, y = _binary_task(seed=6)
    torch.manual_seed(2)
    gnn = reference_build_model(64, 2, 0.0, True, "sum", 0.9, "mean", torch.max(x), 1)
    ds = SubGDataset.GDataset(x, ei, ew, pos, y)
    opt = Adam(gnn.parameters(), lr=5e-3)
    torch.manual_seed(1)
    train.train(opt, gnn, reference_loader(ds, 8), reference_binary_loss)
    train.test(gnn, reference_loader(ds, 8, drop_last=False), metrics.binaryf1, reference_binary_loss)
    assert gnn.__dict__.get("_glass_train_steps") and gnn.__dict__.get("_glass_grad_bucket") is not None
    snap = copy.deepcopy(gnn)
    assert not any(k in m.__dict__ for m in snap.modules() for k in arena._RUNTIME_ATTRS)
    before = {k: v.clone() for k, v in gnn.state_dict().items()}
    assert all(torch.equal(v, snap.state_dict()[k]) for k, v in before.items())
    opt2 = Adam(snap.parameters(), lr=5e-3)
    torch.manual_seed(2)
    train.train(opt2, snap, reference_loader(ds, 8), reference_binary_loss)
    assert _taken_step(snap)._program_step() and _taken_step(snap) is not _taken_step(gnn)
    assert all(torch.equal(v, gnn.state_dict()[k]) for k, v in before.items()), "training the copy changed the original"
    assert any(not torch.equal(v, snap.state_dict()[k]) for k, v in before.items())
    # and the original keeps training on its own graph
    torch.manual_seed(3)
    assert np.isfinite(train.train(opt, gnn, reference_loader(ds, 8), reference_binary_loss))


def _nodeid_task(n, seed=0, n_sub=24, smax=7, n_class=3):
    """A `--use_nodeid` data set as /root/reference/datasets.py:58-61 makes it: x = arange(N) (V = N), unit edge weights."""
    gen = torch.Generator().manual_seed(seed)
    pairs = torch.randint(0, n, (2, 6 * n), generator=gen)
    pairs = pairs[:, pairs[0] != pairs[1]]
    ei = torch.unique(torch.cat([pairs, pairs.flip(0)], dim=1), dim=1)
    ew = torch.ones(ei.shape[1])
    x = torch.arange(n, dtype=torch.int64).reshape(n, 1, -1)
    pos = torch.stack([torch.randperm(n, generator=gen)[:smax] for _ in range(n_sub)])
    pos[::4, -2:] = -1
    pos[1, 0] = pos[0, 0]                    # a node shared by two subgraphs of one batch
    pos[:, 1] = n - 1 - torch.arange(n_sub)  # the last table rows are named too
    y = torch.randint(0, n_class, (n_sub,), generator=gen)
    return x, ei, ew, pos, y


@pytest.mark.parametrize("n", [5000, 9000])
def test_reference_caller_use_nodeid_from_pretrained(n):
    """The README recipe for the real-world sets (/root/reference/GLASSTest.py:152-157): `buildModel` re-assigns
    `conv.input_emb = nn.Embedding.from_pretrained(emb, freeze=False)` — an [N, hidden] table, V = N — on a graph whose
    features are the node ids.  Through impl.train.train with the reference's own objects: the caller lands on the captured
    step program (adoption builds the arena with the table in the big bucket: 5 000 x 64 floats >= 1 MB; 5 000 rows run the
    lookup + emb_gn through the table kernels, 9 000 rows — beyond 8 192 — the [N, H] kernels), three one-batch epochs
    equal an eager twin, the first step's gradients equal the fp64 oracle's on the flat vector INCLUDING the table's rows,
    and optimizer.state_dict() carries the table's moments."""
    from impl import SubGDataset, train, utils
    from glass_amd import stack
    from oracle import glass_oracle as O
    H, L, n_class, bs = 64, 2, 3, 8
    x, ei, ew, pos, y = _nodeid_task(n, seed=n)
    torch.manual_seed(n)
    emb = torch.randn(n, H)                  # (stands for ./Emb/<dataset>_64.pt, the pre-training path's output)
    gnn = reference_build_model(H, L, 0.0, True, "sum", 0.9, "mean", torch.max(x), n_class)
    gnn.conv.input_emb = nn.Embedding.from_pretrained(emb.clone(), freeze=False)   # GLASSTest.py:157
    from impl import config
    gnn = gnn.to(config.device)
    assert gnn.conv.input_emb.weight.requires_grad and tuple(gnn.conv.input_emb.weight.shape) == (n, H)
    sd0 = {k: v.detach().cpu().clone() for k, v in gnn.state_dict().items()}
    twin = copy.deepcopy(gnn)
    xg, eig, ewg, posg, yg = (t.to(DEV) for t in (x, ei, ew, pos, y))
    optimizer = Adam(gnn.parameters(), lr=1e-3)
    opt_twin = Adam(twin.parameters(), lr=1e-3)
    loss_fn = CrossEntropyLoss()
    table_param = gnn.conv.input_emb.weight
    got, want = [], []
    for k in range(3):
        ds = SubGDataset.GDataset(xg, eig, ewg, posg[bs * k:bs * k + bs], yg[bs * k:bs * k + bs])
        got.append(train.train(optimizer, gnn, reference_loader(ds, bs, shuffle=False), loss_fn))
        if k == 0:  # the gradient buffers still hold step 0's gradients (they are rewritten, not accumulated, by the next step)
            grads0 = {kk: p.grad.detach().cpu().clone() for kk, p in gnn.named_parameters()}
        want.append(_eager_epoch(twin, opt_twin, reference_loader(ds, bs, shuffle=False), loss_fn))
    step = _taken_step(gnn)
    assert step.graphed and step._program_step() and stack.step_supported(gnn, loss_fn), "not the captured step program"
    assert gnn.conv.input_emb.weight is table_param, "the table's Parameter identity must survive the adoption"
    arena = gnn.__dict__["_glass_grad_bucket"]
    assert arena.big_start < arena.flat.numel() and any(p is table_param for p in arena.params), "table not in the big bucket"
    assert np.allclose(got, want, rtol=2e-5, atol=0), (got, want)
    pa = torch.cat([p.detach().reshape(-1) for p in gnn.parameters()]).cpu()
    pb = torch.cat([p.detach().reshape(-1) for p in twin.parameters()]).cpu()
    assert rel_inf(pa, pb) < 1e-4
    # step 0 against the fp64 oracle: loss and the flat gradient vector, table included
    orc = O.OracleGLASS(H, L, n - 1, n_class, aggr="mean", pool="sum", z_ratio=0.9).double()
    orc.load_state_dict({k: v.double() for k, v in sd0.items()})
    orc.train()
    p0 = pos[:bs]
    lo = nn.CrossEntropyLoss()(orc(x, ei, ew.double(), p0, O.max_zero_one(x, p0)), y[:bs])
    lo.backward()
    assert abs(got[0] - lo.item()) <= 1e-5 * abs(lo.item()), (got[0], lo.item())
    keys = sorted(k for k, _ in orc.named_parameters())
    assert "conv.input_emb.weight" in keys
    ref = {k: p.grad for k, p in orc.named_parameters()}
    e_all = rel_inf(flat_grads(grads0, keys), flat_grads(ref, keys))
    e_tab = rel_inf(grads0["conv.input_emb.weight"], ref["conv.input_emb.weight"])
    from helpers import record_parity
    record_parity(f"reference_caller/use_nodeid_N{n}", loss_rel=abs(got[0] - lo.item()) / abs(lo.item()), grad_rel_inf=e_all,
                  table_grad_rel_inf=e_tab)
    assert e_all < 1e-5 and e_tab < 1e-5, (e_all, e_tab)
    # the torch optimizer tells the truth about the table's state
    sd = optimizer.state_dict()
    idx = [i for i, p in enumerate(optimizer.param_groups[0]["params"]) if p is table_param][0]
    st = sd["state"][sd["param_groups"][0]["params"][idx]]
    assert float(st["step"]) == 3.0 and tuple(st["exp_avg"].shape) == (n, H)
    tw = opt_twin.state_dict()["state"][opt_twin.state_dict()["param_groups"][0]["params"][idx]]
    assert rel_inf(st["exp_avg"].cpu(), tw["exp_avg"].cpu()) < 1e-4 and rel_inf(st["exp_avg_sq"].cpu(), tw["exp_avg_sq"].cpu()) < 1e-4
    assert int((st["exp_avg"].abs().sum(1) > 0).sum()) > 100, "the table's first moment is empty"


def test_evaluation_before_the_first_training_epoch_is_not_replayed_on_stale_parameters():
    """test() -> train() -> test() (an epoch-0 baseline, or a resumed checkpoint evaluated first): the first test() caches its
    evaluation hipGraph on the model's per-parameter storage; the first train() adopts the optimizer and moves every
    parameter into the flat arena.  The cached graph holds the OLD addresses — it must be dropped (arena.drop_captured_graphs),
    not replayed: the second test() equals eager forwards of the trained model."""
    from impl import SubGDataset, train, metrics, utils
    x, ei, ew, pos, y = _binary_task(seed=8, n_sub=32)
    torch.manual_seed(4)
    gnn = reference_build_model(64, 2, 0.0, True, "sum", 0.9, "mean", torch.max(x), 1)
    ds = SubGDataset.GDataset(x, ei, ew, pos, y)
    opt = Adam(gnn.parameters(), lr=2e-2)

    def eager_score():
        gnn.eval()
        with torch.no_grad():
            pred = torch.cat([gnn(x, ei, ew, pos[i:i + 8], utils.MaxZOZ(x, pos[i:i + 8])) for i in range(0, pos.shape[0], 8)])
        return pred

    s0, l0 = train.test(gnn, reference_loader(ds, 8, shuffle=False, drop_last=False), metrics.binaryf1, reference_binary_loss)
    assert gnn.__dict__.get("_glass_eval_graphs"), "the evaluation did not run from a cached graph: the case is not exercised"
    assert abs(float(l0) - float(reference_binary_loss(eager_score(), y))) < 1e-6
    for epoch in range(3):
        torch.manual_seed(30 + epoch)
        train.train(opt, gnn, reference_loader(ds, 8), reference_binary_loss)
    assert _taken_step(gnn)._program_step()
    s1, l1 = train.test(gnn, reference_loader(ds, 8, shuffle=False, drop_last=False), metrics.binaryf1, reference_binary_loss)
    want = float(reference_binary_loss(eager_score(), y))
    assert abs(float(l1) - want) < 1e-6 * max(1.0, abs(want)), (float(l1), want, float(l0))
    assert abs(float(l1) - float(l0)) > 1e-3, "training did not move the loss: the check above proves nothing"


_SWITCHES = [("glass_amd.train", "USE_STEP", False), ("glass_amd.train", "USE_GRAPH", False), ("glass_amd.train", "USE_HEAD_LABELS", False),
             ("glass_amd.train", "USE_EVAL_GRAPH", False), ("glass_amd.models", "USE_STACK", False), ("glass_amd.ops", "USE_FUSED_DENSE", False),
             ("glass_amd.stack", "USE_COMB_EFF", False), ("glass_amd.stack", "USE_GN_EXACT", False), ("glass_amd.stack", "USE_READOUT", False),
             ("glass_amd.stack", "USE_READOUT_TWO", False), ("glass_amd.stack", "USE_GN_BWD_IN_COMB", False),
             ("glass_amd.stack", "USE_GATHER_IN_TRANS", False), ("glass_amd.stack", "USE_EMBED_TABLE", False),
             ("glass_amd.stack", "USE_FUSED_TAIL", False), ("glass_amd.stack", "USE_FUSED_BWD", False), ("glass_amd.ops", "DENSE_F32_PRODUCTS", True)]


@pytest.mark.parametrize("module,name,value", _SWITCHES, ids=[f"{m.split('.')[-1]}.{n}" for m, n, _ in _SWITCHES])
def test_every_python_switch_family_reproduces_g8(module, name, value, monkeypatch):
    """One suite case per Python A/B switch (the environment variables GLASS_TRAIN_STEP / GLASS_TRAIN_GRAPH / GLASS_HEAD_LABELS /
    GLASS_EVAL_GRAPH / GLASS_STACK / GLASS_FUSED_DENSE / GLASS_COMB_EFF / GLASS_GN_EXACT / GLASS_READOUT(_TWO) /
    GLASS_GN_BWD_IN_COMB / GLASS_GATHER_IN_TRANS / GLASS_EMBED_TABLE / GLASS_FUSED_TAIL / GLASS_FUSED_BWD / GLASS_DENSE_SPLIT) in
    its NON-default position: the reference's three Adam steps (fixture g8: losses and final weights from a run of the
    reference itself) through impl.train.train with the reference's own objects, then an evaluation pass through
    impl.train.test that equals eager forwards — every fallback form of the product is the same arithmetic."""
    import importlib
    from impl import SubGDataset, train, metrics, utils
    monkeypatch.setattr(importlib.import_module(module), name, value)
    g, x, ei, ew, pos, y = _g8()
    gnn = reference_build_model(int(g["hidden"]), int(g["layers"]), 0.0, True, str(g["pool"]), float(g["z_ratio"]), str(g["aggr"]),
                                torch.max(x), 3)
    gnn.load_state_dict(sd_from(g))
    optimizer = Adam(gnn.parameters(), lr=float(g["lr"]))
    loss_fn = CrossEntropyLoss()
    got = []
    for k in range(3):
        ds = SubGDataset.GDataset(x, ei, ew, pos[4 * k:4 * k + 4], y[4 * k:4 * k + 4])
        got.append(train.train(optimizer, gnn, reference_loader(ds, 4, shuffle=False), loss_fn))
    assert np.allclose(got, g["losses"], rtol=1e-5, atol=0), (module, name, got, g["losses"])
    end = sd_from(g, "sd_end/")
    keys = sorted(end)
    mine = {k: v.cpu() for k, v in gnn.state_dict().items()}
    assert rel_inf(flat_grads(mine, keys), flat_grads(end, keys)) < 1e-4
    ds = SubGDataset.GDataset(x, ei, ew, pos, y)
    _s, l1 = train.test(gnn, reference_loader(ds, 4, shuffle=False, drop_last=False), metrics.microf1, loss_fn)
    gnn.eval()
    with torch.no_grad():
        want = loss_fn(torch.cat([gnn(x, ei, ew, pos[i:i + 4], utils.MaxZOZ(x, pos[i:i + 4])) for i in range(0, pos.shape[0], 4)]), y)
    assert abs(float(l1) - float(want)) <= 1e-6 * max(1.0, abs(float(want)))
