"""The link-prediction pre-training step as a step program (glass_amd/ssl.py: StackProgram in unlabeled mode + the pair head
kernels of pairhead.hip) against the fp64 oracle — reference impl/models.py:361-509, GNNEmb.py:108-163."""
import functools

import numpy as np
import pytest
import torch
import torch.nn as nn

from helpers import rel_inf, flat_grads, record_parity, load, sd_from, grads_from
from oracle import glass_oracle as O

pytestmark = pytest.mark.gpu
TOL = 1e-5
DEV = "cuda:0"


def _build(h, layers, max_deg, aggr, dropout=0.0, act=None):
    from impl import models
    act = act or nn.ReLU(inplace=True)
    conv = models.EmbGConv(h, h, h, layers, max_deg=max_deg, activation=act, jk=False, dropout=dropout,
                           conv=functools.partial(models.MyGCNConv, aggr=aggr, activation=act), gn=True)
    head = models.MLP(h, h, 1, 2, dropout=dropout, activation=nn.ReLU(inplace=True))
    return models.EdgeGNN(conv, nn.ModuleList([head]), nn.ModuleList([models.MeanPool()]))


def _oracle(sd, h, layers, max_deg, aggr, x, ei, ew, pairs, y, dt, relu_masks=None):
    orc = O.OracleEdgeGNN(h, layers, max_deg, aggr=aggr, jk=False)
    orc.load_state_dict(sd)
    orc = orc.to(dt).train()
    O.relu_mask_feed(relu_masks or [])
    po = orc(x, ei, ew.to(dt), pairs)
    O.relu_mask_feed([])
    lo = nn.BCEWithLogitsLoss()(po.flatten(), y.to(dt))
    lo.backward()
    return po.detach(), lo.item(), {k: p.grad for k, p in orc.named_parameters()}


def _program_masks(prog, H):
    """The ReLU branches the program took, in the oracle's call order: per layer the trans ReLU (pre-activation kept in T's
    second half: the Linear is the second half of a pair), between layers the ReLU behind gns[l] (h of the next layer, dropout
    off), last the head's."""
    masks = []
    layers = prog.last["layers"]
    for l, rec in enumerate(layers):
        if l > 0:
            masks.append((rec["h"] > 0).cpu())
        masks.append((rec["T"][:, H:] > 0).cpu())
    masks.append((prog.last["hid"] > 0).cpu())
    return masks


@pytest.mark.parametrize("layers,aggr,n_pairs,features", [(2, "mean", 131072, "deg"), (3, "gcn", 50001, "deg"), (1, "sum", 4097, "nodeid")])
def test_pair_program_vs_oracle(layers, aggr, n_pairs, features):
    """ppi_bp-shaped graph (N = 17 080, nnz = 633 902), hidden 64, dropout 0: loss, predictions and every gradient of the
    step program against the fp64 oracle ON THE SAME ReLU BRANCHES (ReLU is not differentiable at 0: a handful of the
    millions of pre-activations lie within fp32 rounding of it — tests/test_gpu_model.py::test_pretraining_step_full_size_vs_oracle),
    plain 1e-5 bar.  131 072 pairs = the reference's batch (GNNEmb.py:144); the others are ragged against the 64-pair
    forward tiles and the 256-pair weight-gradient slabs; use_nodeid features make the table as large as the graph."""
    from glass_amd import synth, ssl
    from glass_amd.arena import ParamArena
    w, ei, ew, x, _pos, _y = synth.make_workload("ppi_bp", seed=0, n_batches=1)
    rng = np.random.default_rng(3 + layers)
    h = 64
    pairs = rng.integers(0, w.n_node, size=(n_pairs, 2))
    pairs[: n_pairs // 50, 0] = 7          # a hub node: a list of > 64 entries (summed by the whole workgroup)
    pairs = torch.from_numpy(pairs)
    y = torch.from_numpy(rng.integers(0, 2, size=n_pairs).astype(np.float32))
    if features == "nodeid":
        x = np.arange(w.n_node, dtype=np.int64).reshape(-1, 1, 1)
    ei, ew, x = (torch.from_numpy(a) for a in (ei, ew, x))
    torch.manual_seed(layers)
    model = _build(h, layers, int(x.max()), aggr)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    model.to(DEV).train()
    arena = ParamArena(model)
    prog = ssl.program_for(model)
    assert prog is not None and prog.covers_arena()
    arena.flat.fill_(7.0)  # overwrite mode: stale contents must not survive
    loss = prog.loss_and_grads(x.to(DEV), ei.to(DEV), ew.to(DEV), pairs.to(DEV), y.to(DEV))
    torch.cuda.synchronize()
    mine = {k: p.grad.cpu().clone() for k, p in model.named_parameters()}
    pred = prog.last["logits"].cpu()
    masks = _program_masks(prog, h)
    assert len(masks) == 2 * layers
    po, lo, theirs = _oracle(sd, h, layers, int(x.max()), aggr, x, ei, ew, pairs, y, torch.float64, masks)
    keys = sorted(mine)
    assert keys == sorted(theirs)
    e_pred, e_loss = rel_inf(pred, po.flatten()), abs(loss.item() - lo) / abs(lo)
    e_grad = rel_inf(flat_grads(mine, keys), flat_grads(theirs, keys))
    print(f"pair program L={layers} {aggr} P={n_pairs} {features}: pred {e_pred:.2e} loss {e_loss:.2e} grad {e_grad:.2e}")
    record_parity(f"pair_program/ppi_bp_L{layers}_{aggr}_P{n_pairs}_{features}", pred_rel_inf=e_pred, loss_rel=e_loss, grad_rel_inf=e_grad)
    assert e_pred < TOL and e_loss < TOL and e_grad < TOL
    # the zero halves of the single-Linear layout stay zero, their gradients too
    c0 = model.conv.convs[0]
    assert float(c0._stack["trans"][0][:h].abs().max()) == 0.0 and float(c0._stack["trans"][2][:h].abs().max()) == 0.0
    # evaluation forward of the program = the training forward's logits (dropout 0)
    assert rel_inf(prog.predict(x.to(DEV), ei.to(DEV), ew.to(DEV), pairs.to(DEV)).cpu().flatten(), pred) < 1e-6


def test_pair_program_g10_golden_shape_is_served_by_the_per_op_path():
    """The reference-run fixtures g10 are hidden 8: not the program's width — the model must still run (per-op path) and match."""
    from glass_amd import ssl
    from glass_amd.arena import ParamArena
    g = load("g10_edgegnn_L2_jk0_mean.npz")
    model = _build(int(g["hidden"]), 2, int(g["x"].max()), "mean")
    model.load_state_dict(sd_from(g))
    model.to(DEV).train()
    ParamArena(model)
    assert ssl.program_for(model) is None
    pred = model(*(torch.from_numpy(g[k]).to(DEV) for k in ("x", "edge_index", "edge_weight", "pairs")))
    assert rel_inf(pred.detach().cpu(), g["pred64"]) < TOL


def test_pair_program_dropout_graph_replay_and_repeatability():
    """Dropout 0.5 (the driver's search space, GNNEmb.py:171): (a) two runs from the same seed are bitwise equal (no float
    atomic anywhere in the step); (b) the hipGraph replay of GNNEmb.GraphedPairStep equals the eager program step; (c) replays
    draw fresh masks (dropout under capture) and about half of the head's hidden units are dropped; (d) a step
    cached for another loss function / graph is not reused."""
    import GNNEmb
    from glass_amd import synth, ssl, ops
    from glass_amd.optim import FlatAdam
    w, ei, ew, x, _pos, _y = synth.make_workload("ppi_bp", seed=0, n_batches=1)
    rng = np.random.default_rng(5)
    n_pairs = 20000
    pairs = torch.from_numpy(rng.integers(0, w.n_node, size=(3, n_pairs, 2))).to(DEV)
    y = torch.from_numpy(rng.integers(0, 2, size=(3, n_pairs)).astype(np.float32)).to(DEV)
    ei, ew, x = (torch.from_numpy(a).to(DEV) for a in (ei, ew, x))

    def run(graph):
        import os
        os.environ["GLASS_SSL_GRAPH"] = "1" if graph else "0"
        torch.manual_seed(0)
        ops.rng_seed(99, DEV)
        model = _build(64, 2, int(x.max()), "mean", dropout=0.5).to(DEV).train()
        opt = GNNEmb.Pretrain.make_optimizer(model, 1e-2)
        assert isinstance(opt, FlatAdam)
        step = GNNEmb.GraphedPairStep(model, lambda p, t: nn.BCEWithLogitsLoss()(p.flatten(), t.flatten()), x, ei, ew, bce_mean=True)
        assert step.program is not None
        losses, zero_share = [], []
        for k in range(6):
            losses.append(step(pairs[k % 3], y[k % 3]).item())
            zero_share.append(float((step.program.last["hid"] == 0).float().mean()))
            opt.step()
        torch.cuda.synchronize()
        os.environ.pop("GLASS_SSL_GRAPH")
        return model.conv._glass_arena.flat_param.clone(), losses, zero_share, step
    a, la, za, step_a = run(True)
    b, lb, _zb, _ = run(True)
    c, lc, _zc, step_c = run(False)
    assert torch.equal(a, b) and la == lb                       # (a)
    assert torch.equal(a, c) and la == lc                       # (b) replay == eager
    assert step_a.graphs and not step_c.graphs
    assert all(np.isfinite(v) for v in la) and la[3] != la[0]   # (c) batch 0 again at k = 3: new masks (and new weights)
    assert all(0.6 < z < 0.9 for z in za)                       # relu zeroes ~half, dropout half of the rest
    # (d) the cache key
    import GNNEmb as G
    run_obj = G.Pretrain.__new__(G.Pretrain)
    assert step_a.key[0] != id(nn.BCEWithLogitsLoss())


def test_pair_head_kernels_vs_fp64():
    """K9 alone through the C ABI on random inputs (P not a multiple of any tile, a hub node, dropout off): logits, loss,
    dW0 / db0 / dw1 / db1 and demb against fp64 autograd."""
    from glass_amd import _lib
    lib = _lib.load()
    n, P, H = 3000, 10007, 64
    g = torch.Generator().manual_seed(1)
    emb = torch.randn(n, H, generator=g)
    pairs = torch.randint(0, n, (P, 2), generator=g)
    pairs[:300, 1] = 11
    y = torch.randint(0, 2, (P, ), generator=g).float()
    W0, b0 = torch.randn(H, H, generator=g) * 0.2, torch.randn(H, generator=g) * 0.1
    w1, b1 = torch.randn(H, generator=g) * 0.3, torch.randn(1, generator=g)
    d = [t.to(DEV).contiguous() for t in (emb, pairs, y, W0, b0, w1, b1)]
    hid, logits, dlogit = torch.empty(P, H, device=DEV), torch.empty(P, device=DEV), torch.empty(P, device=DEV)
    ws = torch.empty(int(lib.glass_pair_head_ws_bytes(n, P)) + 16, dtype=torch.uint8, device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    _lib.check(lib.glass_pair_head_fwd_f32(d[0].data_ptr(), H, n, d[1].data_ptr(), P, d[3].data_ptr(), d[4].data_ptr(), d[5].data_ptr(),
                                           d[6].data_ptr(), d[2].data_ptr(), 0.0, 0, 2, 0, hid.data_ptr(), logits.data_ptr(), dlogit.data_ptr(),
                                           ws.data_ptr(), st), "fwd")
    dW0, db0, dw1, db1 = torch.empty(H, H, device=DEV), torch.empty(H, device=DEV), torch.empty(H, device=DEV), torch.empty(1, device=DEV)
    loss, demb = torch.empty((), device=DEV), torch.empty(n, H, device=DEV)
    _lib.check(lib.glass_pair_head_bwd_f32(d[0].data_ptr(), H, n, d[1].data_ptr(), P, d[3].data_ptr(), d[5].data_ptr(), hid.data_ptr(),
                                           dlogit.data_ptr(), 0.0, dW0.data_ptr(), db0.data_ptr(), dw1.data_ptr(), db1.data_ptr(), 0,
                                           loss.data_ptr(), demb.data_ptr(), H, ws.data_ptr(), st), "bwd")
    torch.cuda.synchronize()
    e64, W64, b64, w64, bb64 = (t.double().requires_grad_(True) for t in (emb, W0, b0, w1, b1))
    pooled = e64[pairs].mean(dim=1)
    pre = pooled @ W64.T + b64
    mask = (hid.cpu() > 0).double()   # the branches the kernel took
    h64 = pre * mask
    x64 = h64 @ w64 + bb64
    l64 = nn.BCEWithLogitsLoss()(x64, y.double())
    l64.backward()
    assert rel_inf(logits.cpu(), x64.detach()) < 1e-6 and abs(loss.item() - l64.item()) < 1e-6 * abs(l64.item())
    for mine, ref in ((dW0, W64.grad), (db0, b64.grad), (dw1, w64.grad), (db1, bb64.grad), (demb, e64.grad)):
        assert rel_inf(mine.cpu(), ref) < 2e-6
    demb2 = torch.empty_like(demb)
    _lib.check(lib.glass_pair_head_bwd_f32(d[0].data_ptr(), H, n, d[1].data_ptr(), P, d[3].data_ptr(), d[5].data_ptr(), hid.data_ptr(),
                                           dlogit.data_ptr(), 0.0, dW0.data_ptr(), db0.data_ptr(), dw1.data_ptr(), db1.data_ptr(), 0,
                                           loss.data_ptr(), demb2.data_ptr(), H, ws.data_ptr(), st), "bwd")
    torch.cuda.synchronize()
    assert torch.equal(demb, demb2)   # bitwise repeatable (exact fixed-point sums over lists in arbitrary order)
