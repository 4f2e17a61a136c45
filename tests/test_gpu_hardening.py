"""Parity hardening: the reference's constructor defaults the fused kernels do not
cover, non-finite values through the exact fixed-point sums, and the C ABI's repeatability contract."""
import functools
import warnings

import numpy as np
import pytest
import torch
import torch.nn as nn

from helpers import load, sd_from, grads_from, rel_inf, flat_grads, record_parity, build_glass
from oracle import glass_oracle as O

pytestmark = pytest.mark.gpu
TOL = 1e-5
DEV = "cuda:0"

POOLS = {"mean": "MeanPool", "max": "MaxPool", "sum": "AddPool", "size": "SizePool"}


def _build_defaults(hidden, layers, max_deg, n_class, aggr, pool, zr, gn, act):
    """GLASS built with the REFERENCE'S constructor defaults where the driver overrides them (impl/models.py:125,192:
    activation nn.ReLU(); :194 gn flag; :300-303 MaxPool)."""
    from impl import models
    conv = models.EmbZGConv(hidden, hidden, layers, max_deg=max_deg, activation=act, jk=True, dropout=0.0,
                            conv=functools.partial(models.GLASSConv, aggr=aggr, z_ratio=zr, dropout=0.0), gn=gn)
    return models.GLASS(conv, nn.ModuleList([nn.Linear(hidden * layers, n_class)]), nn.ModuleList([getattr(models, POOLS[pool])()]))


@pytest.mark.parametrize("name", ["relu_gn_max_mean", "relu_nogn_sum_gcn", "relu_gn_size_sum"])
@pytest.mark.parametrize("arena", [False, True])
def test_g11_reference_constructor_defaults(name, arena):
    """The product on the paths BEHIND the fused kernels (library GEMM + stand-alone mix / GraphNorm / torch activation):
    hidden 48 (no dense family), nn.ReLU(), gn=False, MaxPool — against vectors the reference itself produced (fixture g11,
    fp64 evaluation; tests/golden/make_golden.py g11): logits, loss, every gradient.  With and without a parameter arena."""
    from impl import utils
    from glass_amd import stack, losses
    from glass_amd.arena import ParamArena
    g = load(f"g11_defaults_{name}.npz")
    x, ei, ew = (torch.from_numpy(g[k]).to(DEV) for k in ("x", "edge_index", "edge_weight"))
    pos, y = torch.from_numpy(g["pos"]).to(DEV), torch.from_numpy(g["y"]).to(DEV)
    model = _build_defaults(int(g["hidden"]), int(g["layers"]), int(g["x"].max()), 3, str(g["aggr"]), str(g["pool"]),
                            float(g["z_ratio"]), bool(g["gn"]), nn.ReLU())
    model.load_state_dict(sd_from(g))
    model.to(DEV).train()
    if arena:
        ParamArena(model)
    assert not stack.step_supported(model, losses.CrossEntropy())  # this IS the fallback path
    z = utils.MaxZOZ(x, pos)
    assert np.array_equal(z.cpu().numpy(), g["z"])
    pred = model(x, ei, ew, pos, z)
    loss = nn.CrossEntropyLoss()(pred, y)
    loss.backward()
    ref = grads_from(g, "grad64/")
    keys = sorted(ref)
    mine = {k: p.grad.cpu() for k, p in model.named_parameters()}
    assert sorted(mine) == keys
    e_pred, e_loss = rel_inf(pred.detach().cpu(), g["pred64"]), abs(loss.item() - float(g["loss64"])) / abs(float(g["loss64"]))
    e_grad = rel_inf(flat_grads(mine, keys), flat_grads(ref, keys))
    record_parity(f"fallback_path/g11_{name}{'_arena' if arena else ''}", logits_rel_inf=e_pred, loss_rel=e_loss, grad_rel_inf=e_grad)
    assert e_pred < TOL and e_loss < TOL and e_grad < TOL
    assert rel_inf(pred.detach().cpu(), g["pred"]) < TOL + rel_inf(g["pred"], g["pred64"])


@pytest.mark.parametrize("hidden,act,gn,pool", [(96, "relu", True, "max"), (96, "relu", False, "sum"), (50, "relu", True, "mean"),
                                                (96, "elu", True, "max"), (48, "elu", False, "size")])
def test_model_outside_the_fused_family_vs_oracle(hidden, act, gn, pool):
    """A mid-size graph (N = 3 000, 40 000 edges, ragged subgraphs) at widths / activations / pools / gn flags the step
    program does not take, whatever path the product picks for them, against the fp64 oracle."""
    from impl import utils
    from glass_amd import synth
    from glass_amd.arena import ParamArena
    n, K = 3000, 4
    ei, ew = synth.make_graph(n, 20000, 21, 0.4)
    x = synth.degree_feature(ei, n)
    pos, y = synth.make_subgraphs(n, 30, 12, K, 1, False)
    pos[2, 5:] = -1
    pos[4, :3] = pos[5, :3]
    ei, ew, x, pos, y = (torch.from_numpy(a) for a in (ei, ew, x, pos, y))
    torch.manual_seed(hidden)
    act_mod = nn.ReLU() if act == "relu" else nn.ELU(inplace=True)
    if act == "elu" and not gn:
        act_mod = nn.ELU()  # (in place without gns the reference would alias the JK tensors, impl/models.py:254-258)
    model = _build_defaults(hidden, 2, int(x.max()), K, "mean", pool, 0.85, gn, act_mod)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    model.to(DEV).train()
    ParamArena(model)
    xg, eig, ewg, posg, yg = (t.to(DEV) for t in (x, ei, ew, pos, y))
    pred = model(xg, eig, ewg, posg, utils.MaxZOZ(xg, posg))
    loss = nn.CrossEntropyLoss()(pred, yg)
    loss.backward()
    orc = O.OracleGLASS(hidden, 2, int(x.max()), K, aggr="mean", pool=pool, z_ratio=0.85, gn=gn, act=act)
    orc.load_state_dict(sd)
    orc = orc.double().train()
    po = orc(x, ei, ew.double(), pos, O.max_zero_one(x, pos))
    lo = nn.CrossEntropyLoss()(po, y)
    lo.backward()
    theirs = {k: p.grad for k, p in orc.named_parameters()}
    mine = {k: p.grad.cpu() for k, p in model.named_parameters()}
    keys = sorted(mine)
    assert keys == sorted(theirs)
    e_pred, e_loss = rel_inf(pred.detach().cpu(), po.detach()), abs(loss.item() - lo.item()) / abs(lo.item())
    e_grad = rel_inf(flat_grads(mine, keys), flat_grads(theirs, keys))
    record_parity(f"fallback_path/N3000_hidden{hidden}_{act}_gn{int(gn)}_{pool}", logits_rel_inf=e_pred, loss_rel=e_loss, grad_rel_inf=e_grad)
    assert e_pred < TOL and e_loss < TOL and e_grad < TOL


# ---- non-finite values must stay visible (ADVICE r3 on bucket.h) -------------------------
def _c2_like(hidden=64, n=20000, seed=0):
    """A graph large enough for the hidden-64 step program to use the exact GraphNorm accumulators (gn_acc.h)."""
    from glass_amd import synth
    ei, ew = synth.make_graph(n, 200000, seed, 0.0)
    x = synth.degree_feature(ei, n)
    pos, y = synth.make_subgraphs(n, 40, 10, 4, 1, False)
    return tuple(torch.from_numpy(a) for a in (ei, ew, x, pos, y))


def _finite_share(named):
    return {k: float(torch.isfinite(v).double().mean()) for k, v in named.items()}


def test_nonfinite_values_reach_the_outputs():
    """An Inf activation (a +inf bias entry in layer 0's trans pair) must come out of the step program as non-finite loss and
    gradients, as it does from the reference arithmetic (the CPU oracle here) — the exact fixed-point GraphNorm
    accumulators used to clamp a non-finite partial sum to +-4e18 and return a FINITE wrong statistic."""
    from glass_amd import stack, losses
    from glass_amd.arena import ParamArena
    ei, ew, x, pos, y = _c2_like()
    torch.manual_seed(0)
    model = build_glass(64, 2, int(x.max()), 4, "mean", "sum", 0.9)
    with torch.no_grad():
        model.conv.convs[0].trans_fns[1].bias[3] = float("inf")
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    loss_fn = losses.CrossEntropy()
    model.to(DEV).train()
    arena = ParamArena(model)
    assert stack.step_supported(model, loss_fn)
    xg, eig, ewg, posg, yg = (t.to(DEV) for t in (x, ei, ew, pos, y))
    loss, logits = stack.loss_and_grads(model, loss_fn, xg, eig, ewg, posg, "pos", yg, overwrite=stack.covers_arena(model, arena))
    torch.cuda.synchronize()
    orc = O.OracleGLASS(64, 2, int(x.max()), 4, aggr="mean", pool="sum", z_ratio=0.9)
    orc.load_state_dict(sd)
    orc.train()
    po = orc(x, ei, ew, pos, O.max_zero_one(x, pos))
    lo = nn.CrossEntropyLoss()(po, y)
    lo.backward()
    assert not torch.isfinite(lo)  # the reference arithmetic: NaN loss
    assert not torch.isfinite(loss), f"step program returned a finite loss {loss.item()} from an Inf activation"
    assert not bool(torch.isfinite(logits).all())
    mine = _finite_share({k: p.grad.cpu() for k, p in model.named_parameters()})
    theirs = _finite_share({k: p.grad for k, p in orc.named_parameters()})
    # every parameter whose reference gradient is entirely non-finite must not come out entirely finite here
    bad = [k for k in theirs if theirs[k] == 0.0 and mine[k] == 1.0]
    assert not bad, f"finite gradients where the reference has none: {bad}"


def test_out_of_range_gradient_sums_poison_instead_of_saturating():
    """A 1e20 upstream gradient overflows the backward sums' fixed-point range (|sum| < 2^38): the affected statistics must
    come out NaN (sticky poison bit), not as a saturated finite number.  Every gradient tensor is then either non-finite or
    right (the reference's, scaled) — never finite garbage."""
    from impl import utils
    from glass_amd import stack
    from glass_amd.arena import ParamArena
    ei, ew, x, pos, y = _c2_like(seed=1)
    torch.manual_seed(1)
    model = build_glass(64, 2, int(x.max()), 4, "mean", "sum", 0.9)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    model.to(DEV).train()
    ParamArena(model)
    assert stack.StackProgram.supported(model.conv)
    xg, eig, ewg, posg, yg = (t.to(DEV) for t in (x, ei, ew, pos, y))
    scale = 1e20
    pred = model(xg, eig, ewg, posg, utils.MaxZOZ(xg, posg))
    (nn.CrossEntropyLoss()(pred, yg) * scale).backward()
    torch.cuda.synchronize()
    orc = O.OracleGLASS(64, 2, int(x.max()), 4, aggr="mean", pool="sum", z_ratio=0.9)
    orc.load_state_dict(sd)
    orc = orc.double().train()
    lo = nn.CrossEntropyLoss()(orc(x, ei, ew.double(), pos, O.max_zero_one(x, pos)), y)
    (lo * scale).backward()
    n_bad = 0
    for k, p in model.named_parameters():
        g, r = p.grad.cpu().double(), dict(orc.named_parameters())[k].grad
        if bool(torch.isfinite(g).all()):
            assert rel_inf(g, r) < 1e-4, f"{k}: finite but wrong ({rel_inf(g, r):.2e}) — a saturated sum leaked through"
        else:
            n_bad += 1
    assert n_bad > 0, "the 1e20 gradient did not overflow any exact sum: the test no longer exercises the poison path"


@pytest.mark.parametrize("mode", ["sum", "mean", "max"])
@pytest.mark.parametrize("big", [False, True])
def test_pool_backward_propagates_nonfinite(mode, big):
    """A NaN / Inf in dout reaches demb on every pool backward form — ordered scatter (small batches), node-bucketed exact
    sums (beyond the LDS staging; ExactSum's sticky mark), max pooling's exact form — exactly on the rows the reference's
    scatter-add would poison, and nowhere else."""
    from glass_amd import ops
    n, C = 3000, 64
    B, S = (500, 37) if big else (40, 12)
    rng = np.random.default_rng(5 + big)
    pos = rng.integers(1, n, (B, S))
    pos[rng.random((B, S)) < 0.2] = -1
    pos[:, 0] = 0                      # node 0 in every subgraph: a long list
    post = torch.from_numpy(pos)
    emb = torch.randn(n, C, generator=torch.Generator().manual_seed(1))
    gout = torch.randn(B, C, generator=torch.Generator().manual_seed(2))
    gout[7, 5] = float("nan")
    gout[9, 11] = float("inf")
    ec = emb.double().requires_grad_(True)
    batch, p = O.pad_to_batch(post)
    O.segment_pool(ec[p], batch, B, mode).backward(gout.double())
    eg = emb.to(DEV).requires_grad_(True)
    ops.segment_pool(eg, post.to(DEV), mode).backward(gout.to(DEV))
    ref_bad, mine_bad = ~torch.isfinite(ec.grad), ~torch.isfinite(eg.grad.cpu())
    assert bool(ref_bad.any())
    if mode == "max":
        # torch_scatter routes a subgraph's gradient to the arg-max node only; torch's scatter_reduce("amax") backward (the
        # oracle's stand-in) multiplies the gradient by a 0/1 mask, which turns the Inf into NaN on EVERY row of subgraph 9
        # (0 * inf) — an artefact of the stand-in.  Expected here: exactly the two winning nodes, nothing else.
        want = torch.zeros_like(mine_bad)
        for b, c in ((7, 5), (9, 11)):
            rows = post[b][post[b] >= 0]
            want[rows[emb[rows, c].argmax()], c] = True
        assert torch.equal(mine_bad, want) and bool((ref_bad | ~mine_bad).all())
        ok = ~ref_bad
    else:
        assert torch.equal(ref_bad, mine_bad), f"{int((ref_bad != mine_bad).sum())} entries differ in finiteness"
        ok = ~ref_bad
    assert rel_inf(eg.grad.cpu()[ok], ec.grad[ok]) < 1e-6


def test_atomic_pool_backward_is_a_separate_entry_point():
    """glass_segment_pool_bwd_atomic_f32 (the one float-atomic scatter left, exported under its own name) agrees with the
    exact form within rounding; the repeatable entry point refuses the same call with GLASS_E_WS."""
    from glass_amd import _lib, ops
    lib = _lib.load()
    n, B, S, C = 4000, 500, 37, 64
    rng = np.random.default_rng(3)
    pos = torch.from_numpy(rng.integers(0, n, (B, S))).to(DEV)
    dout = torch.randn(B, C, device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    demb = torch.zeros(n, C, device=DEV)
    assert lib.glass_segment_pool_bwd_f32(dout.data_ptr(), C, pos.data_ptr(), B, S, 0, None, demb.data_ptr(), C, n, C, st) == -4
    _lib.check(lib.glass_segment_pool_bwd_atomic_f32(dout.data_ptr(), C, pos.data_ptr(), B, S, 0, None, demb.data_ptr(), C, n, C, st),
               "glass_segment_pool_bwd_atomic_f32")
    exact = torch.empty(n, C, device=DEV)
    ws = torch.empty(int(lib.glass_segment_pool_bwd_exact_ws_bytes(n, B, S)) + 16, dtype=torch.uint8, device=DEV)
    _lib.check(lib.glass_segment_pool_bwd_exact_f32(dout.data_ptr(), C, pos.data_ptr(), B, S, 0, exact.data_ptr(), C, n, C, ws.data_ptr(), st),
               "glass_segment_pool_bwd_exact_f32")
    torch.cuda.synchronize()
    assert rel_inf(demb.cpu(), exact.cpu()) < 1e-6


def test_scratch_buffers_are_retired_not_freed():
    """ops._scratch hands a captured graph's pointer to nobody else — a buffer that has to grow is kept alive,
    and the sum / max pool backward use distinct keys."""
    from glass_amd import ops
    a = ops._scratch(("t_retire", 1), torch.device(DEV), 1000)
    pa = a.data_ptr()
    b = ops._scratch(("t_retire", 1), torch.device(DEV), 100000)
    assert b.data_ptr() != pa or b.numel() >= 100000
    assert any(t.data_ptr() == pa for t in ops._retired_scratch)
    assert ops._scratch(("t_retire", 1), torch.device(DEV), 10).data_ptr() == b.data_ptr()


# ---------------------------------------------------------------------- product forms of the tiled dense family: edges
def _comb256(scale_log2=0, inf_at=None, gout_inf_at=None, f32_form=False):
    """A hidden-256 comb pair (no activation, zero bias) through ops.dual_linear_mix in one product form; operands scaled by
    2^-scale_log2.  -> (out, d xa, fp64 out, fp64 d xa)"""
    from glass_amd import ops
    from test_gpu_kernels import _pack
    H, N, zr = 256, 2100, 0.8
    gen = torch.Generator().manual_seed(5)
    W = torch.randn(2 * H, 2 * H, generator=gen) / (2 * H) ** 0.5
    xa_h, xb_h = torch.randn(N, H, generator=gen), torch.randn(N, H, generator=gen)
    mask = torch.rand(N, generator=gen) < 0.05
    sc = 2.0 ** (-scale_log2)
    xa64, xb64 = (xa_h.double() * sc).requires_grad_(True), xb_h.double() * sc
    Z = torch.cat((xa64, xb64), -1) @ W.double().t()
    ref = O._mix(mask.reshape(-1, 1), zr, Z[:, :H], Z[:, H:])
    ref.backward(torch.full_like(ref, sc))
    prev = ops.DENSE_F32_PRODUCTS
    ops.DENSE_F32_PRODUCTS = f32_form  # an option of every dense CALL (its act word), not library state
    try:
        Wg, bg = W.to(DEV), torch.zeros(2 * H, device=DEV)
        dW, db = torch.zeros_like(Wg), torch.zeros_like(bg)
        Wimg, WTimg = _pack(Wg, False, H, zr), _pack(Wg, True, H, zr)
        lin1, lin0 = nn.Linear(2 * H, H).to(DEV), nn.Linear(2 * H, H).to(DEV)
        lin1.weight.grad, lin0.weight.grad, lin1.bias.grad, lin0.bias.grad = dW[:H], dW[H:], db[:H], db[H:]
        xa = (xa_h * sc).to(DEV)
        if inf_at is not None:
            xa[inf_at] = float("inf")
        xa.requires_grad_(True)
        xb = (xb_h * sc).to(DEV).requires_grad_(True)
        out = ops.dual_linear_mix(xa, xb, lin1, lin0, mask.to(DEV).to(torch.uint8), zr, 0, (Wg, bg, dW, db, Wimg, WTimg))
        g = torch.full_like(out, sc)
        if gout_inf_at is not None:
            g[gout_inf_at] = float("inf")
        out.backward(g)
    finally:
        ops.DENSE_F32_PRODUCTS = prev
    return out.detach().cpu(), xa.grad.cpu(), ref.detach(), xa64.grad


@pytest.mark.parametrize("f32_form", [False, True])
def test_tiled_product_forms_nonfinite_in_nonfinite_out(f32_form):
    """an Inf activation / an Inf upstream gradient through a hidden-256 layer, in BOTH product forms: every
    value the Inf feeds comes out non-finite (the f32-input MFMA gives +-Inf / NaN as the reference's fp32 GEMM does; the split
    form cuts Inf into (Inf, NaN, NaN) and gives NaN — documented in include/glass_hip.h at GLASS_DENSE_F32_PRODUCTS), and no
    other row is touched."""
    clean, dclean, _r, _d = _comb256(f32_form=f32_form)
    out, dx, _r, _d = _comb256(inf_at=(5, 7), gout_inf_at=(9, 3), f32_form=f32_form)
    assert not bool(torch.isfinite(out[5]).any()), "an Inf operand left finite outputs in its row"
    assert bool(torch.isnan(out[5]).any()) if not f32_form else bool(torch.isinf(out[5]).any())
    assert not bool(torch.isfinite(dx[9]).any()), "an Inf upstream gradient left finite data gradients in its row"
    rows = torch.ones(out.shape[0], dtype=torch.bool)
    rows[5] = False
    assert torch.equal(out[rows], clean[rows])
    rows[5], rows[9] = True, False
    assert torch.equal(dx[rows], dclean[rows])


@pytest.mark.parametrize("scale_log2", [0, 100, 110, 115, 120])
def test_tiled_product_forms_near_the_bottom_of_the_range(scale_log2):
    """Operands scaled by 2^-100 .. 2^-120.  The f32-input form holds 1e-5 against fp64 throughout.  The
    split form (three bf16 pieces per operand) holds it while the low piece stays in bf16's normal range — measured: down to
    operand magnitudes of 2^-115 (8e-7 .. 1.6e-6); at 2^-120 the low pieces flush and the product keeps ~16 significant
    bits (3e-5 / 8e-5): the documented limit below which a caller passes GLASS_DENSE_F32_PRODUCTS (|x| < 2^-117 ~ 6e-36 is
    far below anything a GraphNorm-ed activation takes)."""
    out, dx, ref, dref = _comb256(scale_log2, f32_form=True)
    assert rel_inf(out, ref) < 1e-5 and rel_inf(dx, dref) < 1e-5
    out, dx, ref, dref = _comb256(scale_log2, f32_form=False)
    e_out, e_dx = rel_inf(out, ref), rel_inf(dx, dref)
    record_parity(f"split_form_operands_2^-{scale_log2}", out_rel_inf=e_out, dx_rel_inf=e_dx)
    if scale_log2 <= 115:
        assert e_out < 1e-5 and e_dx < 1e-5, (e_out, e_dx)
    else:
        assert e_out < 2e-4 and e_dx < 2e-4, (e_out, e_dx)  # 16 significant bits left: degraded, never garbage


def test_two_threads_two_streams_two_product_forms():
    """the library is re-entrant per stream with no global state.  Two host threads, each on its own HIP stream,
    call glass_dual_linear_fwd_f32 concurrently with DIFFERENT option words (f32-input products / split products): every one
    of each thread's results is bit-identical to that form's single-threaded result — and the two forms do differ in bits,
    so a leak of one thread's option into the other's calls would show."""
    import threading
    from glass_amd import _lib
    from test_gpu_kernels import _pack
    lib = _lib.load()
    H, N, zr, iters = 256, 4099, 0.8, 40
    gen = torch.Generator().manual_seed(17)
    W = (torch.randn(2 * H, H, generator=gen) / H ** 0.5).to(DEV)
    bias = (0.1 * torch.randn(2 * H, generator=gen)).to(DEV)
    xa = torch.randn(N, H, generator=gen).to(DEV)
    mask = (torch.rand(N, generator=gen) < 0.1).to(DEV).to(torch.uint8)
    Wimg = _pack(W, False, H, zr)

    def fwd(word, T, out, stream):
        rc = lib.glass_dual_linear_fwd_f32(xa.data_ptr(), xa.stride(0), 0, 0, Wimg.data_ptr(), bias.data_ptr(), mask.data_ptr(), zr,
                                           word, T.data_ptr(), T.stride(0), out.data_ptr(), out.stride(0), N, H, 0, 0, 0, 0, 0, 0.0,
                                           0, 0, 0, 0, 0, 0, stream)
        assert rc == 0, lib.glass_last_error_string()

    words = {"f32": 1 | _lib.DENSE_F32_PRODUCTS, "split": 1}
    want = {}
    for name, word in words.items():
        T, out = torch.empty(N, 2 * H, device=DEV), torch.empty(N, H, device=DEV)
        fwd(word, T, out, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        want[name] = out.clone()
    assert not torch.equal(want["f32"], want["split"]), "the two product forms agree bit for bit: the test could not see a leak"
    got, errors = {}, []

    def worker(name):
        try:
            st = torch.cuda.Stream()
            outs = [torch.empty(N, H, device=DEV) for _ in range(iters)]
            T = torch.empty(N, 2 * H, device=DEV)
            for o in outs:
                fwd(words[name], T, o, st.cuda_stream)
            st.synchronize()
            got[name] = outs
        except Exception as e:  # noqa: BLE001
            errors.append((name, repr(e)))

    torch.cuda.synchronize()
    threads = [threading.Thread(target=worker, args=(n, )) for n in words]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for name in words:
        bad = [i for i, o in enumerate(got[name]) if not torch.equal(o, want[name])]
        assert not bad, f"thread '{name}': results {bad} differ from its own product form's"
