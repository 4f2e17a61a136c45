"""Widths outside the kernel families run zero-padded at the next family width (glass_amd/widths.py): the construction, the
logical state_dict, and — on the GPU — the step program against the fp64 oracle on the logical weights."""
import os
import sys

import pytest
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from glass_amd import widths  # noqa: E402
from glass_amd.factory import build_glass  # noqa: E402


def test_fused_width_rule():
    assert [widths.fused_width(h) for h in (8, 17, 32, 33, 48, 64, 65, 96, 128, 129, 200, 256, 300, 512, 513)] == \
        [8, 17, 32, 64, 64, 64, 128, 128, 128, 256, 256, 256, 512, 512, 513]


@pytest.mark.parametrize("hidden,jk", [(96, True), (48, False), (100, True)])
def test_padded_construction_is_the_unpadded_one(hidden, jk):
    """Same logical parameters as an unpadded build from the same seed, the generator left where that build leaves it, zero
    padding, logical state_dict in both directions."""
    torch.manual_seed(11)
    plain = build_glass(hidden, 3, 12, 5, "mean", "sum", 0.8, jk=jk, pad_width=False)
    after_plain = torch.rand(3)
    torch.manual_seed(11)
    padded = build_glass(hidden, 3, 12, 5, "mean", "sum", 0.8, jk=jk)
    after_padded = torch.rand(3)
    assert torch.equal(after_plain, after_padded)
    H, Hp = padded._glass_logical_width
    assert (H, Hp) == (hidden, widths.fused_width(hidden)) and Hp > H
    sd_plain, sd = plain.state_dict(), padded.state_dict()
    assert sorted(sd) == sorted(sd_plain)
    for k, v in sd_plain.items():
        assert torch.equal(sd[k], v), k
    for name, p in padded.named_parameters():
        lg = widths.unpad_tensor(p.detach(), sd_plain[name].shape, H, Hp)
        assert torch.equal(lg, sd_plain[name])
        assert float(p.detach().abs().sum()) == pytest.approx(float(lg.abs().sum()), rel=1e-6), name  # nothing outside the blocks
    # a reference checkpoint (logical shapes) loads; the padding stays zero
    torch.manual_seed(12)
    other = build_glass(hidden, 3, 12, 5, "mean", "sum", 0.8, jk=jk, pad_width=False)
    padded.load_state_dict(other.state_dict())
    for k, v in other.state_dict().items():
        assert torch.equal(padded.state_dict()[k], v), k
    assert padded.conv.convs[0].comb_fns[0].weight.shape == (Hp, 2 * Hp)
    w = padded.conv.convs[0].comb_fns[0].weight.detach()
    assert float(w[H:].abs().sum()) == 0.0 and float(w[:, H:Hp].abs().sum()) == 0.0 and float(w[:, Hp + H:].abs().sum()) == 0.0


@pytest.mark.gpu
@pytest.mark.parametrize("hidden,pool,dropout", [(96, "sum", 0.0), (48, "size", 0.0), (100, "mean", 0.0), (160, "max", 0.0)])
def test_padded_width_runs_the_step_program_exactly(hidden, pool, dropout):
    """A width no kernel family serves, built by the drivers' factory: the step program takes it (no per-op fallback), logits
    and every logical gradient match the fp64 oracle on the LOGICAL weights, the gradients on the padding are exactly zero
    and Adam leaves the padding at zero."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import flat_grads, rel_inf, record_parity
    from impl import utils
    from glass_amd import stack, synth
    from glass_amd.arena import ParamArena
    from glass_amd.optim import FlatAdam
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import glass_oracle as O
    dev = "cuda:0"
    n, K, L = 3000, 4, 2
    ei, ew = synth.make_graph(n, 20000, 21, 0.4)
    x = synth.degree_feature(ei, n)
    pos, y = synth.make_subgraphs(n, 30, 12, K, 1, False)
    pos[2, 5:] = -1
    ei, ew, x, pos, y = (torch.from_numpy(a) for a in (ei, ew, x, pos, y))
    torch.manual_seed(hidden)
    model = build_glass(hidden, L, int(x.max()), K, "mean", pool, 0.85, dropout=dropout)
    H, Hp = model._glass_logical_width
    sd = {k: v.clone() for k, v in model.state_dict().items()}  # logical shapes
    model.to(dev).train()
    arena = ParamArena(model)
    assert stack.StackProgram.supported(model.conv)
    xg, eig, ewg, posg, yg = (t.to(dev) for t in (x, ei, ew, pos, y))
    pred = model(xg, eig, ewg, posg, utils.MaxZOZ(xg, posg))
    loss = nn.CrossEntropyLoss()(pred, yg)
    loss.backward()
    orc = O.OracleGLASS(hidden, L, int(x.max()), K, aggr="mean", pool=pool, z_ratio=0.85, gn=True, act="elu")
    orc.load_state_dict(sd)
    orc = orc.double().train()
    po = orc(x, ei, ew.double(), pos, O.max_zero_one(x, pos))
    lo = nn.CrossEntropyLoss()(po, y)
    lo.backward()
    theirs = {k: p.grad for k, p in orc.named_parameters()}
    mine, pad_max = widths.logical_named_grads(model)
    mine = {k: v.cpu() for k, v in mine.items()}
    keys = sorted(mine)
    assert keys == sorted(theirs)
    e_pred, e_loss = rel_inf(pred.detach().cpu(), po.detach()), abs(loss.item() - lo.item()) / abs(lo.item())
    e_grad = rel_inf(flat_grads(mine, keys), flat_grads(theirs, keys))
    record_parity(f"padded_width/N3000_hidden{hidden}_as_{Hp}_{pool}", logits_rel_inf=e_pred, loss_rel=e_loss, grad_rel_inf=e_grad,
                  grad_on_padding=pad_max)
    assert e_pred < 1e-5 and e_loss < 1e-5 and e_grad < 1e-5
    assert pad_max == 0.0
    # three optimizer steps: the padding never moves
    opt = FlatAdam(arena, lr=1e-2)
    for _ in range(3):
        opt.zero_grad()
        nn.CrossEntropyLoss()(model(xg, eig, ewg, posg, utils.MaxZOZ(xg, posg)), yg).backward()
        opt.step()
    after = model.state_dict()
    for name, p in model.named_parameters():
        lg = widths.unpad_tensor(p.detach(), after[name].shape, H, Hp)
        assert float(p.detach().abs().double().sum()) == pytest.approx(float(lg.abs().double().sum()), rel=1e-9), name
    assert any(not torch.equal(after[k].cpu(), sd[k]) for k in sd)  # (and the logical part did train)
