"""Edge semantics of the two product forms of the tiled dense family (hidden 256, comb pair, no activation, zero bias):
non-finite operands and operands scaled towards the bottom of fp32's range, forward and data gradient, against fp64."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import torch.nn as nn

from glass_amd import ops
from helpers import rel_inf
from test_gpu_kernels import _pack

DEV = "cuda:0"
H, N, zr = 256, 2100, 0.8
gen = torch.Generator().manual_seed(5)
W = torch.randn(2 * H, 2 * H, generator=gen) / (2 * H)**0.5
b = torch.zeros(2 * H)
xa_h, xb_h = torch.randn(N, H, generator=gen), torch.randn(N, H, generator=gen)
mask = (torch.rand(N, generator=gen) < 0.05)


def run(scale_log2, form_f32, inf_at=None):
    ops.DENSE_F32_PRODUCTS = form_f32
    sc = 2.0 ** (-scale_log2)
    Wg, bg = W.to(DEV), b.to(DEV)
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
    g = torch.ones_like(out) * sc
    out.backward(g)
    return out.detach().cpu(), xa.grad.cpu()


from oracle import glass_oracle as O
for s in (0, 60, 90, 100, 105, 110, 115, 120, 125):
    sc = 2.0 ** (-s)
    xa64, xb64 = (xa_h.double() * sc).requires_grad_(True), (xb_h.double() * sc)
    Z = torch.cat((xa64, xb64), -1) @ W.double().t()
    ref = O._mix(mask.reshape(-1, 1), zr, Z[:, :H], Z[:, H:])
    ref.backward(torch.ones_like(ref) * sc)
    row = [s]
    for f32 in (True, False):
        out, gx = run(s, f32)
        row += [rel_inf(out, ref.detach()), rel_inf(gx, xa64.grad)]
    print("scale 2^-%d  f32-form out %.2e dx %.2e | split-form out %.2e dx %.2e" % tuple(row))
for f32 in (True, False):
    out, gx = run(0, f32, inf_at=(5, 7))
    clean, _ = run(0, f32)
    others = torch.ones(N, dtype=torch.bool)
    others[5] = False
    print("Inf operand, form", "f32" if f32 else "split", ": row 5 finite share %.2f, has_inf %s has_nan %s; other rows equal to the clean run: %s"
          % (torch.isfinite(out[5]).float().mean().item(), bool(torch.isinf(out[5]).any()), bool(torch.isnan(out[5]).any()),
             bool(torch.equal(out[others], clean[others]))))
