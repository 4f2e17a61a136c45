"""Losses the training step can fuse with the prediction head.

`CrossEntropy()` / `BCEWithLogits()` behave exactly like the losses the reference driver builds
(GLASSTest.py:57-58, 69) when called as `loss_fn(pred, y)`; in addition `glass_amd.step.TrainStep`
recognises them — and any callable that computes one of them, the reference driver's own lambda included (`fusable_mode`) —
and, when the model's head is a bare nn.Linear (GLASSTest.py:159-160), runs head + loss + their backward as two kernels
(`glass_head_loss_fwd/bwd_f32`) instead of ~12 launches."""
import torch
import torch.nn as nn

from . import _lib, ops


class CrossEntropy(nn.Module):
    mode = 0

    def forward(self, pred, y):
        return nn.functional.cross_entropy(pred, y)


class BCEWithLogits(nn.Module):
    """Binary / multi-label: BCEWithLogitsLoss()(pred.flatten(), y.flatten())."""
    mode = 1

    def forward(self, pred, y):
        return nn.functional.binary_cross_entropy_with_logits(pred.flatten(), y.flatten())


import os as _os
PROBE_OPAQUE_LOSSES = _os.environ.get("GLASS_LOSS_PROBE", "1") != "0"   # 0: never evaluate an opaque loss callable to classify it
_PROBE_TOL = 1e-6
_probed = {}  # id(loss_fn) -> (weakref or None, mode or None)


def _probe(loss_fn):
    """Decide by EVALUATION what an opaque callable computes: it is called on small seeded CPU logit / target pairs — value
    and gradient with respect to the logits must equal the fused head's cross-entropy (int64 class targets, mean reduction)
    or BCE-with-logits on the flattened tensors (float targets, mean reduction) to 1e-6 on every pair.  The reference's
    binary loss is such a callable: `lambda x, y: BCEWithLogitsLoss()(x.flatten(), y.flatten())` (GLASSTest.py:57-58).  The
    pairs have different row counts and class counts, every class occurs, so a sum reduction, class weights, label smoothing
    or a non-default ignore_index all fail the comparison.  Anything that raises is simply not fusable."""
    import torch.nn.functional as F
    gen = torch.Generator().manual_seed(20240613)

    def agree(logits, target, ref):
        try:
            x = logits.clone().requires_grad_(True)
            with torch.enable_grad():
                out = loss_fn(x, target)
            if not (isinstance(out, torch.Tensor) and out.dim() == 0 and out.dtype == torch.float32 and out.requires_grad):
                return False
            g, = torch.autograd.grad(out, x)
            xr = logits.clone().requires_grad_(True)
            with torch.enable_grad():
                want = ref(xr, target)
            gr, = torch.autograd.grad(want, xr)
            got, want = float(out.detach()), float(want.detach())
            return got == got and abs(got - want) <= _PROBE_TOL * max(1.0, abs(want)) and float((g - gr).abs().max()) <= _PROBE_TOL
        except Exception:  # noqa: BLE001 — a loss that cannot take these tensors is not one of the two
            return False

    def ce_pairs():
        for b, k in ((5, 3), (8, 6)):
            yield 3.0 * torch.randn(b, k, generator=gen), torch.arange(b) % k

    def bce_pairs():
        for shape, yshape in (((6, 1), (6, )), ((5, 4), (5, 4))):
            yield 3.0 * torch.randn(*shape, generator=gen), (torch.rand(*yshape, generator=gen) > 0.5).float()

    if all(agree(x, y, F.cross_entropy) for x, y in ce_pairs()):
        return 0
    if all(agree(x, y, lambda a, t: F.binary_cross_entropy_with_logits(a.flatten(), t.flatten())) for x, y in bce_pairs()):
        return 1
    return None


def fusable_mode(loss_fn):
    """0 (cross-entropy) / 1 (BCE with logits on the flattened tensors) when the training step can fuse this loss with the
    head, else None.  Recognised by type: this module's marker classes and torch's own CrossEntropyLoss with default settings
    (what the reference driver builds for multi-class sets, GLASSTest.py:69).  Any other callable — the reference's binary
    case is a lambda around BCEWithLogitsLoss (GLASSTest.py:57-58) — is recognised by what it computes (`_probe`, once per
    callable)."""
    if isinstance(loss_fn, (CrossEntropy, BCEWithLogits)):
        return loss_fn.mode
    if (type(loss_fn) is nn.CrossEntropyLoss and loss_fn.weight is None and loss_fn.reduction == "mean" and
            loss_fn.ignore_index == -100 and getattr(loss_fn, "label_smoothing", 0.0) == 0.0):
        return 0
    if not callable(loss_fn):
        return None
    # Opt-out: the probe CALLS the loss (<= 4 times, on small synthetic CPU tensors) — a stateful callable (running statistics,
    # logging, counters) sees those phantom calls, and one that equals CE / BCE at probe time but depends on state that changes
    # later would be replaced by the fused kernel for good.  `loss_fn._glass_no_fuse = True` (or GLASS_LOSS_PROBE=0 in the
    # environment) keeps such a callable on the plain path: it is then called as it is, once per step, by autograd.
    if getattr(loss_fn, "_glass_no_fuse", False) or not PROBE_OPAQUE_LOSSES:
        return None
    hit = _probed.get(id(loss_fn))
    if hit is not None and (hit[0] is None or hit[0]() is loss_fn):
        return hit[1]
    import weakref
    try:
        ref = weakref.ref(loss_fn)
    except TypeError:
        ref = None
    import warnings
    with warnings.catch_warnings():  # (a loss of another kind may warn about the probe's shapes: not the user's business)
        warnings.simplefilter("ignore")
        mode = _probe(loss_fn)
    if ref is not None:  # (no weak reference possible: the id could be reused by another object — probe again next time)
        if len(_probed) > 256:
            _probed.clear()
        _probed[id(loss_fn)] = (ref, mode)
    return mode


class HeadLossFn(torch.autograd.Function):
    """loss = L(pooled @ W^T + b, target) — forward in one launch, backward in one launch."""
    @staticmethod
    def forward(ctx, pooled, weight, bias, target, mode, direct):
        ops._need_gpu(pooled, weight, target)
        pooled, ldp = ops._rows(pooled)
        B, C = pooled.shape
        K = weight.shape[0]
        w, b = weight.contiguous(), bias.contiguous()
        tgt = target.contiguous().to(torch.int64 if mode == 0 else torch.float32)
        logits = torch.empty((B, K), dtype=torch.float32, device=pooled.device)
        prob = torch.empty(B * K + B, dtype=torch.float32, device=pooled.device)  # probabilities + per-row loss terms
        loss = torch.empty((), dtype=torch.float32, device=pooled.device)
        rc = _lib.load().glass_head_loss_fwd_f32(pooled.data_ptr(), ldp, w.data_ptr(), b.data_ptr(), tgt.data_ptr(), mode,
                                                 B, C, K, logits.data_ptr(), prob.data_ptr(), loss.data_ptr(),
                                                 ops._stream())
        _lib.check(rc, "glass_head_loss_fwd_f32")
        ctx.save_for_backward(pooled, w, prob, tgt)
        ctx.cfg = (mode, B, C, K)
        # direct: accumulate dW / db straight into the flat gradient arena (weight.grad / bias.grad views)
        ctx.direct = (weight, bias) if (direct and weight.grad is not None and bias.grad is not None) else None
        ctx.mark_non_differentiable(logits)
        ctx.set_materialize_grads(False)  # no zero-filled gradient tensor for the logits output
        return loss, logits

    @staticmethod
    def backward(ctx, gl, _glogits):
        pooled, w, prob, tgt = ctx.saved_tensors
        mode, B, C, K = ctx.cfg
        gl = gl.contiguous().reshape(1).to(torch.float32)
        dpooled = torch.empty((B, C), dtype=torch.float32, device=pooled.device)
        if ctx.direct is not None:
            dW, db, acc = ctx.direct[0].grad, ctx.direct[1].grad, 1
        else:
            dW, db, acc = torch.empty_like(w), torch.empty(K, dtype=torch.float32, device=w.device), 0
        rc = _lib.load().glass_head_loss_bwd_f32(pooled.data_ptr(), pooled.stride(0), w.data_ptr(), prob.data_ptr(),
                                                 tgt.data_ptr(), mode, gl.data_ptr(), B, C, K, dpooled.data_ptr(), C,
                                                 dW.data_ptr(), db.data_ptr(), acc, ops._stream())
        _lib.check(rc, "glass_head_loss_bwd_f32")
        if ctx.direct is not None:
            return dpooled, None, None, None, None, None
        return dpooled, dW, db, None, None, None


def head_loss(pooled, linear, target, mode, direct=False):
    """-> (loss, logits) for an nn.Linear head; mode 0 = cross-entropy, 1 = BCE-with-logits."""
    return HeadLossFn.apply(pooled, linear.weight, linear.bias, target, mode, direct)
