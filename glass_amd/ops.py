"""torch.autograd bindings of the HIP kernels (host side of the C ABI in include/glass_hip.h).

Every op here enqueues hand-written gfx950 kernels on torch's current HIP stream through ctypes;
torch only supplies device memory, streams and the autograd tape.  There is no CPU path: a CPU
tensor raises GlassHipError (the CPU restatement lives in oracle/ and is a checker only).
"""
import os

import torch

from . import _lib
from ._lib import ACT_NONE, ACT_ELU, ACT_RELU, POOL_MODES, GlassHipError


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _need_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise GlassHipError(f"glass_amd op got a {t.device} tensor: the HIP path is the only path (no CPU fallback)")


def _rows(t):
    """Row-major 2-D fp32 view -> (tensor, leading dimension)."""
    if t.dim() != 2 or t.dtype != torch.float32:
        raise GlassHipError(f"expected a 2-D float32 tensor, got {tuple(t.shape)} {t.dtype}")
    if t.stride(1) != 1 or (t.shape[0] > 1 and t.stride(0) < t.shape[1]):
        t = t.contiguous()
    return t, (t.stride(0) if t.shape[0] > 1 else max(t.shape[1], t.stride(0)))


# ---------------------------------------------------------------------------------------------
# dropout RNG state: uint64[2] = (seed, step) in DEVICE memory, so captured graphs get fresh masks
# ---------------------------------------------------------------------------------------------
_rng = {}


def _dev(device):
    """One key per device however it is spelled ("cuda:0", torch.device("cuda", 0), torch.device("cuda"))."""
    d = torch.device(device)
    if d.type == "cuda" and d.index is None:
        d = torch.device("cuda", torch.cuda.current_device())
    return d


def rng_state(device):
    device = _dev(device)
    st = _rng.get(device)
    if st is None:
        st = torch.tensor([torch.initial_seed() & 0x7FFFFFFFFFFFFFFF, 0], dtype=torch.int64, device=device)
        _rng[device] = st
    return st


def rng_seed(seed, device):
    """(Re)seed the dropout stream IN PLACE: captured graphs keep reading the same device words."""
    device = _dev(device)
    st = _rng.get(device)
    new = torch.tensor([int(seed) & 0x7FFFFFFFFFFFFFFF, 0], dtype=torch.int64)
    if st is None:
        _rng[device] = new.to(device)
    else:
        st.copy_(new)


# Host-side count of advances of the device-resident stream.  The backward of a dropout regenerates the forward's
# mask from the LIVE (seed, step) words, so a second training forward (which advances them) between a forward and its
# backward would silently pair the gradient with the wrong mask (gradient accumulation, GLASS.NodeEmb over several
# feature channels, two model calls before loss.backward()).  Every autograd node that drew a mask remembers the
# epoch of its forward and refuses to run its backward under another one.  (Replays of a captured step do not touch
# the host counter; there forward and backward sit in the same graph.)
_rng_epoch = {}


def note_rng_advance(device):
    device = _dev(device)
    _rng_epoch[device] = _rng_epoch.get(device, 0) + 1


def rng_epoch(device):
    return _rng_epoch.get(_dev(device), 0)


def check_rng_epoch(device, epoch, what):
    if epoch != rng_epoch(device):
        raise RuntimeError(f"{what}: the dropout stream was advanced by another training forward between this "
                           "forward and its backward; its mask can no longer be regenerated (run backward() before "
                           "the next training forward, or use dropout 0)")


# ---- the dropout words of the CURRENT pass -------------------------------------------------------------------------
# A training forward on the autograd paths takes a private 16-byte snapshot of (seed, step) right after advancing the
# stream; every kernel of that forward AND of its backward reads the snapshot, so any number of training forwards may lie
# between a forward and its backward (GLASS.NodeEmb over several feature channels, reference impl/models.py:336-344;
# gradient accumulation; two model calls before loss.backward()).  The tape-free step program (stack.loss_and_grads)
# runs forward and backward back to back and reads the live words — no copy launch in the replayed step.
_rng_cur = {}


def rng_tensor(device):
    """The (seed, step) words the kernels launched NOW read: the current pass's snapshot, else the live stream."""
    t = _rng_cur.get(_dev(device))
    return t if t is not None else rng_state(device)


def rng_snapshot(device):
    """A private copy of the live words (one small device copy; a captured graph replays it with the step)."""
    return rng_state(device).clone()


class rng_scope:
    """`with rng_scope(device, words):` — the kernels launched inside read `words` (None: the live stream)."""
    def __init__(self, device, words):
        self.device, self.words = _dev(device), words

    def __enter__(self):
        self.prev = _rng_cur.get(self.device)
        _rng_cur[self.device] = self.words
        return self.words

    def __exit__(self, *exc):
        _rng_cur[self.device] = self.prev
        return False


def rng_advance(device):
    _lib.check(_lib.load().glass_rng_advance(rng_state(device).data_ptr(), _stream()), "glass_rng_advance")
    note_rng_advance(device)


# ---------------------------------------------------------------------------------------------
# K4  MaxZOZ
# ---------------------------------------------------------------------------------------------
def maxzoz(n_nodes, pos):
    """z[n] = 1 iff n occurs in pos (int64, -1 = padding).  reference impl/utils.py:32-45"""
    _need_gpu(pos)
    pos = pos.contiguous()
    if pos.dtype != torch.int64:
        pos = pos.to(torch.int64)
    z = torch.empty(n_nodes, dtype=torch.int64, device=pos.device)
    _lib.check(_lib.load().glass_maxzoz_i64(pos.data_ptr(), pos.numel(), z.data_ptr(), n_nodes, _stream()),
               "glass_maxzoz_i64")
    return z


# ---------------------------------------------------------------------------------------------
# K1  aggregation
# ---------------------------------------------------------------------------------------------
class SpMMFn(torch.autograd.Function):
    """y = A @ x with A a graph.CSRAdj; backward dx = A^T @ dy (A carries no gradient:
    edge weights never require grad in the reference, datasets.py:126,226)."""
    @staticmethod
    def forward(ctx, adj, x):
        _need_gpu(x)
        x, _ = _rows(x)
        ctx.adj = adj
        return adj.fwd.spmm(x)

    @staticmethod
    def backward(ctx, dy):
        dy, _ = _rows(dy)
        return None, ctx.adj.bwd.spmm(dy)


def spmm(adj, x):
    return SpMMFn.apply(adj, x)


# ---------------------------------------------------------------------------------------------
# K3+K4  label + embedding
# ---------------------------------------------------------------------------------------------
class EmbedLabelFn(torch.autograd.Function):
    """(h, mask) = (W[x], label byte).  Backward dW = S^T @ dh on K1 via graph.Selection."""
    @staticmethod
    def forward(ctx, weight, x_flat, z, selection):
        _need_gpu(weight, x_flat, z)
        n, (V, H) = x_flat.shape[0], weight.shape
        w = weight.contiguous()
        out = torch.empty((n, H), dtype=torch.float32, device=w.device)
        mask = torch.empty(n, dtype=torch.uint8, device=w.device)
        zp = 0 if z is None else z.data_ptr()
        rc = _lib.load().glass_embed_label_f32(x_flat.data_ptr(), w.data_ptr(), V, zp, 0, 0, out.data_ptr(), H,
                                               mask.data_ptr(), n, H, _stream())
        _lib.check(rc, "glass_embed_label_f32")
        ctx.selection = selection
        ctx.mark_non_differentiable(mask)
        ctx.set_materialize_grads(False)  # no zero-filled gradient tensor for the mask output
        return out, mask

    @staticmethod
    def backward(ctx, dout, _dmask):
        dout, _ = _rows(dout)
        return ctx.selection.op.spmm(dout), None, None, None


def embed_label(weight, x_flat, z, selection):
    return EmbedLabelFn.apply(weight, x_flat, z, selection)


# ---------------------------------------------------------------------------------------------
# mix
# ---------------------------------------------------------------------------------------------
class MixFn(torch.autograd.Function):
    """out = mask ? zr*a1+(1-zr)*a0 : zr*a0+(1-zr)*a1 with a = act(T), T = [T1 | T0]  [N,2H]."""
    @staticmethod
    def forward(ctx, T, mask, z_ratio, act):
        _need_gpu(T, mask)
        T, ldt = _rows(T)
        n, H = T.shape[0], T.shape[1] // 2
        out = torch.empty((n, H), dtype=torch.float32, device=T.device)
        rc = _lib.load().glass_mix_fwd_f32(T.data_ptr(), ldt, mask.data_ptr(), float(z_ratio), act, out.data_ptr(), H,
                                           n, H, _stream())
        _lib.check(rc, "glass_mix_fwd_f32")
        ctx.save_for_backward(T if act != ACT_NONE else None, mask)
        ctx.z_ratio, ctx.act, ctx.shape = float(z_ratio), act, (n, H)
        return out

    @staticmethod
    def backward(ctx, dout):
        T, mask = ctx.saved_tensors
        n, H = ctx.shape
        dout, ldd = _rows(dout)
        dT = torch.empty((n, 2 * H), dtype=torch.float32, device=dout.device)
        tp, ldt = (0, 0) if T is None else (T.data_ptr(), T.stride(0))
        rc = _lib.load().glass_mix_bwd_f32(dout.data_ptr(), ldd, tp, ldt, mask.data_ptr(), ctx.z_ratio, ctx.act,
                                           dT.data_ptr(), 2 * H, n, H, _stream())
        _lib.check(rc, "glass_mix_bwd_f32")
        return dT, None, None, None


def mix(T, mask, z_ratio, act):
    return MixFn.apply(T, mask, z_ratio, act)


# ---------------------------------------------------------------------------------------------
# K5  the two Linears of a weight-set pair as one GEMM; weight gradient on the fp32 matrix cores
# ---------------------------------------------------------------------------------------------
import os as _os

# (Weight gradients on a side stream beside the backward chain were measured on MI355X at ppi_bp-shape, hipGraph replay,
# 3 interleaved A/B rounds: 0.765 ms/step forked vs 0.678 on one stream — the fork/join dependencies cost more than the
# overlap of these 10-20 us kernels buys.  The variant is not part of the product; DESIGN.md §7 keeps the record.)
USE_FUSED_DENSE = _os.environ.get("GLASS_FUSED_DENSE", "1") != "0"  # A/B switch: fused MFMA dense path
# Product form of the LDS-tiled dense kernels (hidden 128 / 256 / 512): the default is the split-bf16 form; GLASS_DENSE_SPLIT=0
# (or setting this flag) asks every dense call for the f32-input MFMA instead — an option of each CALL (the library itself
# keeps no such state: include/glass_hip.h GLASS_DENSE_F32_PRODUCTS)
DENSE_F32_PRODUCTS = _os.environ.get("GLASS_DENSE_SPLIT", "1") == "0"


def act_word(act):
    """The `act` argument of the dense entries: activation code + this call's options."""
    return int(act) | (_lib.DENSE_F32_PRODUCTS if DENSE_F32_PRODUCTS else 0)
_wgrad_ws = {}
_retired_ws = []


def _wgrad_workspace(device, N, O, I, slot=0, min_bytes=0):
    nbytes = max(_lib.load().glass_linear_wgrad_ws_bytes(N, O, I), min_bytes)
    ws = _wgrad_ws.get((device, slot))
    if ws is None or ws.numel() * 4 < nbytes:
        if ws is not None:
            _retired_ws.append(ws)  # a captured graph may still launch with this pointer: never hand it back
        ws = torch.empty(nbytes // 4 + 16, dtype=torch.float32, device=device)
        _wgrad_ws[(device, slot)] = ws
    return ws


def linear_wgrad(G, X, dW, db, accumulate, slot=0):
    """dW (+)= G^T @ X, db (+)= colsum(G) with glass_linear_wgrad_f32; False if the shape is unsupported.
    `slot` selects a scratch buffer (calls that may overlap on different streams need different ones)."""
    G, ldg = _rows(G)
    X, ldx = _rows(X)
    N, O = G.shape
    I = X.shape[1]
    thin = O <= 3 and I % 4 == 0 and ldx % 4 == 0 and X.data_ptr() % 16 == 0  # hidden -> 1 heads: the library's thin kernels
    if dW.stride(1) != 1 or not (thin or not (O % 4 or I % 2 or ldg % 4 or ldx % 2 or G.data_ptr() % 16 or X.data_ptr() % 8)):
        return False
    ws = _wgrad_workspace(G.device, N, O, I, slot)
    rc = _lib.load().glass_linear_wgrad_f32(G.data_ptr(), ldg, X.data_ptr(), ldx, N, O, I, dW.data_ptr(), dW.stride(0),
                                            0 if db is None else db.data_ptr(), int(accumulate), ws.data_ptr(),
                                            _stream())
    _lib.check(rc, "glass_linear_wgrad_f32")
    return True


class LinearFn(torch.autograd.Function):
    """y = x @ W^T + b for a plain nn.Linear whose weight gradient reduces over MANY rows into a tiny [O, I] output (the
    SSL pre-training path's MyGCNConv / MLP layers over all nodes or a 131 072-edge batch): forward and dx are library
    GEMMs through torch, dW / db the split-K MFMA kernel (glass_linear_wgrad_f32) — the library takes 160-290 us per
    such weight gradient at ppi_bp-shape (rocprofv3: 44 % of a pre-training step), the kernel 20-40."""
    @staticmethod
    def forward(ctx, x, W, b):
        ctx.save_for_backward(x, W)
        ctx.has_bias = b is not None
        return torch.addmm(b, x, W.t()) if b is not None else x @ W.t()

    @staticmethod
    def backward(ctx, dy):
        x, W = ctx.saved_tensors
        if not ctx.needs_input_grad[0]:
            dx = None
        elif W.shape[0] == 1:
            dx = dy * W  # one output: an outer product (bitwise what the K = 1 GEMM gives), no library call
        else:
            dx = dy @ W
        dW = db = None
        if ctx.needs_input_grad[1]:
            dyc, xc = dy.contiguous(), x.contiguous()
            dW = torch.empty_like(W)
            db = torch.empty(W.shape[0], dtype=W.dtype, device=W.device) if ctx.has_bias else None
            if not (dyc.is_cuda and linear_wgrad(dyc, xc, dW, db, accumulate=False, slot=("linear", W.shape))):
                dW = dyc.t() @ xc
                db = dyc.sum(0) if ctx.has_bias else None
        elif ctx.has_bias and ctx.needs_input_grad[2]:
            db = dy.sum(0)
        return dx, dW, db


def linear(x, lin):
    """nn.Linear forward through LinearFn (2-D CUDA fp32 inputs; anything else: torch's own)."""
    if x.dim() == 2 and x.is_cuda and x.dtype == torch.float32 and lin.weight.dtype == torch.float32:
        return LinearFn.apply(x, lin.weight, lin.bias)
    return torch.nn.functional.linear(x, lin.weight, lin.bias)


def _wgrad_supported(G, X, dW):
    O, I = G.shape[1], X.shape[1]
    return not (O % 4 or I % 2 or G.stride(0) % 4 or X.stride(0) % 2 or G.data_ptr() % 16 or X.data_ptr() % 8 or
                G.stride(1) != 1 or X.stride(1) != 1 or dW.stride(1) != 1)


def _arena_grads_live(stack, params):
    """The in-place gradient path writes into the arena views stack[2] / stack[3] and hands autograd None.  That is only
    correct while every parameter's .grad still IS its arena view: optimizer.zero_grad(set_to_none=True) (torch's
    default) detaches them, after which gradients written to the arena would never reach the optimizer."""
    base = stack[2].untyped_storage().data_ptr()
    return all(p.grad is not None and p.grad.untyped_storage().data_ptr() == base for p in params)


class StackedLinearFn(torch.autograd.Function):
    """T = x @ [W1; W0]^T + [b1 | b0]  — both weight sets of a GLASSConv Linear pair in one GEMM
    (rocBLAS/hipBLASLt through torch).  With a ParamArena the stacked weight is a view (no cat) and
    the backward accumulates dW / db straight into the gradient arena with the split-K MFMA kernel."""
    @staticmethod
    def forward(ctx, x, w1, w0, b1, b0, stack):
        _need_gpu(x, w1)
        if stack is not None:
            W, b = stack[0], stack[1]
        else:
            W, b = torch.cat((w1, w0)), torch.cat((b1, b0))
        ctx.save_for_backward(x, W)
        ctx.stack = stack
        ctx.params = (w1, w0, b1, b0)
        ctx.split = w1.shape[0]
        return torch.addmm(b, x, W.t())

    @staticmethod
    def backward(ctx, dT):
        x, W = ctx.saved_tensors
        dT, _ = _rows(dT)
        if ctx.stack is not None and not _arena_grads_live(ctx.stack, ctx.params):
            ctx.stack = None  # .grad no longer aliases the arena: gradients go back through autograd
        dx = torch.mm(dT, W) if ctx.needs_input_grad[0] else None
        if ctx.stack is not None and _wgrad_supported(dT, x, ctx.stack[2]) and ctx.stack[2].is_contiguous():
            # arena views: dW / db accumulate straight into the gradient arena (no temporaries, no autograd add kernels)
            if linear_wgrad(dT, x, ctx.stack[2], ctx.stack[3], True):
                return dx, None, None, None, None, None
        dW = torch.empty_like(W)
        db = torch.empty(W.shape[0], dtype=W.dtype, device=W.device)
        if not linear_wgrad(dT, x, dW, db, False):
            dW = torch.mm(dT.t(), x)
            db = dT.sum(0)
        k = ctx.split
        return dx, dW[:k], dW[k:], db[:k], db[k:], None


def stacked_linear(x, lin1, lin0, stack=None):
    return StackedLinearFn.apply(x, lin1.weight, lin0.weight, lin1.bias, lin0.bias, stack)


class JoinColsFn(torch.autograd.Function):
    """`buf` already holds the column blocks `parts` (each op wrote its output straight into its slice of
    the JK buffer); this only tells autograd that buf depends on them — no copy in either direction
    (replaces torch.cat((xs...), -1) of reference impl/models.py:263)."""
    @staticmethod
    def forward(ctx, buf, *parts):
        off = 0
        for p in parts:
            if p.data_ptr() != buf.data_ptr() + 4 * off or p.stride(0) != buf.stride(0) or p.shape[0] != buf.shape[0]:
                raise GlassHipError("JoinColsFn: parts must be consecutive column slices of buf")
            off += p.shape[1]
        ctx.widths = [p.shape[1] for p in parts]
        return buf.view_as(buf)

    @staticmethod
    def backward(ctx, g):
        outs, off = [], 0
        for w in ctx.widths:
            outs.append(g[:, off:off + w])
            off += w
        return (None, *outs)


def join_cols(buf, parts):
    return JoinColsFn.apply(buf, *parts)


# A/B switches of the fused dense path per hidden size (the library itself keeps no state): GLASS_DENSE_H128=0 /
# GLASS_DENSE_H256=0 send that width back to library GEMMs + stand-alone mix kernels.
_DENSE_OFF = {h for h in (128, 256, 512) if os.environ.get(f"GLASS_DENSE_H{h}", "1") == "0"}


def dual_linear_supported(H):
    return int(H) not in _DENSE_OFF and bool(_lib.load().glass_dual_linear_supported(int(H)))


class DualLinearMixFn(torch.autograd.Function):
    """out = mix(act(Z1), act(Z0)) with Z = [xa || xb] @ [W1;W0]^T + [b1|b0] in ONE kernel on the fp32 matrix
    cores (glass_dual_linear_fwd_f32); xb=None for the trans pair (ELU, Z kept for the backward), xb=x_ for
    the comb pair (no activation, no cat, Z never materialised).  Backward: one fused data-gradient kernel
    and the split-K weight-gradient kernel, both synthesising dZ from `dout` on the fly; weight / bias
    gradients are accumulated straight into the gradient arena (stack = arena views W, b, dW, db + the packed
    operand images of W and W^T)."""
    @staticmethod
    def forward(ctx, xa, xb, w1, w0, b1, b0, mask, z_ratio, act, stack, out):
        _need_gpu(xa, mask)
        xa, lda = _rows(xa)
        n, H = xa.shape
        if xb is not None:
            xb, ldb = _rows(xb)
        Wimg, b = stack[4], stack[1]
        T = torch.empty((n, 2 * H), dtype=torch.float32, device=xa.device) if act != ACT_NONE else None
        if out is None:
            out = torch.empty((n, H), dtype=torch.float32, device=xa.device)
        rc = _lib.load().glass_dual_linear_fwd_f32(xa.data_ptr(), lda, 0 if xb is None else xb.data_ptr(),
                                                   0 if xb is None else ldb, Wimg.data_ptr(), b.data_ptr(),
                                                   mask.data_ptr(), float(z_ratio), act_word(act), 0 if T is None else T.data_ptr(),
                                                   2 * H, out.data_ptr(), out.stride(0), n, H, 0, 0, 0, 0, 0, 0.0, 0, 0, 0, 0,
                                                   0, 0, _stream())
        _lib.check(rc, "glass_dual_linear_fwd_f32")
        ctx.save_for_backward(xa, xb, T, mask)
        ctx.cfg = (float(z_ratio), act, stack, n, H)
        ctx.params = (w1, w0, b1, b0)
        return out

    @staticmethod
    def backward(ctx, dout):
        xa, xb, T, mask = ctx.saved_tensors
        z_ratio, act, stack, n, H = ctx.cfg
        dout, ldd = _rows(dout)
        n_out = H if xb is None else 2 * H
        lib = _lib.load()
        tp, ldt = (0, 0) if T is None else (T.data_ptr(), T.stride(0))
        din = None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            din = torch.empty((n, n_out), dtype=torch.float32, device=dout.device)
            rc = lib.glass_dual_linear_dgrad_f32(dout.data_ptr(), ldd, tp, ldt, mask.data_ptr(), z_ratio, act_word(act),
                                                 stack[5].data_ptr(), n_out, 0, 0, 0.0, 0, 0, din.data_ptr(), n_out, n, H,
                                                 0, 0, 0, 0, 0, 0, 0.0, 0, 0, _stream())
            _lib.check(rc, "glass_dual_linear_dgrad_f32")
        I = n_out
        ws = _wgrad_workspace(dout.device, n, 2 * H, I)
        live = _arena_grads_live(stack, ctx.params)
        if live:   # accumulate straight into the gradient arena
            dW, dbias, accumulate = stack[2], stack[3], 1
        else:      # .grad was detached from the arena (zero_grad(set_to_none=True)): hand the gradients to autograd
            dW = torch.empty((2 * H, I), dtype=torch.float32, device=dout.device)
            dbias = torch.empty(2 * H, dtype=torch.float32, device=dout.device)
            accumulate = 0
        rc = lib.glass_dual_linear_wgrad_f32(dout.data_ptr(), ldd, tp, ldt, mask.data_ptr(), z_ratio, act_word(act), xa.data_ptr(),
                                             xa.stride(0), 0 if xb is None else xb.data_ptr(),
                                             0 if xb is None else xb.stride(0), n, H, dW.data_ptr(),
                                             dW.stride(0), dbias.data_ptr(), accumulate, ws.data_ptr(), _stream())
        _lib.check(rc, "glass_dual_linear_wgrad_f32")
        da = din if xb is None else (None if din is None else din[:, :H])
        db_ = None if (xb is None or din is None) else din[:, H:]
        if live:
            return da, db_, None, None, None, None, None, None, None, None, None
        return da, db_, dW[:H], dW[H:], dbias[:H], dbias[H:], None, None, None, None, None


def dual_linear_mix(xa, xb, lin1, lin0, mask, z_ratio, act, stack, out=None):
    return DualLinearMixFn.apply(xa, xb, lin1.weight, lin0.weight, lin1.bias, lin0.bias, mask, z_ratio, act, stack, out)


# ---------------------------------------------------------------------------------------------
# K6  GraphNorm (+ELU +dropout)
# ---------------------------------------------------------------------------------------------
_gn_ws = {}


def _graphnorm_ws(device, n_rows, C):
    """Per-workgroup partial sums of the stand-alone GraphNorm kernels: one buffer per width and per BRANCH
    (graph.set_workspace_branch: the parallel evaluation branches of evalstep.EvalGraph run the same kernels concurrently
    and must not share it).  A buffer that has to grow is retired, not freed (a captured graph may hold its pointer)."""
    from . import graph as ggraph
    key = (device, C, ggraph._ws_branch)
    nbytes = _lib.load().glass_graphnorm_ws_bytes(n_rows, C)
    ws = _gn_ws.get(key)
    if ws is None or ws.numel() * 8 < nbytes:
        if ws is not None:
            _retired_ws.append(ws)
        ws = torch.empty(nbytes // 8 + 1, dtype=torch.float64, device=device)
        _gn_ws[key] = ws
    return ws


class GraphNormFn(torch.autograd.Function):
    """y = dropout(act(GraphNorm(x))) over the whole graph (PyG GraphNorm with batch=None)."""
    @staticmethod
    def forward(ctx, x, gamma, beta, alpha, eps, act, p_drop, call_id, direct=False):
        _need_gpu(x, gamma)
        x, ldx = _rows(x)
        n, C = x.shape
        y = torch.empty((n, C), dtype=torch.float32, device=x.device)
        saved = torch.empty(4 * C, dtype=torch.float32, device=x.device)
        ws = _graphnorm_ws(x.device, n, C)
        words = rng_tensor(x.device) if p_drop > 0 else None  # this pass's (seed, step): the backward re-reads THESE
        rng = words.data_ptr() if words is not None else 0
        g, b, a = gamma.contiguous(), beta.contiguous(), alpha.contiguous()
        rc = _lib.load().glass_graphnorm_fwd_f32(x.data_ptr(), ldx, y.data_ptr(), C, n, C, g.data_ptr(), b.data_ptr(),
                                                 a.data_ptr(), eps, saved.data_ptr(), act, p_drop, rng, call_id,
                                                 ws.data_ptr(), _stream())
        _lib.check(rc, "glass_graphnorm_fwd_f32")
        ctx.save_for_backward(x, g, a, saved)
        ctx.cfg = (act, p_drop, call_id)
        ctx.rng_words = words
        ctx.rng_epoch = rng_epoch(x.device)
        # direct: parameter gradients are accumulated straight into the flat gradient arena
        ctx.direct = (gamma, beta, alpha) if (direct and all(t.grad is not None for t in (gamma, beta, alpha))) else None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, g, a, saved = ctx.saved_tensors
        act, p_drop, call_id = ctx.cfg
        n, C = x.shape
        dy, lddy = _rows(dy)
        dx = torch.empty((n, C), dtype=torch.float32, device=x.device)
        if ctx.direct is not None:
            dg, db, da = (t.grad for t in ctx.direct)
            accumulate = 1
        else:
            dparams = torch.empty((3, C), dtype=torch.float32, device=x.device)
            dg, db, da = dparams[0], dparams[1], dparams[2]
            accumulate = 0
        ws = _graphnorm_ws(x.device, n, C)
        if p_drop > 0 and ctx.rng_words is rng_state(x.device):  # no snapshot was taken: the live words must be unchanged
            check_rng_epoch(x.device, ctx.rng_epoch, "GraphNormFn.backward")
        rng = ctx.rng_words.data_ptr() if p_drop > 0 else 0
        rc = _lib.load().glass_graphnorm_bwd_f32(dy.data_ptr(), lddy, x.data_ptr(), x.stride(0), dx.data_ptr(), C, 0, 0,
                                                 n, C, g.data_ptr(), a.data_ptr(), saved.data_ptr(), dg.data_ptr(),
                                                 db.data_ptr(), da.data_ptr(), accumulate, act, p_drop, rng, call_id,
                                                 ws.data_ptr(), _stream())
        _lib.check(rc, "glass_graphnorm_bwd_f32")
        if ctx.direct is not None:
            return dx, None, None, None, None, None, None, None, None
        return dx, dg, db, da, None, None, None, None, None


def graphnorm(x, gamma, beta, alpha, eps=1e-5, act=ACT_NONE, p_drop=0.0, call_id=0, direct=False):
    return GraphNormFn.apply(x, gamma, beta, alpha, float(eps), int(act), float(p_drop), int(call_id), bool(direct))


# ---------------------------------------------------------------------------------------------
# K7  subgraph pooling
# ---------------------------------------------------------------------------------------------
# Largest padded node matrices (B * Smax entries) the ORDERED, atomic-free scatters stage in LDS (pool.hip kPoolOrderedMax,
# readout.hip kReadoutOrderedMax).  Beyond them the pool backward and the fused readout bucket the entries by node and sum
# in exact fixed point (bucket.h: still no float atomic), and so does max pooling's backward.  The C ABI's one float-atomic
# scatter (glass_segment_pool_bwd_atomic_f32) is not used by this module.
POOL_ORDERED_MAX, READOUT_ORDERED_MAX = 12288, 16384


_scratch_bufs = {}


_retired_scratch = []


def _scratch(key, device, nbytes):
    """Uninitialised device scratch of at least `nbytes`, one buffer per key (kept for the process's life).  A buffer that
    has to grow is RETIRED, not freed: a captured hipGraph may still launch with its pointer (as _wgrad_workspace does)."""
    buf = _scratch_bufs.get((device, key))
    if buf is None or buf.numel() < nbytes:
        if buf is not None:
            _retired_scratch.append(buf)
        buf = _scratch_bufs[(device, key)] = torch.empty(int(nbytes) + 16, dtype=torch.uint8, device=device)
    return buf


class SegmentPoolFn(torch.autograd.Function):
    """out[b] = reduce over the non-padding nodes of pos[b] of emb[node]  (sum|mean|max|size)."""
    @staticmethod
    def forward(ctx, emb, pos, mode):
        _need_gpu(emb, pos)
        if mode not in POOL_MODES:
            raise NotImplementedError  # reference: GLASSTest.py:168-171
        emb, lde = _rows(emb)
        pos = pos.contiguous()
        if pos.dtype != torch.int64:
            pos = pos.to(torch.int64)
        n, C = emb.shape
        B, Smax = pos.shape
        out = torch.empty((B, C), dtype=torch.float32, device=emb.device)
        if Smax == 2 and mode != "max":  # node pairs (link-prediction batches): lane groups per pair, not a workgroup
            rc = _lib.load().glass_pair_pool_f32(emb.data_ptr(), lde, pos.data_ptr(), B, POOL_MODES[mode], out.data_ptr(), C,
                                                 n, C, _stream())
            _lib.check(rc, "glass_pair_pool_f32")
            ctx.save_for_backward(pos, None)
            ctx.cfg = (mode, n, C)
            return out
        argmax = torch.empty((B, C), dtype=torch.int32, device=emb.device) if mode == "max" else None
        rc = _lib.load().glass_segment_pool_f32(emb.data_ptr(), lde, pos.data_ptr(), B, Smax, POOL_MODES[mode],
                                                out.data_ptr(), C, 0 if argmax is None else argmax.data_ptr(), n, C,
                                                _stream())
        _lib.check(rc, "glass_segment_pool_f32")
        ctx.save_for_backward(pos, argmax)
        ctx.cfg = (mode, n, C)
        return out

    @staticmethod
    def backward(ctx, dout):
        pos, argmax = ctx.saved_tensors
        mode, n, C = ctx.cfg
        dout, ldd = _rows(dout)
        B, Smax = pos.shape
        if Smax == 2 and mode != "max":  # exact, atomic-free backward; writes every row
            lib = _lib.load()
            demb = torch.empty((n, C), dtype=torch.float32, device=dout.device)
            ws = _scratch(("pair_pool", n, B), dout.device, lib.glass_pair_pool_ws_bytes(n, B))
            rc = lib.glass_pair_pool_bwd_f32(dout.data_ptr(), ldd, pos.data_ptr(), B, POOL_MODES[mode], demb.data_ptr(), C, n,
                                             C, ws.data_ptr(), _stream())
            _lib.check(rc, "glass_pair_pool_bwd_f32")
            return demb, None, None
        if mode != "max" and B * Smax + B > POOL_ORDERED_MAX:  # beyond the LDS staging: bucketed by node, exact sums (no float atomics)
            lib = _lib.load()
            demb = torch.empty((n, C), dtype=torch.float32, device=dout.device)
            ws = _scratch(("pool_exact", n, B, Smax), dout.device, lib.glass_segment_pool_bwd_exact_ws_bytes(n, B, Smax))
            rc = lib.glass_segment_pool_bwd_exact_f32(dout.data_ptr(), ldd, pos.data_ptr(), B, Smax, POOL_MODES[mode],
                                                      demb.data_ptr(), C, n, C, ws.data_ptr(), _stream())
            _lib.check(rc, "glass_segment_pool_bwd_exact_f32")
            return demb, None, None
        if mode == "max":  # exact too: the node's subgraph list (deduplicated per row), gradients of the columns it won
            lib = _lib.load()
            demb = torch.empty((n, C), dtype=torch.float32, device=dout.device)
            ws = _scratch(("pool_exact_max", n, B, Smax), dout.device, lib.glass_segment_pool_bwd_exact_ws_bytes(n, B, Smax))
            rc = lib.glass_segment_pool_max_bwd_exact_f32(dout.data_ptr(), ldd, pos.data_ptr(), B, Smax, argmax.data_ptr(),
                                                          demb.data_ptr(), C, n, C, ws.data_ptr(), _stream())
            _lib.check(rc, "glass_segment_pool_max_bwd_exact_f32")
            return demb, None, None
        demb = torch.zeros((n, C), dtype=torch.float32, device=dout.device)
        rc = _lib.load().glass_segment_pool_bwd_f32(dout.data_ptr(), ldd, pos.data_ptr(), B, Smax, POOL_MODES[mode],
                                                    0 if argmax is None else argmax.data_ptr(), demb.data_ptr(), C, n,
                                                    C, _stream())
        _lib.check(rc, "glass_segment_pool_bwd_f32")
        return demb, None, None


def segment_pool(emb, pos, mode):
    return SegmentPoolFn.apply(emb, pos, mode)
