"""The EmbZGConv stack as ONE explicit forward / backward program over the C ABI.

`models.EmbZGConv.forward` (reference impl/models.py:241-272, layers 158-173) normally records one autograd node
per kernel group (ops.py).  On the small graphs of the BASELINE configs the step is a chain of dependent 4-15 us
kernels, and both the tape and the op granularity cost real launches.  Here the whole stack is a single autograd
node (`StackFn`) — or no tape at all (`loss_and_grads`, used by step.TrainStep) — whose forward and backward walk
the layers themselves:

  * gradient sums of tensors with two consumers ride as the `addend` operand of the kernel producing the other
    summand; parameter gradients go straight into the gradient arena (accumulated, or overwritten so that the arena
    needs no zero-fill);
  * embedding lookup + emb_gn + dropout run through the V-row table (K3n) when the table is small;
  * GraphNorm statistics come from the epilogue of the kernel that produced the tensor, the apply rides in the
    operand load of the consuming fused dense kernel, backward column sums come from the data-gradient epilogues;
  * the weight-gradient partial sums of all layers are reduced by one launch at the end of the backward pass;
  * with a fusable readout (`step_supported`): final GraphNorm apply + pooling + Linear head + loss and their
    backward are the four launches of K8r, and the labels are scattered straight from `pos`.

Used when every layer takes the fused dense path (ParamArena present, GLASSConv layers of equal width that
glass_dual_linear_supported() accepts — hidden 64 / 128 —, ELU, GraphNorm on); anything else keeps the per-op path.
Same kernels, same dropout call ids -> same dropout masks as the per-op path; no float atomics -> bitwise repeatable.
"""
import os

import numpy as np
import torch

from . import _lib, ops
from .ops import ACT_ELU, ACT_NONE, ACT_RELU, _stream

USE_EMBED_TABLE = os.environ.get("GLASS_EMBED_TABLE", "1") != "0"  # A/B switch: lookup + emb_gn through the table
USE_READOUT = os.environ.get("GLASS_READOUT", "1") != "0"          # A/B switch: fused training readout (K8r)


def _check(rc, what):
    _lib.check(rc, what)


USE_GN_EXACT = os.environ.get("GLASS_GN_EXACT", "1") != "0"  # A/B switch: exact GraphNorm accumulators instead of partials + finalize
USE_READOUT_TWO = os.environ.get("GLASS_READOUT_TWO", "1") != "0"  # readout as two launches (backward sums in exact accumulators)
USE_GN_BWD_IN_COMB = os.environ.get("GLASS_GN_BWD_IN_COMB", "1") != "0"  # gns[l]'s backward apply inside the comb backward launch
USE_GN_EXACT_FWD = os.environ.get("GLASS_GN_EXACT_FWD", "1") != "0"  # ... for the forward sums as well
# above this many rows the partials + finalize form is kept: thousands of workgroups adding to the same few replicas would
# queue at the memory-side atomic units, and a ~5 us finalize launch no longer shows against the kernels around it
GN_EXACT_MAX_ROWS = 1 << 18
GN_EXACT_FWD_ONLY_MAX_ROWS = 1 << 14  # the forward sums alone (hidden 128, one layer): see StackProgram.forward
USE_FUSED_TAIL = os.environ.get("GLASS_FUSED_TAIL", "1") != "0"  # A/B switch: K1 slot sums + table backward + Adam as one launch
USE_GATHER_IN_TRANS = os.environ.get("GLASS_GATHER_IN_TRANS", "1") != "0"  # A/B switch: embedding lookup inside layer 0's trans kernel
USE_COMB_EFF = os.environ.get("GLASS_COMB_EFF", "1") != "0"  # A/B switch: comb pair through effective per-label weights


_PROLOGUE_DONE = False


class prologue_done:
    """Context: the weight images are fresh (someone ran the arena's prologue launch on a stream every forward below is
    ordered behind) — forwards that need nothing else from the prologue (no dropout step, no accumulator zero-fill, no table
    statistics: the evaluation forward) skip their own.  evalstep.EvalGraph packs once, then forks its branches."""
    def __init__(self, on=True):
        self.on = bool(on)

    def __enter__(self):
        global _PROLOGUE_DONE
        self.prev, _PROLOGUE_DONE = _PROLOGUE_DONE, (self.on or _PROLOGUE_DONE)

    def __exit__(self, *exc):
        global _PROLOGUE_DONE
        _PROLOGUE_DONE = self.prev


class BatchLabels:
    """Label state of one subgraph batch on the device (glass_batch_labels; utils.MaxZOZ, reference impl/utils.py:32-45):
    the label bytes `mask` (maintained incrementally from batch to batch), the unique labeled rows `rows[:count]` in
    first-occurrence order.  `cap` = capacity of the list = B * Smax (fixes the grids of the kernels that walk it)."""
    def __init__(self, n_nodes, cap, device):
        self.n, self.cap = int(n_nodes), int(cap)
        self.mask = torch.zeros(self.n, dtype=torch.uint8, device=device)
        self.rows = torch.zeros(max(self.cap, 1), dtype=torch.int32, device=device)
        self.count = torch.zeros(4, dtype=torch.int32, device=device)
        self.ws = torch.full((self.n,), 2**31 - 1, dtype=torch.int32, device=device)  # owner words: INT32_MAX between calls
        self.loaded = False
        # epoch source of the in-graph label launch (glass_step_head_f32): a glass_batch_cursor in device memory + what it
        # points to (kept alive here); None: the batch is labeled by an eager load() / load_gather() in front of the step
        self.cursor = None
        self._epoch = None
        self.in_head = False   # the step's head launch labels the batch (set by TrainStep while an epoch source is installed)

    def set_epoch(self, pos_all, y_all, idx_batches, pos_dst, y_dst, wrap=False):
        """Install the epoch's batches for the in-graph label launch: idx_batches int64 [n_batches, n_idx] (device, contiguous:
        the rows of every batch, in step order), pos_all / y_all the data set's matrices; the cursor restarts at batch 0
        (wrap: it cycles over the batches instead of stopping at the last one).  One small host-to-device copy per epoch; the step's graph reads everything through the cursor (fixed address)."""
        n_all, smax = pos_all.shape
        n_b, n_idx = idx_batches.shape
        if n_idx * smax != self.cap:
            raise ValueError(f"BatchLabels: batches of {n_idx * smax} entries, capacity {self.cap}")
        yrb = 0 if y_all is None else y_all.element_size() * (y_all.numel() // max(y_all.shape[0], 1))
        if self.cursor is None:
            self.cursor = torch.zeros(8, dtype=torch.int64, device=self.mask.device)
        desc = torch.tensor([pos_all.data_ptr(), 0 if y_all is None else y_all.data_ptr(), idx_batches.data_ptr(), n_all, n_b, 0, 1 if wrap else 0, 0],
                            dtype=torch.int64)
        self.cursor.copy_(desc)
        self._epoch = (pos_all, y_all, idx_batches, pos_dst, y_dst, int(n_idx), int(smax), int(yrb))

    def rewind(self):
        """The cursor back to the epoch's first batch (device-side: a fill on the current stream, outside any capture)."""
        if self.cursor is not None:
            self.cursor[5:6].zero_()

    def head_args(self):
        """The label arguments of glass_step_head_f32 (behind the prologue's)."""
        _pos_all, y_all, _idx, pos_dst, y_dst, n_idx, smax, yrb = self._epoch
        return (self.cursor.data_ptr(), n_idx, smax, yrb, pos_dst.data_ptr(), 0 if y_all is None else y_dst.data_ptr(),
                self.mask.data_ptr(), self.rows.data_ptr(), self.count.data_ptr(), self.ws.data_ptr(), self.n)

    def head_signature(self):
        """What a captured head launch baked in besides the cursor's address: batch shape and target row size."""
        return None if self._epoch is None else self._epoch[5:]

    def load(self, pos_src, pos_dst=None, y_src=None, y_dst=None):
        """Labels of `pos_src`; with pos_dst / y_dst the batch is also copied into those fixed buffers (pos_dst must hold
        the PREVIOUS batch — all -1 before the first one — because its labels are cleared incrementally)."""
        if pos_src.numel() != self.cap:
            raise ValueError(f"BatchLabels: batch of {pos_src.numel()} entries, capacity {self.cap}")
        yb = 0 if y_src is None else y_src.numel() * y_src.element_size()
        incremental = 1 if pos_dst is not None else 0  # without fixed buffers the N label bytes are zero-filled each time
        rc = _lib.load().glass_batch_labels(pos_src.data_ptr(), pos_src.numel(), 0 if pos_dst is None else pos_dst.data_ptr(),
                                            0 if y_src is None else y_src.data_ptr(), 0 if y_dst is None else y_dst.data_ptr(),
                                            yb, self.mask.data_ptr(), self.rows.data_ptr(), self.count.data_ptr(),
                                            self.ws.data_ptr(), self.n, incremental, _stream())
        _check(rc, "glass_batch_labels")
        self.loaded = True

    def load_gather(self, pos_all, y_all, idx, pos_dst, y_dst):
        """The batch = rows `idx` of the data set's node / target matrices, selected by the label launch itself
        (glass_batch_labels_gather: ZGDataloader's `pos[perm], y[perm]` without the two index kernels); pos_dst / y_dst
        receive the selected rows (pos_dst must hold the previous batch, as in load())."""
        n_all, smax = pos_all.shape
        if idx.numel() * smax != self.cap:
            raise ValueError(f"BatchLabels: batch of {idx.numel() * smax} entries, capacity {self.cap}")
        yrb = 0 if y_all is None else y_all.element_size() * (y_all.numel() // max(y_all.shape[0], 1))
        rc = _lib.load().glass_batch_labels_gather(pos_all.data_ptr(), n_all, smax, 0 if y_all is None else y_all.data_ptr(), yrb,
                                                   idx.data_ptr(), idx.numel(), pos_dst.data_ptr(),
                                                   0 if y_all is None else y_dst.data_ptr(), self.mask.data_ptr(),
                                                   self.rows.data_ptr(), self.count.data_ptr(), self.ws.data_ptr(), self.n, 1,
                                                   _stream())
        _check(rc, "glass_batch_labels_gather")
        self.loaded = True


def _comb_eff_ok(conv, labels, H):
    """forward AND backward of the comb pair in effective-weight form (hidden 64)"""
    return (USE_COMB_EFF and labels is not None and getattr(conv, "_stack_eff", {}).get("comb", (None, None))[1] is not None and
            bool(_lib.load().glass_comb_eff_supported(H)) and labels.n <= _lib.load().glass_comb_eff_max_rows(4 * H))


def _comb_eff_fwd_ok(conv, labels, H):
    """the forward alone (also hidden 128)"""
    return (USE_COMB_EFF and labels is not None and "comb" in getattr(conv, "_stack_eff", {}) and
            bool(_lib.load().glass_comb_eff_fwd_supported(H)) and labels.n <= _lib.load().glass_comb_eff_max_rows(4 * H))


class _PendingStats:
    """Forward sums of a GraphNorm still in exact accumulators (gn_acc.h): the kernel that applies it derives the
    coefficients from `acc` (n_src consecutive blocks) and writes `saved` [4C] for the backward."""
    def __init__(self, saved, acc, n_src, mod, n_rep):
        self.saved, self.src = saved, _lib.GnSrc.of(acc, n_src, n_rep, mod)


def _saved_args(saved):
    """(saved pointer, glass_gn_src pointer) of a GraphNorm prologue: final statistics, or pending exact sums."""
    if isinstance(saved, _PendingStats):
        return saved.saved.data_ptr(), saved.src.ptr
    return saved.data_ptr(), 0


# replicas of the exact accumulators (gn_acc.h) by producer: the stand-alone statistics kernel's workgroups all finish
# together (their adds to one replica queue up: 10.6 / 8.1 / 6.6 us with 4 / 8 / 16), the dense kernels' epilogues are spread
# over the launch — and every consumer workgroup reads n_rep * 2 KB
REP_STATS, REP_DENSE = 16, int(os.environ.get("GLASS_REP_DENSE", "16"))


def _rep(t):
    """replicas in use for an accumulator tensor (int64), 0 for the float64 partials form"""
    return REP_DENSE if t.dtype == torch.int64 else 0


def _stats_args(stats):
    """(pointer, stats_exact) of an epilogue statistics target: float64 partials, or int64 exact accumulators."""
    if stats is None:
        return 0, 0
    return stats.data_ptr(), _rep(stats)


def _comb_eff_fwd(xa, xb, conv, mask, out, stats, gn, labels):
    n, H = xa.shape
    saved, gact, gp, gcall, xa_out = gn
    grng = ops.rng_tensor(xa.device).data_ptr() if gp > 0 else 0
    rc = _lib.load().glass_comb_eff_fwd_f32(xa.data_ptr(), xa.stride(0), xb.data_ptr(), xb.stride(0),
                                            conv._stack_eff["comb"][0].data_ptr(), conv._stack["comb"][1].data_ptr(),
                                            mask.data_ptr(), float(conv.z_ratio), out.data_ptr(), out.stride(0), n, H,
                                            *_stats_args(stats), *_saved_args(saved), ops.act_word(gact), float(gp), grng, gcall,
                                            xa_out.data_ptr(), xa_out.stride(0), labels.rows.data_ptr(),
                                            labels.count.data_ptr(), labels.cap, _stream())
    _check(rc, "glass_comb_eff_fwd_f32")


def _comb_eff_bwd(dsrc, conv, mask, out, xa, xb, pending, acc, gn, labels, dsrc_gn=None):
    """dsrc_gn (a _lib.GnBwdSrc): dsrc is not materialised — the kernels derive it on load from the gradient of the
    GraphNorm behind this layer (glass_gn_bwd_src); dsrc is then only a shape carrier (the tensor dy)."""
    n, H = dsrc.shape
    gpart, gx, gsaved, galpha, gact, gp, gcall = gn
    rng = ops.rng_tensor(dsrc.device).data_ptr() if (gp > 0 or (dsrc_gn is not None and dsrc_gn.p_drop > 0)) else 0
    stack = conv._stack["comb"]
    ws = ops._wgrad_workspace(dsrc.device, n, 2 * H, 2 * H, slot=("stack", len(pending)),
                              min_bytes=int(_lib.load().glass_comb_eff_ws_bytes(n, H, labels.cap)))
    rc = _lib.load().glass_comb_eff_bwd_f32(0 if dsrc_gn is not None else dsrc.data_ptr(), dsrc.stride(0), mask.data_ptr(), float(conv.z_ratio),
                                            conv._stack_eff["comb"][1].data_ptr(), out.data_ptr(), out.stride(0), n, H,
                                            gpart.data_ptr(), gx.data_ptr(), gx.stride(0), gsaved.data_ptr(),
                                            galpha.data_ptr(), ops.act_word(gact), float(gp), rng, gcall, _rep(gpart),
                                            xa.data_ptr(), xa.stride(0),
                                            xb.data_ptr(), xb.stride(0), ws.data_ptr(), labels.rows.data_ptr(),
                                            labels.count.data_ptr(), labels.cap, 0 if dsrc_gn is None else dsrc_gn.ptr, _stream())
    _check(rc, "glass_comb_eff_bwd_f32")
    # (9th field: the partials are in S / L form for a labeled-row list of that capacity)
    pending.append((ws.data_ptr(), n, 2 * H, 2 * H, stack[2].data_ptr(), stack[2].stride(0), stack[3].data_ptr(), acc, labels.cap))


class _GN:
    """Launch helpers for one GraphNorm module (weights read in place; gradients accumulated in place)."""
    def __init__(self, mod):
        self.mod = mod

    def fwd(self, x, y, act, p_drop, call_id):
        m = self.mod
        n, C = x.shape
        saved = torch.empty(4 * C, dtype=torch.float32, device=x.device)
        ws = ops._graphnorm_ws(x.device, n, C)
        rng = ops.rng_tensor(x.device).data_ptr() if p_drop > 0 else 0
        rc = _lib.load().glass_graphnorm_fwd_f32(x.data_ptr(), x.stride(0), y.data_ptr(), y.stride(0), n, C,
                                                 m.weight.data_ptr(), m.bias.data_ptr(), m.mean_scale.data_ptr(),
                                                 float(m.eps), saved.data_ptr(), act, float(p_drop), rng, call_id,
                                                 ws.data_ptr(), _stream())
        _check(rc, "glass_graphnorm_fwd_f32")
        return saved

    def stats(self, x):
        """saved[4C] (statistics + finalize, no apply pass): the consumer normalises while loading."""
        m = self.mod
        n, C = x.shape
        saved = torch.empty(4 * C, dtype=torch.float32, device=x.device)
        rc = _lib.load().glass_graphnorm_stats_f32(x.data_ptr(), x.stride(0), n, C, m.weight.data_ptr(), m.bias.data_ptr(),
                                                   m.mean_scale.data_ptr(), float(m.eps), saved.data_ptr(),
                                                   ops._graphnorm_ws(x.device, n, C).data_ptr(), _stream())
        _check(rc, "glass_graphnorm_stats_f32")
        return saved

    def stats_exact(self, x, acc):
        """The statistics pass alone, into exact accumulators: the consumer derives the coefficients (no finalize launch)."""
        n, C = x.shape
        saved = torch.empty(4 * C, dtype=torch.float32, device=x.device)
        rc = _lib.load().glass_graphnorm_stats_exact_f32(x.data_ptr(), x.stride(0), n, C, acc.data_ptr(), REP_STATS, _stream())
        _check(rc, "glass_graphnorm_stats_exact_f32")
        return _PendingStats(saved, acc, 1, self.mod, REP_STATS)

    def finalize(self, stats, n_rows):
        """saved[4C] from statistics the producers' epilogues wrote (list of [nblk, 2, C_each] float64 buffers)."""
        m = self.mod
        C_each, nblk = stats[0].shape[2], stats[0].shape[0]
        C = C_each * len(stats)
        saved = torch.empty(4 * C, dtype=torch.float32, device=stats[0].device)
        ptrs = np.array([t.data_ptr() for t in stats], dtype=np.uint64)
        rc = _lib.load().glass_graphnorm_finalize_f32(ptrs.ctypes.data, len(stats), nblk, C_each, n_rows,
                                                      m.weight.data_ptr(), m.bias.data_ptr(), m.mean_scale.data_ptr(),
                                                      float(m.eps), saved.data_ptr(), _stream())
        _check(rc, "glass_graphnorm_finalize_f32")
        return saved

    def apply(self, x, y, saved, act, p_drop, call_id):
        n, C = x.shape
        rng = ops.rng_tensor(x.device).data_ptr() if p_drop > 0 else 0
        rc = _lib.load().glass_graphnorm_apply_f32(x.data_ptr(), x.stride(0), y.data_ptr(), y.stride(0), n, C,
                                                   saved.data_ptr(), act, float(p_drop), rng, call_id, _stream())
        _check(rc, "glass_graphnorm_apply_f32")

    def bwd_from_stats(self, dy, x, saved, dx, partial, act, p_drop, call_id, addend=None, acc=1):
        """Backward with the two column sums already accumulated (data-gradient epilogue): finalize + apply."""
        m = self.mod
        n, C = x.shape
        ws = ops._graphnorm_ws(x.device, n, C)
        rng = ops.rng_tensor(x.device).data_ptr() if p_drop > 0 else 0
        ap, lda = (0, 0) if addend is None else (addend.data_ptr(), addend.stride(0))
        rc = _lib.load().glass_graphnorm_bwd_from_stats_f32(dy.data_ptr(), dy.stride(0), x.data_ptr(), x.stride(0),
                                                            dx.data_ptr(), dx.stride(0), ap, lda, n, C,
                                                            m.weight.data_ptr(), m.mean_scale.data_ptr(), saved.data_ptr(),
                                                            partial.data_ptr(), -_rep(partial) if partial.dtype == torch.int64 else partial.shape[0],
                                                            m.weight.grad.data_ptr(), m.bias.grad.data_ptr(),
                                                            m.mean_scale.grad.data_ptr(), acc, act, float(p_drop), rng,
                                                            call_id, ws.data_ptr(), _stream())
        _check(rc, "glass_graphnorm_bwd_from_stats_f32")

    def bwd(self, dy, x, saved, dx, act, p_drop, call_id, addend=None, acc=1):
        m = self.mod
        n, C = x.shape
        ws = ops._graphnorm_ws(x.device, n, C)
        rng = ops.rng_tensor(x.device).data_ptr() if p_drop > 0 else 0
        ap, lda = (0, 0) if addend is None else (addend.data_ptr(), addend.stride(0))
        rc = _lib.load().glass_graphnorm_bwd_f32(dy.data_ptr(), dy.stride(0), x.data_ptr(), x.stride(0), dx.data_ptr(),
                                                 dx.stride(0), ap, lda, n, C, m.weight.data_ptr(),
                                                 m.mean_scale.data_ptr(), saved.data_ptr(), m.weight.grad.data_ptr(),
                                                 m.bias.grad.data_ptr(), m.mean_scale.grad.data_ptr(), acc, act,
                                                 float(p_drop), rng, call_id, ws.data_ptr(), _stream())
        _check(rc, "glass_graphnorm_bwd_f32")


def _dual_fwd(xa, xb, stack, mask, z_ratio, act, T, out, stats=None, gn=None, xa_index=None):
    """stats: [ceil(n/64), 2, H] float64 — per-workgroup column sums of `out` for the GraphNorm that follows.
    gn = (saved, act, p_drop, call_id, xa_out): xa is the raw input of a GraphNorm with final statistics `saved`; the
    kernel normalises (+act +dropout) while loading and writes the normalised operand to xa_out.
    xa_index (int64 [n]): operand row k is row xa_index[k] of xa (the embedding table; needs gn, whose side output xa_out
    is then the gathered, normalised [n, H] layer input)."""
    n, H = (xa_index.shape[0] if xa_index is not None else xa.shape[0]), xa.shape[1]
    if gn is not None:
        saved, gact, gp, gcall, xa_out = gn
        grng = ops.rng_tensor(xa.device).data_ptr() if gp > 0 else 0
        gargs = (*_saved_args(saved), gact, float(gp), grng, gcall, xa_out.data_ptr(), xa_out.stride(0))
    else:
        gargs = (0, 0, 0, 0.0, 0, 0, 0, 0)
    iargs = (0, 0) if xa_index is None else (xa_index.data_ptr(), xa.shape[0])
    rc = _lib.load().glass_dual_linear_fwd_f32(xa.data_ptr(), xa.stride(0), 0 if xb is None else xb.data_ptr(),
                                               0 if xb is None else xb.stride(0), stack[4].data_ptr(),
                                               stack[1].data_ptr(), mask.data_ptr(), float(z_ratio), ops.act_word(act),
                                               0 if T is None else T.data_ptr(), 0 if T is None else T.stride(0),
                                               out.data_ptr(), out.stride(0), n, H, *_stats_args(stats), *gargs, *iargs,
                                               _stream())
    _check(rc, "glass_dual_linear_fwd_f32")


def _dual_dgrad(dsrc, T, stack, mask, z_ratio, act, n_out, addend, out, drop=None, gn=None):
    """drop = (p, call_id): also multiply by that dropout's mask (gradient w.r.t. the pre-dropout layer input).
    gn = (partial, x, saved, alpha, act, p, call_id): the first H output columns are the gradient of that GraphNorm's
    output; its backward column sums are accumulated into `partial` [ceil(n/64), 2, H] float64 by the epilogue."""
    n, H = dsrc.shape
    p_drop, call_id = drop if drop is not None else (0.0, 0)
    if gn is not None:
        gpart, gx, gsaved, galpha, gact, gp, gcall = gn
        gargs = (gpart.data_ptr(), gx.data_ptr(), gx.stride(0), gsaved.data_ptr(), galpha.data_ptr(), gact, float(gp), gcall,
                 _rep(gpart))  # int64 buffer = the exact accumulators of gn_acc.h
    else:
        gp, gargs = 0.0, (0, 0, 0, 0, 0, 0, 0.0, 0, 0)
    rng = ops.rng_tensor(dsrc.device).data_ptr() if (p_drop > 0 or gp > 0) else 0
    rc = _lib.load().glass_dual_linear_dgrad_f32(dsrc.data_ptr(), dsrc.stride(0), 0 if T is None else T.data_ptr(),
                                                 0 if T is None else T.stride(0), mask.data_ptr(), float(z_ratio), ops.act_word(act),
                                                 stack[5].data_ptr(), n_out, 0 if addend is None else addend.data_ptr(),
                                                 0 if addend is None else addend.stride(0), float(p_drop), rng, call_id,
                                                 out.data_ptr(), out.stride(0), n, H, *gargs, _stream())
    _check(rc, "glass_dual_linear_dgrad_f32")


USE_FUSED_BWD = os.environ.get("GLASS_FUSED_BWD", "1") != "0"  # A/B switch: data + weight gradient of a pair as one launch


def _dual_bwd(dsrc, T, stack, mask, z_ratio, act, n_out, addend, out, xa, xb, pending, acc, drop=None, gn=None):
    """Backward of one Linear pair: _dual_dgrad + _dual_wgrad through glass_dual_linear_bwd_f32 (ONE launch at hidden 64 on
    small graphs, where the two are independent latency-bound kernels; two launches inside the library otherwise)."""
    if not USE_FUSED_BWD:
        _dual_dgrad(dsrc, T, stack, mask, z_ratio, act, n_out, addend, out, drop, gn)
        _dual_wgrad(dsrc, T, stack, mask, z_ratio, act, xa, xb, pending, acc)
        return
    n, H = dsrc.shape
    I = H if xb is None else 2 * H
    p_drop, call_id = drop if drop is not None else (0.0, 0)
    if gn is not None:
        gpart, gx, gsaved, galpha, gact, gp, gcall = gn
        gargs = (gpart.data_ptr(), gx.data_ptr(), gx.stride(0), gsaved.data_ptr(), galpha.data_ptr(), gact, float(gp), gcall,
                 _rep(gpart))
    else:
        gp, gargs = 0.0, (0, 0, 0, 0, 0, 0, 0.0, 0, 0)
    rng = ops.rng_tensor(dsrc.device).data_ptr() if (p_drop > 0 or gp > 0) else 0
    ws = ops._wgrad_workspace(dsrc.device, n, 2 * H, I, slot=("stack", len(pending)))
    rc = _lib.load().glass_dual_linear_bwd_f32(dsrc.data_ptr(), dsrc.stride(0), 0 if T is None else T.data_ptr(),
                                               0 if T is None else T.stride(0), mask.data_ptr(), float(z_ratio), ops.act_word(act),
                                               stack[5].data_ptr(), n_out, 0 if addend is None else addend.data_ptr(),
                                               0 if addend is None else addend.stride(0), float(p_drop), rng, call_id,
                                               out.data_ptr(), out.stride(0), n, H, *gargs, xa.data_ptr(), xa.stride(0),
                                               0 if xb is None else xb.data_ptr(), 0 if xb is None else xb.stride(0),
                                               ws.data_ptr(), _stream())
    _check(rc, "glass_dual_linear_bwd_f32")
    pending.append((ws.data_ptr(), n, 2 * H, I, stack[2].data_ptr(), stack[2].stride(0), stack[3].data_ptr(), acc))


# (Forking the weight-gradient partial kernels onto a second stream inside the captured step was measured on MI355X at C2,
# two interleaved rounds: 0.441 vs 0.386 ms/step — the fork/join edges of the replayed graph cost more than overlapping
# these 16 us kernels with the 12-20 us kernels of the main chain buys.  Not part of the product)


def _dual_wgrad(dout, T, stack, mask, z_ratio, act, xa, xb, pending, acc=1):
    """Per-slab partial sums of dW / db into a scratch buffer of their own; the reduction into the gradient arena
    is deferred to ONE launch at the end of the backward pass (`_reduce_pending`)."""
    n, H = dout.shape
    I = H if xb is None else 2 * H
    ws = ops._wgrad_workspace(dout.device, n, 2 * H, I, slot=("stack", len(pending)))
    rc = _lib.load().glass_dual_linear_wgrad_f32(dout.data_ptr(), dout.stride(0), 0 if T is None else T.data_ptr(),
                                                 0 if T is None else T.stride(0), mask.data_ptr(), float(z_ratio), ops.act_word(act),
                                                 xa.data_ptr(), xa.stride(0), 0 if xb is None else xb.data_ptr(),
                                                 0 if xb is None else xb.stride(0), n, H, 0, 0, 0, 1, ws.data_ptr(),
                                                 _stream())
    _check(rc, "glass_dual_linear_wgrad_f32")
    pending.append((ws.data_ptr(), n, 2 * H, I, stack[2].data_ptr(), stack[2].stride(0), stack[3].data_ptr(), acc))


def _reduce_pending(pending, product=None):
    """One launch for all deferred weight-gradient reductions.  product = (graph.Selection, x): the same launch also carries
    the selection product G = S^T x of the embedding backward, WITHOUT its reduce step (the table backward sums the partial
    rows itself); returns (G, partials pointer, reduce-list pointer, number of reduce rows) then."""
    if product is not None:
        return _reduce_pending_with_product(pending, *product)
    if not pending:
        return
    cols = list(zip(*[p if len(p) == 9 else p + (0, ) for p in pending]))
    u64 = lambda v: np.array(v, dtype=np.uint64)
    i64 = lambda v: np.array(v, dtype=np.int64)
    ws, N, O, I, dW, ld, db = u64(cols[0]), i64(cols[1]), i64(cols[2]), i64(cols[3]), u64(cols[4]), i64(cols[5]), u64(cols[6])
    acc, cap = np.array(cols[7], dtype=np.int32), i64(cols[8])
    rc = _lib.load().glass_linear_wgrad_reduce_batch_f32(len(pending), ws.ctypes.data, N.ctypes.data, O.ctypes.data,
                                                         I.ctypes.data, dW.ctypes.data, ld.ctypes.data, db.ctypes.data,
                                                         acc.ctypes.data, cap.ctypes.data, _stream())
    _check(rc, "glass_linear_wgrad_reduce_batch_f32")


def _reduce_pending_with_product(pending, sel, x):
    cols = list(zip(*[p if len(p) == 9 else p + (0, ) for p in pending])) if pending else [()] * 9
    u64 = lambda v: np.array(v, dtype=np.uint64)
    i64 = lambda v: np.array(v, dtype=np.int64)
    ws, N, O, I, dW, ld, db = u64(cols[0]), i64(cols[1]), i64(cols[2]), i64(cols[3]), u64(cols[4]), i64(cols[5]), u64(cols[6])
    acc, cap = np.array(cols[7], dtype=np.int32), i64(cols[8])
    op = sel.op
    H = x.shape[1]
    G = torch.empty((op.n_rows, H), dtype=torch.float32, device=x.device)
    wsp = op.workspace(H)
    rc = _lib.load().glass_wgrad_reduce_spmm_f32(len(pending), ws.ctypes.data, N.ctypes.data, O.ctypes.data, I.ctypes.data,
                                                 dW.ctypes.data, ld.ctypes.data, db.ctypes.data, acc.ctypes.data,
                                                 cap.ctypes.data, op.rowptr.data_ptr(), op.col.data_ptr(), op.val.data_ptr(),
                                                 x.data_ptr(), x.stride(0), G.data_ptr(), G.stride(0), op.n_rows, H,
                                                 sel._hdr_noreduce.ctypes.data, op.plan.data_ptr(), wsp.data_ptr(), _stream())
    _check(rc, "glass_wgrad_reduce_spmm_f32")
    return G, wsp.data_ptr(), op.plan.data_ptr() + 4 * sel._off_reduce, sel._n_reduce


class StackProgram:
    """Forward / backward of one EmbZGConv over preselected kernels.  `supported(emb)` is the gate."""
    def __init__(self, emb):
        self.emb = emb
        self.loss_sum = None  # float32 device scalar the readout adds every step's loss to (step.TrainStep's epoch sum)

    @staticmethod
    def supported_unlabeled(emb):
        """The pre-training stack (models.EmbGConv of MyGCNConv layers: reference impl/models.py:361-473) on the same
        program: no labels (an all-zero label byte vector, an empty labeled-row list), every Linear as the second half of a
        pair with z_ratio = 1 (arena._pairs), no GraphNorm behind the embedding, none behind the last layer, no JK — hidden
        64 (the staged kernels + exact GraphNorm accumulators), ELU or ReLU."""
        from .models import EmbGConv, MyGCNConv, _act_code
        if not (isinstance(emb, EmbGConv) and ops.USE_FUSED_DENSE and getattr(emb, "_glass_arena", None) is not None and
                emb.gns is not None and not emb.jk and len(emb.convs)):
            return False
        code = _act_code(emb.activation)
        H = emb.input_emb.weight.shape[1]
        lib = _lib.load()
        if code not in (ACT_ELU, ACT_RELU) or H != 64 or not (USE_GN_EXACT and lib.glass_gn_exact_supported(H) and USE_COMB_EFF and
                                                            lib.glass_comb_eff_supported(H)):
            return False
        for c in emb.convs:
            st = getattr(c, "_stack", {})
            if not (isinstance(c, MyGCNConv) and _act_code(c.activation) == code and len(st.get("trans", ())) == 6 and
                    len(st.get("comb", ())) == 6 and getattr(c, "_stack_eff", {}).get("comb", (None, None))[1] is not None and
                    c.trans_fn.weight.shape == (H, H) and c.comb_fn.weight.shape == (H, 2 * H)):
                return False
        mods = [c.gn for c in emb.convs] + list(emb.gns)
        return all(getattr(m, "_direct_grad", False) and m.weight.grad is not None for m in mods) and \
            emb.input_emb.weight.grad is not None

    @staticmethod
    def supported(emb):
        from .models import GLASSConv, EmbGConv, _act_code
        if isinstance(emb, EmbGConv):
            return StackProgram.supported_unlabeled(emb)
        if not ops.USE_FUSED_DENSE or getattr(emb, "_glass_arena", None) is None or emb.gns is None:
            return False
        code = _act_code(emb.activation)
        if code not in (ACT_ELU, ACT_RELU) or not len(emb.convs):
            return False
        H = emb.input_emb.weight.shape[1]
        if not ops.dual_linear_supported(H):
            return False
        for c in emb.convs:
            st = getattr(c, "_stack", {})
            if not (isinstance(c, GLASSConv) and _act_code(c.activation) == code and
                    len(st.get("trans", ())) == 6 and len(st.get("comb", ())) == 6 and
                    c.trans_fns[0].weight.shape == (H, H) and c.comb_fns[0].weight.shape == (H, 2 * H)):
                return False
        mods = [emb.emb_gn] + [c.gn for c in emb.convs] + list(emb.gns)
        return all(getattr(m, "_direct_grad", False) and m.weight.grad is not None for m in mods) and \
            emb.input_emb.weight.grad is not None

    # ---------------------------------------------------------------------------------------------
    def forward(self, x_flat, z, edge_index, edge_weight, keep, readout=None, acc=1, labels=None, snapshot_rng=False):
        """Returns (out, state).  keep=False (no gradient wanted): intermediates are dropped as soon as possible.
        readout = (pos, pool_mode, head Linear, target, loss_mode): instead of the final GraphNorm apply, run the
        fused training readout (K8r) — out = (loss, logits) and state carries the gradient of the JK buffer.
        snapshot_rng: this pass keeps a private copy of the dropout words (seed, step) for its backward, so other training
        forwards may come in between (the autograd path); False: forward and backward run back to back on the live words."""
        dev = x_flat.device
        with ops.rng_scope(dev, None):  # (the prologue advances the LIVE stream; the snapshot, if any, is taken right after)
            return self._forward(x_flat, z, edge_index, edge_weight, keep, readout, acc, labels, snapshot_rng)

    def _forward(self, x_flat, z, edge_index, edge_weight, keep, readout, acc, labels, snapshot_rng):
        from .models import buildAdj
        emb, lib = self.emb, _lib.load()
        dev = x_flat.device
        n = x_flat.shape[0]
        W = emb.input_emb.weight
        V, H = W.shape
        L = len(emb.convs)
        train = emb.training
        p = float(emb.dropout) if train else 0.0
        f32 = dict(dtype=torch.float32, device=dev)
        # once-per-step prologue, one launch: operand images of the current weights + new dropout masks
        advance = train and (p > 0 or any(c.dropout > 0 for c in emb.convs))
        from .models import _act_code
        act = _act_code(emb.activation)  # ELU or ReLU, the same code in every layer (supported())
        st = {"n": n, "H": H, "L": L, "p": p, "x_flat": x_flat, "acc": int(acc), "rng_epoch": ops.rng_epoch(dev), "act": act,
              "stat_rows": int(lib.glass_dual_linear_stat_rows(H))}  # rows per workgroup of the fused dense kernels
        # labels: z (int64 [N], > 0 = labeled), None (all labeled), or ("pos", pos): labeled = the nodes listed in the
        # padded subgraph matrix — utils.MaxZOZ without materialising z (a byte memset + scatter inside the gather)
        # labels (BatchLabels, already loaded for this batch): the label bytes are an input and the unique labeled rows are
        # listed — the comb pairs then run in effective-weight form at hidden 64.  With ("pos", pos) and no labels handed
        # in, they are computed here (one extra launch; the replayed training step hands them in).
        unl = not hasattr(emb, "emb_gn")  # the unlabeled pre-training stack (models.EmbGConv): see supported_unlabeled
        st["unlabeled"] = unl
        if unl:
            labels = emb.__dict__.get("_glass_no_labels")
            if labels is None or labels.n != n or labels.mask.device != dev:
                labels = emb.__dict__["_glass_no_labels"] = BatchLabels(n, 1, dev)  # zero label bytes, count 0 (capacity 1:
                labels.loaded = True                                                 # the S / L reduce keys on cap > 0)
        if isinstance(z, tuple) and labels is None and train:
            labels = BatchLabels(n, z[1].numel(), dev)
            labels.load(z[1])
        if labels is not None:
            zp, pp, npos = 0, 0, -1
            mask = labels.mask
        elif isinstance(z, tuple):
            zp, pp, npos = 0, z[1].data_ptr(), z[1].numel()
            mask = torch.empty(n, dtype=torch.uint8, device=dev)
        else:
            zp, pp, npos = (0 if z is None else z.data_ptr()), 0, 0
            mask = torch.empty(n, dtype=torch.uint8, device=dev)
        st["labels"] = labels
        h = torch.empty((n, H), **f32)
        st["mask"] = mask
        gn0 = None if unl else emb.emb_gn
        use_table = (V <= _lib.EMBED_NORM_MAX_ROWS and USE_EMBED_TABLE) or unl
        # Hidden 64 with the label bytes already made (labels): layer 0's trans kernel gathers its operand rows from the
        # embedding table itself (xa_index) and normalises them with emb_gn's table statistics, which ride in the prologue
        # launch next to the weight packing — no table-apply kernel, no gather launch.
        first_gn = None
        # exact cross-workgroup GraphNorm sums (gn_acc.h) for the backward column sums: one int64 block per GraphNorm whose
        # backward sums come from a data-gradient epilogue (conv.gn of every layer, gns[l] between layers), zero-filled by the
        # prologue launch — their finalize launches disappear
        # ... and for the forward sums when the fused readout applies the final GraphNorm: [L] blocks for conv.gn's sums of a_l,
        # then [L] for the sums of c_l (consecutive: the column blocks of the jumping-knowledge buffer)
        acc_all = acc_bwd = acc_fwd = acc_ro = None
        exact_all = USE_GN_EXACT and bool(lib.glass_gn_exact_supported(H)) and n <= GN_EXACT_MAX_ROWS
        # the forward sums alone where every kernel that applies such a GraphNorm takes the pending form: the staged comb
        # forward (conv.gn) and the readout (final GraphNorm) do at hidden 128 too, the tiled trans kernel (gns[l] between
        # layers) does not — so: one layer, comb pair in effective-weight form
        # (measured at hidden 128, one layer, with / without: 0.2552 / 0.2564 ms at N = 12 000, 0.3526 / 0.3491 at 20 000,
        # 0.6272 / 0.6106 at 50 000 — every consumer workgroup folds 64 KB of replicas, which only pays while the launch is
        # about one workgroup per CU: hence the row limit)
        exact_fwd_only = (not exact_all and USE_GN_EXACT and n <= GN_EXACT_FWD_ONLY_MAX_ROWS and
                          bool(lib.glass_gn_exact_fwd_supported(H)) and L == 1 and
                          all(_comb_eff_fwd_ok(conv, labels, H) for conv in emb.convs))
        if exact_all or exact_fwd_only:
            n_bwd = 2 * L - 1 if (keep and exact_all) else 0
            n_fwd = 2 * L if ((readout is not None or unl) and USE_GN_EXACT_FWD) else 0
            # ... and one block of L*H (jk) / H columns for the final GraphNorm's backward sums, added to by the readout's
            # first kernel and folded by its backfill launch (two launches instead of three)
            w_h = int(lib.glass_gn_exact_words(H))
            w_ro = int(lib.glass_gn_exact_words(H * L if emb.jk else H)) if (readout is not None and keep and USE_READOUT_TWO) else 0
            if n_bwd + n_fwd:
                acc_all = torch.empty((n_bwd + n_fwd) * w_h + w_ro, dtype=torch.int64, device=dev)
                blocks = acc_all[:(n_bwd + n_fwd) * w_h].view(n_bwd + n_fwd, w_h)
                acc_bwd = blocks[:n_bwd] if n_bwd else None
                acc_fwd = blocks[n_bwd:] if n_fwd else None
                acc_ro = acc_all[(n_bwd + n_fwd) * w_h:] if w_ro else None
        elif (USE_GN_EXACT and USE_READOUT_TWO and readout is not None and keep and labels is not None and H % 4 == 0 and
              (H * L if emb.jk else H) <= 128 and n <= (1 << 16)):
            # the readout's own block alone: its backward column sums in exact accumulators, folded by the backfill launch —
            # two launches instead of three (every backfill workgroup folds n_rep * 2C sums: narrow outputs, mid-size graphs)
            acc_all = acc_ro = torch.empty(int(lib.glass_gn_exact_words(H * L if emb.jk else H)), dtype=torch.int64, device=dev)
        st["gn_exact"] = acc_bwd
        st["gn_exact_readout"] = acc_ro
        # the step's head launch labels the batch itself (TrainStep.begin_epoch: prologue || labels, glass_step_head_f32)
        head = labels if (labels is not None and getattr(labels, "in_head", False) and train) else None
        if unl:
            # no GraphNorm behind the embedding (impl/models.py:461-463): layer 0's trans kernel gathers the table rows under
            # identity coefficients (mean 0, rstd 1, scale 1, shift 0) and applies the embedding's dropout (call id 1)
            sel = emb._selection(x_flat)
            ident = emb.__dict__.get("_glass_ident_saved")
            if ident is None or ident.device != dev:
                ident = emb.__dict__["_glass_ident_saved"] = torch.cat([torch.zeros(H, **f32), torch.ones(2 * H, **f32), torch.zeros(H, **f32)])
            emb._glass_arena.refresh_transposes(ops.rng_state(dev) if advance else None, zero=acc_all, head=head)
            st["emb_table"], st["emb_saved"] = sel, ident
            first_gn = (ident, ACT_NONE, p, 1)
        elif use_table and labels is not None and USE_GATHER_IN_TRANS and lib.glass_dual_linear_fwd_gather_supported(H):
            sel = emb._selection(x_flat)
            saved = torch.empty(4 * H, **f32)
            emb._glass_arena.refresh_transposes(ops.rng_state(dev) if advance else None,
                                                table=(W, V, sel.op.rowptr, gn0, saved, None), zero=acc_all, head=head)
            st["emb_table"], st["emb_saved"] = sel, saved
            first_gn = (saved, ACT_NONE, p, 1)
        elif _PROLOGUE_DONE and not advance and acc_all is None and head is None:
            pass  # (evaluation branch behind a shared prologue: prologue_done)
        else:
            # once-per-step prologue, one launch: operand images of the current weights + new dropout masks
            emb._glass_arena.refresh_transposes(ops.rng_state(dev) if advance else None, zero=acc_all, head=head)
        st["rng_epoch"] = ops.rng_epoch(dev)  # (the prologue launch above advanced the dropout stream)
        st["rng_words"] = ops.rng_snapshot(dev) if (snapshot_rng and advance and keep) else None
        ops._rng_cur[ops._dev(dev)] = st["rng_words"]  # the rest of this forward (inside forward()'s rng_scope) reads the snapshot
        if first_gn is not None:
            pass
        elif use_table:
            # K3n: lookup + emb_gn + dropout through the V-row table (statistics are count-weighted sums over W)
            sel = emb._selection(x_flat)
            saved = torch.empty(4 * H, **f32)
            table = torch.empty((V, H), **f32)
            rng = ops.rng_tensor(dev).data_ptr() if p > 0 else 0
            _check(lib.glass_embed_norm_fwd_f32(x_flat.data_ptr(), W.data_ptr(), V, sel.op.rowptr.data_ptr(),
                                                gn0.weight.data_ptr(), gn0.bias.data_ptr(), gn0.mean_scale.data_ptr(),
                                                float(gn0.eps), saved.data_ptr(), table.data_ptr(),
                                                zp, pp, npos, p, rng, 1, h.data_ptr(), H,
                                                mask.data_ptr(), n, H, _stream()), "glass_embed_norm_fwd_f32")
            st["emb_table"], st["emb_saved"] = sel, saved
        else:
            # K3+K4: embedding gather + label byte, then the whole-graph GraphNorm
            h0 = torch.empty((n, H), **f32)
            _check(lib.glass_embed_label_f32(x_flat.data_ptr(), W.data_ptr(), V, zp, pp, npos, h0.data_ptr(), H,
                                             mask.data_ptr(), n, H, _stream()),
                   "glass_embed_label_f32")
            st["h0"] = h0
            st["emb_saved"] = _GN(gn0).fwd(h0, h, ACT_NONE, p, 1)
        C_out = H * L if emb.jk else H
        jk = torch.empty((n, C_out), **f32)
        layers, cstats = [], []
        pending_gn = None  # (saved, act, p, call_id) of the GraphNorm between the previous layer and this one
        c_prev = None
        for l, conv in enumerate(emb.convs):
            if conv.adj is None:
                conv.adj = buildAdj(edge_index, edge_weight, n, conv.aggr)
            pc = float(conv.dropout) if train else 0.0
            T = torch.empty((n, 2 * H), **f32)
            m = torch.empty((n, H), **f32)
            if pending_gn is None and first_gn is not None:
                # h = dropout(emb_gn(input_emb(x))) is this kernel's side output
                _dual_fwd(W, None, conv._stack["trans"], mask, conv.z_ratio, act, T, m, gn=(*first_gn, h), xa_index=x_flat)
            elif pending_gn is None:
                _dual_fwd(h, None, conv._stack["trans"], mask, conv.z_ratio, act, T, m)
            else:
                # gns[l-1] (+ELU +dropout) is applied by the trans kernel while it loads c_{l-1}; h = its side output
                h = torch.empty((n, H), **f32)
                _dual_fwd(c_prev, None, conv._stack["trans"], mask, conv.z_ratio, act, T, m, gn=(*pending_gn, h))
            a = conv.adj.fwd.spmm(m)
            # conv.gn: statistics + finalize here, the apply (+dropout) rides in the comb kernel's operand load
            g = torch.empty((n, H), **f32)
            gsaved = _GN(conv.gn).stats(a) if acc_fwd is None else _GN(conv.gn).stats_exact(a, acc_fwd[l])
            last = l + 1 == L
            c = jk[:, l * H:(l + 1) * H] if emb.jk else (jk if last else torch.empty((n, H), **f32))
            # the comb kernel's epilogue also leaves the column statistics of c for the GraphNorm(s) that read it
            if unl and last:
                cstat = None  # (nothing normalises the last layer's output: impl/models.py:469)
                _comb_eff_fwd(a, h, conv, mask, c, None, (gsaved, ACT_NONE, pc, conv.call_base, g), labels)
            elif _comb_eff_fwd_ok(conv, labels, H):
                cstat = acc_fwd[L + l] if acc_fwd is not None else \
                    torch.empty((int(lib.glass_comb_eff_fwd_blocks(n, H, labels.cap)), 2, H), dtype=torch.float64, device=dev)
                _comb_eff_fwd(a, h, conv, mask, c, cstat, (gsaved, ACT_NONE, pc, conv.call_base, g), labels)
            else:
                cstat = acc_fwd[L + l] if acc_fwd is not None else \
                    torch.empty((-(-n // st["stat_rows"]), 2, H), dtype=torch.float64, device=dev)
                _dual_fwd(a, h, conv._stack["comb"], mask, conv.z_ratio, ACT_NONE, None, c, cstat,
                          gn=(gsaved, ACT_NONE, pc, conv.call_base, g))
            cstats.append(cstat)
            rec = {"h": h, "T": T, "a": a, "g": g, "gsaved": getattr(gsaved, "saved", gsaved), "c": c, "pc": pc}
            if not last:
                if acc_fwd is not None:   # the next layer's trans kernel derives gns[l]'s coefficients from the sums of c_l
                    nsaved = _PendingStats(torch.empty(4 * H, **f32), cstat, 1, emb.gns[l], REP_DENSE)
                else:
                    nsaved = _GN(emb.gns[l]).finalize([cstat], n)
                rec["nsaved"] = getattr(nsaved, "saved", nsaved)
                pending_gn, c_prev = (nsaved, act, p, conv.call_base + 1), c
            layers.append(rec if keep else None)
        st["jk"], st["layers"] = jk, layers
        if unl:
            return jk, (st if keep else None)  # the raw output of the last layer (no final GraphNorm, no JK)
        gnf = _GN(emb.gns[-1])
        if acc_fwd is not None:  # the readout's first kernel derives the final GraphNorm's coefficients
            st["final_saved"] = _PendingStats(torch.empty(4 * C_out, **f32), acc_fwd[L:] if emb.jk else acc_fwd[2 * L - 1],
                                              L if emb.jk else 1, emb.gns[-1], REP_DENSE)
        else:
            st["final_saved"] = gnf.finalize(cstats if emb.jk else cstats[-1:], n)
        if readout is not None:
            return self._readout(st, jk, *readout), st
        out = torch.empty((n, C_out), **f32)
        gnf.apply(jk, out, st["final_saved"], ACT_NONE, 0.0, 0)
        return out, (st if keep else None)

    def _readout(self, st, jk, pos, pool_mode, head, target, loss_mode):
        lib, gn = _lib.load(), self.emb.gns[-1]
        n, C = jk.shape
        B, Smax = pos.shape
        sws_bytes = int(lib.glass_readout_scatter_ws_bytes(n, B, Smax))  # > 0: beyond the ordered scatter's LDS staging
        sws = ops._scratch(("readout_scatter", n, B, Smax), jk.device, sws_bytes).data_ptr() if sws_bytes > 0 else 0
        K = head.weight.shape[0]
        dev = jk.device
        f32 = dict(dtype=torch.float32, device=dev)
        acc_ro = st.get("gn_exact_readout")  # (the library takes the two-launch form only with the listed pooled rows)
        final = st["final_saved"]  # (kept alive across the call below: the glass_gn_src struct lives in it)
        saved, src = _saved_args(final)
        st["final_saved"] = getattr(final, "saved", final)
        ws = torch.empty(lib.glass_readout_ws_bytes(B, C, K) // 8 + 1, dtype=torch.float64, device=dev)
        pooled, logits = torch.empty((B, C), **f32), torch.empty((B, K), **f32)
        loss, djk = torch.empty((), **f32), torch.empty((n, C), **f32)
        tgt = target.contiguous().to(torch.int64 if loss_mode == 0 else torch.float32)
        labels = st.get("labels")  # the batch's label bytes + unique labeled rows = its pooled rows (same pos)
        if labels is None and not (C % 4 == 0 and jk.stride(0) % 4 == 0):
            # any-width (scalar) form needs the pooled rows marked: the labels came as a z tensor, so mark them from pos here
            labels = BatchLabels(n, pos.numel(), dev)
            labels.load(pos)
        largs = (0, 0, 0) if labels is None else (labels.mask.data_ptr(), labels.rows.data_ptr(), labels.count.data_ptr())
        _check(lib.glass_readout_train_f32(jk.data_ptr(), jk.stride(0), saved, gn.weight.data_ptr(),
                                           gn.mean_scale.data_ptr(), pos.data_ptr(), B, Smax, _lib.POOL_MODES[pool_mode],
                                           head.weight.data_ptr(), head.bias.data_ptr(), tgt.data_ptr(), loss_mode, K,
                                           _one(dev).data_ptr(), pooled.data_ptr(), logits.data_ptr(), loss.data_ptr(),
                                           djk.data_ptr(), djk.stride(0), head.weight.grad.data_ptr(),
                                           head.bias.grad.data_ptr(), st["acc"], gn.weight.grad.data_ptr(),
                                           gn.bias.grad.data_ptr(), gn.mean_scale.grad.data_ptr(), st["acc"], ws.data_ptr(),
                                           n, C, *largs, src, 0 if acc_ro is None else acc_ro.data_ptr(), REP_DENSE, sws,
                                           0 if self.loss_sum is None else self.loss_sum.data_ptr(), _stream()),
               "glass_readout_train_f32")
        st["djk"] = djk
        return loss, logits

    # ---------------------------------------------------------------------------------------------
    def backward(self, st, dout, tail_hook=None, fused_opt=None):
        with ops.rng_scope(st["mask"].device, st.get("rng_words")):  # the masks of THIS pass's forward
            return self._backward(st, dout, tail_hook, fused_opt)

    def _backward(self, st, dout, tail_hook, fused_opt):
        """tail_hook: called once every gradient except the embedding's and emb_gn's has been written (the data-parallel
        step starts the all-reduce of the small gradient bucket there, beside the rest of this backward pass)."""
        emb = self.emb
        n, H, L, p = st["n"], st["H"], st["L"], st["p"]
        act = st["act"]
        dev = st["mask"].device
        f32 = dict(dtype=torch.float32, device=dev)
        mask, jk = st["mask"], st["jk"]
        if st.get("rng_words") is None and (p > 0 or any(rec is not None and rec["pc"] > 0 for rec in st["layers"])):
            ops.check_rng_epoch(dev, st["rng_epoch"], "StackProgram.backward")  # live words: they must be this pass's still
        acc = st["acc"]  # 1: add into the gradient arena; 0: overwrite (every gradient is written exactly once)
        if st.get("unlabeled"):
            djk = dout  # no final GraphNorm: the caller's gradient is the last layer's
        elif "djk" in st:  # the fused readout already went through the final GraphNorm
            djk = st["djk"]
        else:
            djk = torch.empty_like(jk)
            _GN(emb.gns[-1]).bwd(dout, jk, st["final_saved"], djk, ACT_NONE, 0.0, 0, acc=acc)
        dh_next = None   # gradient w.r.t. the input of layer l+1 (= output of gns[l])
        npart = None     # backward column sums of gns[l], accumulated by layer l+1's trans data-gradient epilogue
        pending = []     # weight gradients whose partial sums are written but not yet reduced
        nblk = -(-n // st["stat_rows"])
        f64 = dict(dtype=torch.float64, device=dev)
        for l in range(L - 1, -1, -1):
            conv, rec = emb.convs[l], st["layers"][l]
            last = l + 1 == L
            # gradient w.r.t. the raw conv output c_l
            dc_src = None
            if last:
                dc = djk[:, l * H:(l + 1) * H] if emb.jk else djk
            elif (USE_GN_BWD_IN_COMB and npart.dtype == torch.int64 and _comb_eff_ok(conv, st.get("labels"), H) and
                  _lib.load().glass_comb_eff_bwd_gn_src_supported(n, H)):
                # gns[l]'s backward apply rides in the comb launch's operand loads (glass_gn_bwd_src): no launch, no dc
                m = emb.gns[l]
                ad = djk[:, l * H:(l + 1) * H] if emb.jk else None
                dc_src = _lib.GnBwdSrc(npart.data_ptr(), _rep(npart), dh_next.data_ptr(), dh_next.stride(0), rec["c"].data_ptr(),
                                       rec["c"].stride(0), 0 if ad is None else ad.data_ptr(), 0 if ad is None else ad.stride(0),
                                       rec["nsaved"].data_ptr(), m.weight.data_ptr(), m.mean_scale.data_ptr(),
                                       m.weight.grad.data_ptr(), m.bias.grad.data_ptr(), m.mean_scale.grad.data_ptr(), acc, act,
                                       float(p), conv.call_base + 1)
                dc = dh_next  # (shape carrier)
            else:
                dc = torch.empty((n, H), **f32)
                _GN(emb.gns[l]).bwd_from_stats(dh_next, rec["c"], rec["nsaved"], dc, npart, act, p, conv.call_base + 1,
                                               addend=djk[:, l * H:(l + 1) * H] if emb.jk else None, acc=acc)
            din = torch.empty((n, 2 * H), **f32)  # [d g | d x_]
            # conv.gn's backward column sums come from this kernel's epilogue
            labels = st.get("labels")
            acc_all = st.get("gn_exact")
            if acc_all is not None:
                gpart = acc_all[2 * l]   # exact accumulators: the sums are final when the kernel is, no finalize launch
            if acc_all is not None and _comb_eff_ok(conv, labels, H):
                _comb_eff_bwd(dc, conv, mask, din, rec["g"], rec["h"], pending, acc,
                              (gpart, rec["a"], rec["gsaved"], conv.gn.mean_scale, ACT_NONE, rec["pc"], conv.call_base), labels,
                              dsrc_gn=dc_src)
            elif acc_all is not None:
                _dual_bwd(dc, None, conv._stack["comb"], mask, conv.z_ratio, ACT_NONE, 2 * H, None, din, rec["g"], rec["h"],
                          pending, acc, gn=(gpart, rec["a"], rec["gsaved"], conv.gn.mean_scale, ACT_NONE, rec["pc"], conv.call_base))
            elif _comb_eff_ok(conv, labels, H):
                gpart = torch.empty((int(_lib.load().glass_comb_eff_blocks(n, H, labels.cap)), 2, H), **f64)
                _comb_eff_bwd(dc, conv, mask, din, rec["g"], rec["h"], pending, acc,
                              (gpart, rec["a"], rec["gsaved"], conv.gn.mean_scale, ACT_NONE, rec["pc"], conv.call_base), labels)
            else:
                gpart = torch.empty((nblk, 2, H), **f64)
                _dual_bwd(dc, None, conv._stack["comb"], mask, conv.z_ratio, ACT_NONE, 2 * H, None, din, rec["g"], rec["h"],
                          pending, acc, gn=(gpart, rec["a"], rec["gsaved"], conv.gn.mean_scale, ACT_NONE, rec["pc"], conv.call_base))
            da = torch.empty((n, H), **f32)
            _GN(conv.gn).bwd_from_stats(din[:, :H], rec["a"], rec["gsaved"], da, gpart, ACT_NONE, rec["pc"], conv.call_base,
                                        acc=acc)
            dm = conv.adj.bwd.spmm(da)
            dh = torch.empty((n, H), **f32)
            # layer 0 on the table path: the epilogue applies emb_gn's dropout mask (call id 1); layers above: the
            # epilogue accumulates the backward column sums of gns[l-1], whose output this gradient belongs to
            drop = (p, 1) if (l == 0 and "emb_table" in st) else None
            gn = None
            if l > 0:
                below, cb = st["layers"][l - 1], emb.convs[l - 1]
                npart = acc_all[2 * l - 1] if acc_all is not None else torch.empty((nblk, 2, H), **f64)
                gn = (npart, below["c"], below["nsaved"], emb.gns[l - 1].mean_scale, act, p, cb.call_base + 1)
            _dual_bwd(dm, rec["T"], conv._stack["trans"], mask, conv.z_ratio, act, H, din[:, H:], dh, rec["h"], None, pending,
                      acc, drop, gn)
            dh_next = dh
            st["layers"][l] = None  # release this layer's activations
        W = emb.input_emb.weight
        self.applied_optimizer = False
        if st.get("unlabeled"):
            # no GraphNorm behind the embedding: the selection product IS the table's gradient (layer 0's data-gradient
            # epilogue applied the embedding's dropout mask already)
            _reduce_pending(pending, None)
            if tail_hook is not None:
                tail_hook()
            sel = st["emb_table"]
            if acc:
                W.grad.add_(sel.op.spmm(dh_next))
            else:
                sel.op.spmm(dh_next, out=W.grad)
            return
        gn0 = emb.emb_gn
        fused_tail = "emb_table" in st and USE_FUSED_TAIL
        # the deferred weight-gradient reductions — and, on the table path, the selection product of the embedding
        # backward in the same launch (both only wait for the end of the chain above)
        prod = _reduce_pending(pending, (st["emb_table"], dh_next) if fused_tail else None)
        if tail_hook is not None:
            tail_hook()
        if fused_tail:
            # the selection product's partial-row sums, the table backward and — with fused_opt (optim.FlatAdam) — Adam over
            # the whole arena as ONE launch
            sel, arena = st["emb_table"], emb._glass_arena
            offs = [arena.offset_of(t) for t in (W, gn0.weight, gn0.bias, gn0.mean_scale)]
            opt_ok = fused_opt is not None and all(o is not None for o in offs) and fused_opt.fusable() and acc == 0
            G, ws, rrows, n_red = prod
            oargs = (*fused_opt.fused_args(), *offs) if opt_ok else (0, 0, 0, 0, 0, 0, 0.0, 0.0, 0.0, 0.0, 0, 0, 0, 0, 0)
            _check(_lib.load().glass_embed_norm_bwd_adam_f32(G.data_ptr(), W.data_ptr(), W.shape[0], sel.op.rowptr.data_ptr(),
                                                             gn0.weight.data_ptr(), gn0.mean_scale.data_ptr(),
                                                             st["emb_saved"].data_ptr(), W.grad.data_ptr(), acc,
                                                             gn0.weight.grad.data_ptr(), gn0.bias.grad.data_ptr(),
                                                             gn0.mean_scale.grad.data_ptr(), acc, H, ws, rrows, n_red,
                                                             *oargs, _stream()),
                   "glass_embed_norm_bwd_adam_f32")
            self.applied_optimizer = opt_ok
            return
        if "emb_table" in st:
            sel = st["emb_table"]
            G = sel.op.spmm(dh_next)  # [V,H]: per table row, the sum of its nodes' (masked) gradients, on K1
            _check(_lib.load().glass_embed_norm_bwd_f32(G.data_ptr(), W.data_ptr(), W.shape[0], sel.op.rowptr.data_ptr(),
                                                        gn0.weight.data_ptr(), gn0.mean_scale.data_ptr(),
                                                        st["emb_saved"].data_ptr(), W.grad.data_ptr(), acc,
                                                        gn0.weight.grad.data_ptr(), gn0.bias.grad.data_ptr(),
                                                        gn0.mean_scale.grad.data_ptr(), acc, H, _stream()),
                   "glass_embed_norm_bwd_f32")
            return
        dh0 = torch.empty((n, H), **f32)
        _GN(gn0).bwd(dh_next, st["h0"], st["emb_saved"], dh0, ACT_NONE, p, 1, acc=acc)
        # embedding backward: dW (+)= S^T @ dh0 on K1
        if acc:
            W.grad.add_(emb._selection(st["x_flat"]).op.spmm(dh0))
        else:
            emb._selection(st["x_flat"]).op.spmm(dh0, out=W.grad)


    # ---------------------------------------------------------------------------------------------
    def written_params(self, head):
        """Every parameter whose gradient loss_and_grads writes (exactly once per call)."""
        emb = self.emb
        if not hasattr(emb, "emb_gn"):  # the unlabeled pre-training stack: head gradients are the pair head's business
            out = [emb.input_emb.weight]
            for gn in [c.gn for c in emb.convs] + list(emb.gns):
                out += [gn.weight, gn.bias, gn.mean_scale]
            for c in emb.convs:
                out += [c.trans_fn.weight, c.trans_fn.bias, c.comb_fn.weight, c.comb_fn.bias]
            return out
        out = [emb.input_emb.weight, head.weight, head.bias]
        for gn in [emb.emb_gn] + [c.gn for c in emb.convs] + list(emb.gns):
            out += [gn.weight, gn.bias, gn.mean_scale]
        for c in emb.convs:
            for lin in list(c.trans_fns) + list(c.comb_fns):
                out += [lin.weight, lin.bias]
        return out

    def loss_and_grads(self, x_flat, z, edge_index, edge_weight, pos, pool_mode, head, target, loss_mode, overwrite=False,
                       tail_hook=None, labels=None, fused_opt=None):
        """One training pass WITHOUT the autograd tape: forward, fused readout, backward; every parameter gradient
        (stack, final GraphNorm, head) is accumulated into the gradient arena — or, with overwrite=True, stored over
        whatever is there (no zero-fill of the arena needed when written_params() covers it).  Returns (loss, logits)."""
        with torch.no_grad():
            (loss, logits), st = self.forward(x_flat, z, edge_index, edge_weight, True,
                                              readout=(pos, pool_mode, head, target, loss_mode), acc=0 if overwrite else 1,
                                              labels=labels)
            self.backward(st, None, tail_hook, fused_opt if overwrite else None)
        return loss, logits


_ones = {}


def _one(device):
    t = _ones.get(device)
    if t is None:
        t = _ones[device] = torch.ones((), dtype=torch.float32, device=device)
    return t


def step_supported(model, loss_fn):
    """True when `model` (models.GLASS) + loss can run as StackProgram.loss_and_grads: one feature channel, the stack
    program's own conditions, pooling sum|mean|size straight from the padded node matrix, a bare Linear head whose
    gradients live in the arena, a fusable loss."""
    import torch.nn as nn
    from . import losses
    from .models import EmbZGConv, PoolModule
    emb = getattr(model, "conv", None)
    if not (USE_READOUT and isinstance(emb, EmbZGConv) and StackProgram.supported(emb) and emb.training):
        return False
    pool, head = model.pools[0], model.preds[0]
    if not (isinstance(pool, PoolModule) and pool.trans_fn is None and pool.mode in ("sum", "mean", "size")):
        return False
    if not (type(head) is nn.Linear and head.bias is not None and head.weight.grad is not None and
            head.bias.grad is not None and losses.fusable_mode(loss_fn) is not None):
        return False
    C = emb.gns[-1].weight.shape[0]
    return head.weight.shape[1] == C and bool(_lib.load().glass_readout_supported(C, head.weight.shape[0],
                                                                                  _lib.POOL_MODES[pool.mode]))


def covers_arena(model, arena):
    """loss_and_grads writes the gradient of every parameter in the arena -> overwrite mode needs no zero-fill."""
    prog = _program(model.conv)
    return {id(p) for p in arena.params} <= {id(p) for p in prog.written_params(model.preds[0])}


def _program(emb):
    prog = emb.__dict__.get("_glass_stack_prog")
    if prog is None:
        prog = emb.__dict__["_glass_stack_prog"] = StackProgram(emb)
    return prog


def loss_and_grads(model, loss_fn, x, edge_index, edge_weight, pos, z, target, overwrite=False, tail_hook=None, labels=None,
                   fused_opt=None):
    """fused_opt (optim.FlatAdam over the model's arena, overwrite mode only): the optimizer step rides in the backward's
    last launch when it can — `applied_optimizer(model)` tells whether it did (else call fused_opt.step())."""
    """(loss, logits) of model(x, ..., pos, z) under loss_fn, gradients accumulated in place (see step_supported).
    z = "pos": label the nodes listed in pos (what utils.MaxZOZ(x, pos) would mark) without materialising z."""
    from . import losses
    emb = model.conv
    if x.dim() != 3 or x.shape[1] != 1 or x.shape[2] != 1:
        raise NotImplementedError("one integer feature per node (x of shape [N,1,1])")
    x_flat = x.reshape(x.shape[0])
    if x_flat.dtype != torch.int64:
        x_flat = x_flat.to(torch.int64)
    prog = _program(emb)
    pos = pos.contiguous()
    if pos.dtype != torch.int64:
        pos = pos.to(torch.int64)
    if isinstance(z, str):
        if z != "pos":
            raise ValueError("z must be a tensor, None or 'pos'")
        z = ("pos", pos)
    return prog.loss_and_grads(x_flat, z, edge_index, edge_weight, pos, model.pools[0].mode, model.preds[0], target,
                               losses.fusable_mode(loss_fn), overwrite, tail_hook, labels, fused_opt)


def applied_optimizer(model):
    return bool(getattr(_program(model.conv), "applied_optimizer", False))


class StackFn(torch.autograd.Function):
    """autograd node of a whole EmbZGConv.  The parameters are listed as inputs only so that the node is
    differentiable; their gradients are accumulated in place (arena) and None is returned for them."""
    @staticmethod
    def forward(ctx, prog, x_flat, z, edge_index, edge_weight, *params):
        keep = any(ctx.needs_input_grad)
        out, st = prog.forward(x_flat, z, edge_index, edge_weight, keep, snapshot_rng=True)
        ctx.prog, ctx.st, ctx.n_in = prog, st, 5 + len(params)
        return out

    @staticmethod
    def backward(ctx, dout):
        st, ctx.st = ctx.st, None
        if st is None:
            raise RuntimeError("StackFn: backward called twice (activations are released after the first pass)")
        dout, _ = ops._rows(dout)
        ctx.prog.backward(st, dout)
        return (None, ) * ctx.n_in


def run(emb, x_flat, z, edge_index, edge_weight):
    prog = _program(emb)
    params = [p for p in emb.parameters() if p.requires_grad]
    return StackFn.apply(prog, x_flat, z, edge_index, edge_weight, *params)
