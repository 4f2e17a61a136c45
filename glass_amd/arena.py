"""Flat parameter / gradient arena for a GLASS model.

All parameters become views into ONE contiguous fp32 buffer and all gradients views into a second
one (what DDP/FSDP call flattening).  That buys, on a path whose step is ~100 short kernels:
  * one fused Adam launch over the whole model (glass_amd.optim.FlatAdam),
  * one memset to zero the gradients, one all-reduce for data parallelism (dist.FlatGradBucket),
  * the two weight sets of a GLASSConv Linear pair laid out back to back ([W1; W0], [b1 | b0]), so
    the stacked [2H,H] weight of the fused `x -> [f1 | f0]` GEMM is a VIEW (no torch.cat per step),
    and the weight-gradient kernel accumulates straight into the gradient arena.
state_dict keys and shapes are untouched (Parameters stay where the reference put them).
Build the arena AFTER model.to(device): Module.to() re-allocates every parameter separately.
"""
import torch

from .dist import FlatGradBucket


def _pairs(model):
    """[(first, second)] Linear pairs to lay out adjacently: index 1 (labeled) before index 0."""
    from .models import GLASSConv
    out = []
    for m in model.modules():
        if isinstance(m, GLASSConv):
            out.append((m, "trans", m.trans_fns[1], m.trans_fns[0]))
            out.append((m, "comb", m.comb_fns[1], m.comb_fns[0]))
    return out


class ParamArena(FlatGradBucket):
    def __init__(self, model):
        from .models import GraphNorm, GLASSConv
        params = [p for p in model.parameters() if p.requires_grad]
        if not params:
            raise ValueError("model has no trainable parameters")
        dev, dtype = params[0].device, params[0].dtype
        order, seen, groups = [], set(), []
        for mod, kind, l1, l0 in _pairs(model):
            for a, b in ((l1.weight, l0.weight), (l1.bias, l0.bias)):
                if a.shape == b.shape and id(a) not in seen and id(b) not in seen:
                    groups.append((mod, kind, a, b))
                    seen.update((id(a), id(b)))
        grouped = {id(a): (a, b) for _, _, a, b in groups}
        second = {id(b) for _, _, a, b in groups}
        offsets, off = {}, 0
        for p in params:
            if id(p) in second:
                continue
            off = (off + 3) // 4 * 4  # 16-B alignment of every group start
            offsets[id(p)] = off
            off += p.numel()
            if id(p) in grouped:
                b = grouped[id(p)][1]
                offsets[id(b)] = off  # exactly adjacent: [a; b] is one stacked tensor
                off += b.numel()
        total = (off + 3) // 4 * 4
        self.model = model
        model._glass_grad_bucket = self  # dist.bucket_for(model) -> the arena (train.train's all-reduce hook)
        self.params = params
        self.flat_param = torch.zeros(total, dtype=dtype, device=dev)
        self.flat = torch.zeros(total, dtype=dtype, device=dev)  # gradients (FlatGradBucket API)
        self._offsets = offsets
        with torch.no_grad():
            for p in params:
                o = offsets[id(p)]
                view = self.flat_param[o:o + p.numel()].view_as(p)
                view.copy_(p.data)
                p.data = view
                p.grad = self.flat[o:o + p.numel()].view_as(p)
        # stacked views for the fused Linear pairs
        for mod in model.modules():
            if isinstance(mod, GLASSConv):
                mod._stack = {}
            if isinstance(mod, GraphNorm):
                mod._direct_grad = True
        stacks = {}
        for mod, kind, a, b in groups:
            o = offsets[id(a)]
            n2 = a.numel() + b.numel()
            shape = (2 * a.shape[0], ) + tuple(a.shape[1:])
            stacks.setdefault((id(mod), kind), [mod, kind]).append(
                (self.flat_param[o:o + n2].view(shape), self.flat[o:o + n2].view(shape)))
        self._transposes = []
        for (_, kind), (mod, _k, *views) in stacks.items():
            if len(views) == 2:  # weight and bias both stackable
                (W, dW), (b, db) = views
                WT = torch.empty((W.shape[1], W.shape[0]), dtype=dtype, device=dev)  # refreshed once per step
                mod._stack[kind] = (W, b, dW, db, WT)
                self._transposes.append((W, WT))
        import numpy as np
        k = len(self._transposes)
        self._tp_args = (np.array([w.data_ptr() for w, _ in self._transposes], dtype=np.uint64),
                         np.array([t.data_ptr() for _, t in self._transposes], dtype=np.uint64),
                         np.array([w.shape[0] for w, _ in self._transposes], dtype=np.int64),
                         np.array([w.shape[1] for w, _ in self._transposes], dtype=np.int64), k)
        from .models import EmbZGConv
        for mod in model.modules():
            if isinstance(mod, EmbZGConv):
                mod._glass_arena = self  # EmbZGConv.forward refreshes the transposes once per training forward
        self.refresh_transposes()

    def refresh_transposes(self):
        """W^T of every stacked weight (operand layout of the fused data-gradient kernel): one launch for the
        whole model, called once per training forward because Adam changes the weights in between."""
        src, dst, rows, cols, k = self._tp_args
        if k == 0:
            return
        from . import _lib
        for i in range(0, k, 16):
            n = min(16, k - i)
            rc = _lib.load().glass_transpose_batch_f32(src[i:].ctypes.data, dst[i:].ctypes.data, rows[i:].ctypes.data,
                                                       cols[i:].ctypes.data, n, torch.cuda.current_stream().cuda_stream)
            _lib.check(rc, "glass_transpose_batch_f32")

    def attached(self):
        base_p, base_g = self.flat_param.untyped_storage().data_ptr(), self.flat.untyped_storage().data_ptr()
        return all(p.data.untyped_storage().data_ptr() == base_p and p.grad is not None
                   and p.grad.untyped_storage().data_ptr() == base_g for p in self.params)

    def all_reduce_mean(self):
        from .ops import join_side_streams
        join_side_streams()  # the side-stream weight gradients must have landed in the arena
        super().all_reduce_mean()

    def zero(self):
        from .ops import join_side_streams
        join_side_streams()
        if not self.attached():
            raise RuntimeError("ParamArena detached (model.to()/zero_grad(set_to_none=True) after flattening?)")
        self.flat.zero_()
