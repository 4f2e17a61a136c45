"""Flat parameter / gradient arena for a GLASS model.

All parameters become views into ONE contiguous fp32 buffer and all gradients views into a second
one (what DDP/FSDP call flattening).  That buys, on a path whose step is a chain of short dependent kernels:
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
    """[(module, kind, first, second)] Linear pairs to lay out adjacently: index 1 (labeled) before index 0.
    first = None: a SINGLE Linear (the unlabeled layers of the pre-training path, MyGCNConv: reference
    impl/models.py:361-397) laid out as the second half of a pair whose first half is an all-zero block of the flat buffer —
    with z_ratio = 1 and no labeled row the pair kernels then compute exactly the single layer (the first half is weighted
    by 1 - z = 0 in the forward, its data gradient term is 0, its weight gradient 0): the whole fused stack serves the
    unlabeled model without a second kernel family."""
    from .models import GLASSConv, MyGCNConv
    out = []
    for m in model.modules():
        if isinstance(m, GLASSConv):
            out.append((m, "trans", m.trans_fns[1], m.trans_fns[0]))
            out.append((m, "comb", m.comb_fns[1], m.comb_fns[0]))
        elif isinstance(m, MyGCNConv):
            out.append((m, "trans", None, m.trans_fn))
            out.append((m, "comb", None, m.comb_fn))
    return out


BIG_PARAM_ELEMS = 1 << 18  # parameters from 1 MB up (the use_nodeid embedding table) form the "big" bucket

# what a ParamArena (and the loops of train.py) hang on a model and its submodules: per-process runtime objects
_RUNTIME_ATTRS = ("_glass_grad_bucket", "_glass_arena", "_glass_stack_prog", "_stack", "_stack_eff", "_direct_grad", "_sel_cache",
                  "_chan_cache", "_glass_train_steps", "_glass_eval_graphs", "_glass_adopted_adam")


def drop_captured_graphs(model):
    """Forget every hipGraph cached on `model` (train.py: `_glass_train_steps`, `_glass_eval_graphs`).  A captured graph holds the
    parameters' device ADDRESSES: whenever a parameter's storage moves — an arena built around an existing model, a
    re-allocated .data copied back by reattach() — a cached graph would keep replaying on the old, freed storage (an
    evaluation before the first training epoch is the common case: test() caches its graph on the per-parameter storage,
    the first train() then adopts the optimizer and builds the arena).  The next call captures afresh."""
    for name in ("_glass_train_steps", "_glass_eval_graphs"):
        c = model.__dict__.get(name)
        if c:
            c.clear()


def strip_runtime(model):
    """Make `model` a plain module again: every parameter gets storage of its own (a copy of its current values), gradients
    are dropped, and every runtime attachment — arena, stacked weight views and operand images, step / evaluation graphs, the
    adopted optimizer engine — is removed.  The next training call builds fresh ones.  Used by the models' __deepcopy__: a
    deep copy of an arena would keep the ORIGINAL's device pointers in its launch argument arrays."""
    with torch.no_grad():
        for p in model.parameters():
            p.data = p.data.clone()
            p.grad = None
    for mod in model.modules():
        for name in _RUNTIME_ATTRS:
            mod.__dict__.pop(name, None)
    return model


def deepcopy_plain(module, memo):
    """__deepcopy__ body of the model classes: an ordinary deep copy of the module tree, then strip_runtime on the copy."""
    import copy
    new = module.__class__.__new__(module.__class__)
    memo[id(module)] = new
    for k, v in module.__dict__.items():
        new.__dict__[k] = None if k in _RUNTIME_ATTRS else copy.deepcopy(v, memo)
    for k in _RUNTIME_ATTRS:
        new.__dict__.pop(k, None)
    return strip_runtime(new)


class ParamArena(FlatGradBucket):
    def __init__(self, model, big_elems=BIG_PARAM_ELEMS):
        from .models import GraphNorm, GLASSConv
        params = [p for p in model.parameters() if p.requires_grad]
        if not params:
            raise ValueError("model has no trainable parameters")
        dev, dtype = params[0].device, params[0].dtype
        # layout: [layer / head parameters][embedding-sized parameters].  The second group is the "big" bucket of the
        # data-parallel exchange (dist.GradExchange): reduce-scatter + sharded Adam + all-gather instead of all-reduce.
        # The big bucket starts with the parameters of the GraphNorm right behind such an embedding (emb_gn): in the
        # backward pass their gradients are written together with the table's, after every layer / head gradient is
        # final — so the small bucket can be all-reduced while that tail of the backward still runs (step.TrainStep).
        from .models import EmbZGConv
        emb_w = {id(m.weight) for m in model.modules() if isinstance(m, torch.nn.Embedding)}
        big = [p for p in params if id(p) in emb_w and p.numel() >= big_elems]
        tail = []
        for m in model.modules():
            if isinstance(m, EmbZGConv) and any(m.input_emb.weight is q for q in big) and getattr(m, "emb_gn", None) is not None:
                tail += [p for p in m.emb_gn.parameters() if p.requires_grad]
        big = tail + big if big else []
        params = [p for p in params if not any(p is q for q in big)] + big
        order, seen, groups = [], set(), []
        for mod, kind, l1, l0 in _pairs(model):
            if l1 is None:  # a single Linear: [zero block; weight]
                for b in (l0.weight, l0.bias):
                    if b is not None and id(b) not in seen:
                        groups.append((mod, kind, None, b))
                        seen.add(id(b))
                continue
            for a, b in ((l1.weight, l0.weight), (l1.bias, l0.bias)):
                if a.shape == b.shape and id(a) not in seen and id(b) not in seen:
                    groups.append((mod, kind, a, b))
                    seen.update((id(a), id(b)))
        grouped = {id(a): (a, b) for _, _, a, b in groups if a is not None}
        second = {id(b) for _, _, a, b in groups if a is not None}
        single = {id(b) for _, _, a, b in groups if a is None}
        offsets, off = {}, 0
        stack_start = {}  # id(second half) -> element offset of the stacked [first; second] block
        for p in params:
            if id(p) in second:
                continue
            off = (off + 3) // 4 * 4  # 16-B alignment of every group start
            if id(p) in single:
                stack_start[id(p)] = off
                off += p.numel()          # the zero half (never a parameter; Adam leaves zeros with zero gradients alone)
            offsets[id(p)] = off
            off += p.numel()
            if id(p) in grouped:
                b = grouped[id(p)][1]
                stack_start[id(b)] = offsets[id(p)]
                offsets[id(b)] = off  # exactly adjacent: [a; b] is one stacked tensor
                off += b.numel()
        from . import dist as gdist
        self.big_start = None
        if big:
            self.big_start = offsets[id(big[0])]
            quantum = 4 * gdist.world_size()  # every rank's shard of the big bucket: whole float4s
            total = self.big_start + -(-(off - self.big_start) // quantum) * quantum
        else:
            total = (off + 3) // 4 * 4
            self.big_start = total
        self._exchange = None
        # the optimizer updates the big bucket shard-wise (reduce-scatter + all-gather): only an optimizer that knows how
        # (optim.FlatAdam) switches this on; otherwise all_reduce_mean() leaves the mean gradient of EVERY parameter in
        # `flat`, the FlatGradBucket contract any torch optimizer relies on
        self.shard_optimizer = False
        self.model = model
        model._glass_grad_bucket = self  # dist.bucket_for(model) -> the arena (train.train's all-reduce hook)
        self.params = params
        self.flat_param = torch.zeros(total, dtype=dtype, device=dev)
        self.flat = torch.zeros(total, dtype=dtype, device=dev)  # gradients (FlatGradBucket API)
        self._offsets = offsets
        drop_captured_graphs(model)  # every parameter's storage moves below
        with torch.no_grad():
            for p in params:
                o = offsets[id(p)]
                view = self.flat_param[o:o + p.numel()].view_as(p)
                view.copy_(p.data)
                p.data = view
                p.grad = self.flat[o:o + p.numel()].view_as(p)
        # stacked views for the fused Linear pairs
        from .models import MyGCNConv
        for mod in model.modules():
            if isinstance(mod, (GLASSConv, MyGCNConv)):
                mod._stack = {}
                mod._stack_eff = {}
            if isinstance(mod, MyGCNConv):
                mod.z_ratio, mod.dropout = 1.0, 0.0  # (what the pair kernels are told: only the second half counts)
            if isinstance(mod, GraphNorm):
                mod._direct_grad = True
        stacks = {}
        for mod, kind, a, b in groups:
            o = stack_start[id(b)]
            n2 = 2 * b.numel()
            shape = (2 * b.shape[0], ) + tuple(b.shape[1:])
            stacks.setdefault((id(mod), kind), [mod, kind]).append(
                (self.flat_param[o:o + n2].view(shape), self.flat[o:o + n2].view(shape)))
        self._packs = []  # (src weight view, dst image, NT, KT, transposed)
        from . import _lib
        for (_, kind), (mod, _k, *views) in stacks.items():
            if len(views) == 2:  # weight and bias both stackable
                (W, dW), (b, db) = views
                O, K = W.shape  # [2H, K]
                from . import ops
                if ops.dual_linear_supported(O // 2) and _lib.load().glass_dual_linear_layout(O // 2) == 2:
                    # hidden <= 32 (dense_narrow.hip): the kernels read the row-major weight itself — no packed images
                    mod._stack[kind] = (W, b, dW, db, W, W)
                elif O % 64 == 0 and K % 64 == 0 and ops.dual_linear_supported(O // 2):
                    # MFMA images of W (forward operand, [NT=2H][KT=K]) and of W^T (data-gradient operand,
                    # [NT=K][KT=2H]); refreshed by ONE launch per training forward (Adam changes W in between).
                    # flags = transposed | layout << 1 (layout 0: wave16 images; tiled kernels: forward operand
                    # paired = 1, data-gradient operand plain = 2, or split = 3 for the 128-wide output of hidden 128's
                    # trans pair: both halves of the product side by side in one 256-slot tile)
                    tiled = _lib.load().glass_dual_linear_layout(O // 2) == 1
                    # data-gradient operand (NT = K output columns): the library names its layout; layout 4 (comb pair
                    # at hidden 256 / 512) appends the effective weight of unlabeled rows -> 1.5 x the floats, and needs
                    # the pair's z_ratio
                    dlay = int(_lib.load().glass_dual_linear_dgrad_layout(O // 2, K))
                    flay = int(_lib.load().glass_dual_linear_fwd_layout(O // 2, K))  # 5: same appendix for the forward
                    # image sizes from the library (appendix of layouts 4 / 5; the cut image behind tiled layouts)
                    Wimg = torch.empty(int(_lib.load().glass_dense_image_floats(O, K, 0 | (flay << 1))), dtype=W.dtype, device=W.device)
                    WTimg = torch.empty(int(_lib.load().glass_dense_image_floats(K, O, 1 | (dlay << 1))), dtype=W.dtype, device=W.device)
                    self._packs.append((W, Wimg, O, K, 0 | (flay << 1), mod))
                    self._packs.append((W, WTimg, K, O, 1 | (dlay << 1), mod))
                    mod._stack[kind] = (W, b, dW, db, Wimg, WTimg)
                    if kind == "comb" and K == O and _lib.load().glass_comb_eff_supported(O // 2):
                        # hidden 64: the comb pair's effective per-label weights (layouts 6 / 7: unlabeled-row image, then
                        # labeled-row image) for glass_comb_eff_fwd/bwd_f32 — taken when the batch's labeled rows are listed
                        lay2 = int(_lib.load().glass_comb_eff_dgrad_layout2(O // 2))
                        We = torch.empty_like(Wimg)
                        WTe = torch.empty(WTimg.numel() * (2 if lay2 else 1), dtype=W.dtype, device=W.device)
                        self._packs.append((W, We, O, K, 0 | (int(_lib.load().glass_comb_eff_fwd_layout(O // 2)) << 1), mod))
                        self._packs.append((W, WTe, K, O, 1 | (7 << 1), mod))
                        if lay2:  # the staged backward's own column order: a second pair of images behind the first
                            self._packs.append((W, WTe[WTimg.numel():], K, O, 1 | (lay2 << 1), mod))
                        mod._stack_eff = {"comb": (We, WTe)}
                    elif kind == "comb" and K == O and _lib.load().glass_comb_eff_fwd_supported(O // 2):
                        # hidden 128: the forward alone runs in effective-weight form (layout 8 image of W_unl | W_lab,
                        # 2H * H floats each); its backward stays on the tiled kernels
                        We = torch.empty(W.numel(), dtype=W.dtype, device=W.device)
                        self._packs.append((W, We, O, K, 0 | (int(_lib.load().glass_comb_eff_fwd_layout(O // 2)) << 1), mod))
                        mod._stack_eff = {"comb": (We, None)}
                else:
                    mod._stack[kind] = (W, b, dW, db)
        import numpy as np
        self._pack_args = (np.array([w.data_ptr() for w, *_ in self._packs], dtype=np.uint64),
                           np.array([p[1].data_ptr() for p in self._packs], dtype=np.uint64),
                           np.array([p[1].numel() for p in self._packs], dtype=np.int64),  # what each image buffer holds
                           np.array([p[2] for p in self._packs], dtype=np.int64),
                           np.array([p[3] for p in self._packs], dtype=np.int64),
                           np.array([p[4] for p in self._packs], dtype=np.int32),
                           np.zeros(len(self._packs), dtype=np.float32), len(self._packs))
        from .models import EmbZGConv, EmbGConv
        for mod in model.modules():
            if isinstance(mod, (EmbZGConv, EmbGConv)):
                mod._glass_arena = self  # EmbZGConv.forward refreshes the images once per training forward
        self.refresh_transposes()

    def refresh_transposes(self, rng_state=None, table=None, zero=None, head=None):
        """Re-pack every stacked weight into the operand images of the fused dense kernels (forward: W,
        data gradient: W^T): one launch for the whole model.  rng_state (the device-resident dropout counter,
        ops.rng_state): advanced by the same launch — the two once-per-step prologue jobs share it."""
        # head: a stack.BatchLabels with an epoch source (set_epoch) — the label launch of the step's batch rides in this
        # launch (glass_step_head_f32: prologue || labels, the batch named by the device-resident cursor)
        src, dst, cap, nt, kt, tr, zr, k = self._pack_args
        from . import _lib, ops
        for i, p in enumerate(self._packs):  # the pairs' CURRENT z_ratio (the kernels receive the live value as well)
            zr[i] = float(getattr(p[5], "z_ratio", 0.0))
        if k == 0 and rng_state is not None and table is None and zero is None:
            ops.rng_advance(rng_state.device)
        # table = (W, V, class_rowptr, emb_gn module, saved[4H], table-or-None): emb_gn's statistics through the embedding
        # table ride in the (first) pack launch — glass_step_prologue_f32
        for i in range(0, max(k, 1), 16):
            n = min(16, k - i)
            rng = rng_state.data_ptr() if (rng_state is not None and i == 0) else 0
            st = torch.cuda.current_stream().cuda_stream
            if (table is not None or zero is not None or head is not None) and i == 0:
                # zero = an int64 tensor the same launch zero-fills (the step's exact GraphNorm accumulators)
                zargs = (0, 0) if zero is None else (zero.data_ptr(), zero.numel())
                if table is not None:
                    W, V, rowptr, gn, saved, tab = table
                    targs = (W.data_ptr(), V, rowptr.data_ptr(), gn.weight.data_ptr(), gn.bias.data_ptr(), gn.mean_scale.data_ptr(),
                             float(gn.eps), saved.data_ptr(), 0 if tab is None else tab.data_ptr(), W.shape[1])
                else:
                    targs = (0, 0, 0, 0, 0, 0, 0.0, 0, 0, 0)
                if head is not None:
                    rc = _lib.load().glass_step_head_f32(src[i:].ctypes.data, dst[i:].ctypes.data, cap[i:].ctypes.data, nt[i:].ctypes.data,
                                                         kt[i:].ctypes.data, tr[i:].ctypes.data, zr[i:].ctypes.data, max(n, 0), rng,
                                                         *targs, *zargs, *head.head_args(), st)
                    _lib.check(rc, "glass_step_head_f32")
                    head.loaded = True
                else:
                    rc = _lib.load().glass_step_prologue_f32(src[i:].ctypes.data, dst[i:].ctypes.data, cap[i:].ctypes.data, nt[i:].ctypes.data,
                                                             kt[i:].ctypes.data, tr[i:].ctypes.data, zr[i:].ctypes.data, max(n, 0), rng,
                                                             *targs, *zargs, st)
                    _lib.check(rc, "glass_step_prologue_f32")
            elif n > 0:
                rc = _lib.load().glass_dense_pack_batch_f32(src[i:].ctypes.data, dst[i:].ctypes.data, cap[i:].ctypes.data, nt[i:].ctypes.data,
                                                            kt[i:].ctypes.data, tr[i:].ctypes.data, zr[i:].ctypes.data, n, rng, st)
                _lib.check(rc, "glass_dense_pack_batch_f32")
            else:
                continue
            if rng_state is not None and i == 0:
                ops.note_rng_advance(rng_state.device)

    def offset_of(self, param):
        """Element offset of a parameter inside the flat buffers (None: not in the arena)."""
        return self._offsets.get(id(param))

    def attached(self):
        base_p, base_g = self.flat_param.untyped_storage().data_ptr(), self.flat.untyped_storage().data_ptr()
        return all(p.data.untyped_storage().data_ptr() == base_p and p.grad is not None
                   and p.grad.untyped_storage().data_ptr() == base_g for p in self.params)

    def reattach(self):
        """Point every parameter's .grad (and .data, if something re-allocated it) back into the flat buffers: what
        `optimizer.zero_grad()` with torch's default set_to_none=True (reference impl/train.py:11) undoes each step.  The
        parameter VALUES are kept (copied into the arena when .data had moved); a gradient tensor living elsewhere is dropped
        — it is rewritten before it is read."""
        base_p = self.flat_param.untyped_storage().data_ptr()
        base_g = self.flat.untyped_storage().data_ptr()
        if any(p.data.untyped_storage().data_ptr() != base_p for p in self.params):
            drop_captured_graphs(self.model)  # (a moved .data: cached graphs hold the old addresses)
        with torch.no_grad():
            for p in self.params:
                o = self._offsets[id(p)]
                if p.data.untyped_storage().data_ptr() != base_p:
                    view = self.flat_param[o:o + p.numel()].view_as(p)
                    view.copy_(p.data)
                    p.data = view
                if p.grad is None or p.grad.untyped_storage().data_ptr() != base_g:
                    p.grad = self.flat[o:o + p.numel()].view_as(p)

    @property
    def exchange(self):
        """dist.GradExchange over this arena (built on first use, once a process group with > 1 rank exists)."""
        from . import dist as gdist
        if self._exchange is None and gdist.is_distributed():
            self._exchange = gdist.GradExchange(self.flat, self.flat_param, self.big_start)
        return self._exchange

    def sharded(self):
        """True when the big bucket is exchanged by reduce-scatter: the optimizer must then update the small bucket
        and this rank's shard only, and call exchange.gather_params()."""
        ex = self.exchange
        return self.shard_optimizer and ex is not None and ex.has_big

    def adopt_grad_storage(self, flat):
        """Move the gradient arena into `flat` (same length and dtype; e.g. an allocation the peers of a one-shot exchange map:
        glass_amd/peer.py): every .grad becomes a view of it.  Captured graphs hold the old addresses and are dropped."""
        if flat.numel() != self.flat.numel() or flat.dtype != self.flat.dtype or flat.device != self.flat.device:
            raise ValueError("adopt_grad_storage: the new buffer must match the gradient arena")
        drop_captured_graphs(self.model)
        for mod in self.model.modules():  # step programs hold launch arguments with the old gradient addresses
            mod.__dict__.pop("_glass_stack_prog", None)
        self.flat = flat
        self._exchange = None
        with torch.no_grad():
            for p in self.params:
                o = self._offsets[id(p)]
                p.grad = self.flat[o:o + p.numel()].view_as(p)
        # (stacked gradient views of the fused pairs point into the old buffer: rebuild what hangs on the model)
        for mod in self.model.modules():
            st = getattr(mod, "_stack", None)
            if isinstance(st, dict):
                for kind, tup in list(st.items()):
                    W, b, dW, db = tup[:4]
                    oW = W.data_ptr() - self.flat_param.data_ptr()
                    ob = b.data_ptr() - self.flat_param.data_ptr()
                    ndW = self.flat[oW // 4:oW // 4 + W.numel()].view_as(W)
                    ndb = self.flat[ob // 4:ob // 4 + b.numel()].view_as(b)
                    st[kind] = (W, b, ndW, ndb) + tuple(tup[4:])

    def all_reduce_mean(self):
        if getattr(self, "_peer", None) is not None:
            return  # the one-shot exchange rides in the optimizer's launch (glass_amd/peer.py)
        ex = self.exchange
        if ex is None:
            return
        if ex.has_big and not self.shard_optimizer:
            FlatGradBucket.all_reduce_mean(self)  # one all-reduce over the whole flat buffer
            return
        ex.reduce_small()
        ex.reduce_big()

    def zero(self):
        if not self.attached():
            raise RuntimeError("ParamArena detached (model.to()/zero_grad(set_to_none=True) after flattening?)")
        self.flat.zero_()
