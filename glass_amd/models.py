"""GLASS model stack on the MI355X HIP kernels — the host-side mirror of the reference's
`impl/models.py` surface (same class names, constructor arguments, forward signatures,
state_dict keys and error types), so `GLASSTest.py`-style drivers are drop-in
callers.  The arithmetic runs in libglass_hip (ops.py); dense Linears stay on rocBLAS through
torch; there is no CPU path.

Reference map (file:line under /root/reference):
  Seq 10-24 · MLP 27-80 · buildAdj 83-111 · GLASSConv 114-174 · EmbZGConv 177-272 ·
  PoolModule/AddPool/MaxPool/MeanPool/SizePool 275-319 · GLASS 322-355   (impl/models.py)
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

import os

from . import ops, stack
from .graph import CSRAdj, Selection
from .ops import ACT_ELU, ACT_NONE, ACT_RELU
from .utils import batch2pad


USE_STACK = os.environ.get("GLASS_STACK", "1") != "0"  # A/B switch: whole-stack program vs per-op autograd nodes


# ---------------------------------------------------------------------------------------------
class GraphNorm(nn.Module):
    """Whole-graph GraphNorm (PyG 1.7.2 semantics with batch=None): parameters `weight`, `bias`,
    `mean_scale` initialised to 1, 0, 1.  `forward` optionally fuses the ELU and the inverted
    dropout that follow every GraphNorm in the reference (models.py:166,251,258-259)."""
    def __init__(self, in_channels, eps=1e-5):
        super().__init__()
        self.in_channels, self.eps = in_channels, eps
        self.weight = nn.Parameter(torch.ones(in_channels))
        self.bias = nn.Parameter(torch.zeros(in_channels))
        self.mean_scale = nn.Parameter(torch.ones(in_channels))

    def reset_parameters(self):
        nn.init.ones_(self.weight)
        nn.init.zeros_(self.bias)
        nn.init.ones_(self.mean_scale)

    def forward(self, x, batch=None, act=ACT_NONE, p_drop=0.0, call_id=0):
        if batch is not None:
            raise NotImplementedError("GLASS only uses whole-graph GraphNorm (batch=None)")
        return ops.graphnorm(x, self.weight, self.bias, self.mean_scale, self.eps, act, p_drop, call_id,
                             getattr(self, "_direct_grad", False))


def _act_code(activation):
    """ELU(alpha=1) (the driver's choice, GLASSTest.py:143) and ReLU (the reference's constructor default,
    impl/models.py:125,192; the pre-training path, GNNEmb.py:90) are fused into the kernels; any other module runs as a
    torch GPU op."""
    if isinstance(activation, nn.ELU) and activation.alpha == 1.0:
        return ACT_ELU
    if type(activation) is nn.ReLU:
        return ACT_RELU
    return None


class Seq(nn.Module):
    """nn.Sequential whose first module also receives the extra positional / keyword arguments."""
    def __init__(self, modlist):
        super().__init__()
        self.modlist = nn.ModuleList(modlist)

    def forward(self, *args, **kwargs):
        it = iter(self.modlist)
        out = next(it)(*args, **kwargs)
        for m in it:
            out = m(out)
        return out


class Linear(nn.Linear):
    """nn.Linear (same parameters, same state_dict keys) whose weight gradient runs on the split-K MFMA kernel
    (ops.LinearFn) — for layers applied to every node or to a large edge batch."""
    def forward(self, x):
        return ops.linear(x, self)


class MLP(nn.Module):
    """Linear / GraphNorm / Dropout / activation stack with the reference's layer ordering
    (only GNNEmb / GNNSeg use it; GLASSTest's head is a bare nn.Linear)."""
    def __init__(self, input_channels, hidden_channels, output_channels, num_layers, dropout=0, tail_activation=False,
                 activation=nn.ReLU(inplace=True), gn=False):
        super().__init__()

        def tail(width):
            mods = [GraphNorm(width)] if gn else []
            if dropout > 0:
                mods.append(nn.Dropout(p=dropout, inplace=True))
            return mods + [activation]

        dims = [input_channels] + [hidden_channels] * (num_layers - 1) + [output_channels]
        mods = []
        for i in range(num_layers):
            mods.append(Linear(dims[i], dims[i + 1]))
            if i + 1 < num_layers or tail_activation:
                mods += tail(dims[i + 1])
        self.seq = Seq(mods)

    def forward(self, x):
        return self.seq(x)


# ---------------------------------------------------------------------------------------------
_adj_cache = []  # [(edge_index, edge_weight, n, aggr, CSRAdj)] — tensors are held so pointers stay valid


def buildAdj(edge_index, edge_weight, n_node: int, aggr: str):
    """Normalised adjacency for aggr in {mean,sum,gcn} as a device CSR (graph.CSRAdj) instead of the
    reference's COO.  One instance is shared by every layer asking for the same (graph, aggr) —
    the reference builds an identical copy per layer (models.py:154-156)."""
    if aggr not in ("mean", "sum", "gcn"):
        raise NotImplementedError
    for ei, ew, n, a, adj in _adj_cache:
        if ei is edge_index and ew is edge_weight and n == int(n_node) and a == aggr:
            return adj
    adj = CSRAdj(edge_index, edge_weight, n_node, aggr)
    _adj_cache.append((edge_index, edge_weight, int(n_node), aggr, adj))
    del _adj_cache[:-8]
    return adj


class GLASSConv(nn.Module):
    """Labeled message-passing layer: two weight sets (index 1 = labeled, 0 = unlabeled) mixed by
    z_ratio before and after the neighbour aggregation."""
    def __init__(self, in_channels: int, out_channels: int, activation=nn.ReLU(inplace=True), aggr="mean",
                 z_ratio=0.8, dropout=0.2):
        super().__init__()
        self.trans_fns = nn.ModuleList([nn.Linear(in_channels, out_channels), nn.Linear(in_channels, out_channels)])
        self.comb_fns = nn.ModuleList([nn.Linear(in_channels + out_channels, out_channels),
                                       nn.Linear(in_channels + out_channels, out_channels)])
        self.adj = None
        self.activation = activation
        self.aggr = aggr
        self.gn = GraphNorm(out_channels)
        self.z_ratio = z_ratio
        self.dropout = dropout
        self.call_base = 0  # set by EmbZGConv: distinguishes dropout streams of different layers

    def reset_parameters(self):
        for lin in list(self.trans_fns) + list(self.comb_fns):
            lin.reset_parameters()
        self.gn.reset_parameters()

    def forward(self, x_, edge_index, edge_weight, mask, out=None):
        """`out` (optional, fused path only): a preallocated [N,H] view (a slice of the JK buffer) to write into."""
        if self.adj is None:
            self.adj = buildAdj(edge_index, edge_weight, x_.shape[0], self.aggr)
        if mask.dtype != torch.uint8:
            mask = mask.reshape(-1).to(torch.uint8)
        p = self.dropout if self.training else 0.0
        code = _act_code(self.activation)
        stack = getattr(self, "_stack", {})  # set by arena.ParamArena: stacked weight views
        H = x_.shape[1]
        if (code is not None and "trans" in stack and "comb" in stack and len(stack["trans"]) == 6 and
                self.trans_fns[0].weight.shape == (H, H) and ops.dual_linear_supported(H) and ops.USE_FUSED_DENSE):
            # fused dense path: Linear pair + ELU + mix in one MFMA kernel each; no cat, no [N,2H] round trips
            m = ops.dual_linear_mix(x_, None, self.trans_fns[1], self.trans_fns[0], mask, self.z_ratio, code,
                                    stack["trans"])
            a = ops.spmm(self.adj, m)
            g = self.gn(a, p_drop=p, call_id=self.call_base)
            return ops.dual_linear_mix(g, x_, self.comb_fns[1], self.comb_fns[0], mask, self.z_ratio, ACT_NONE,
                                       stack["comb"], out)
        # both weight sets in one GEMM: T = [f1 | f0]
        T = ops.stacked_linear(x_, self.trans_fns[1], self.trans_fns[0], stack.get("trans"))
        if code is None:
            T = self.activation(T)
        m = ops.mix(T, mask, self.z_ratio, ACT_NONE if code is None else code)
        a = ops.spmm(self.adj, m)
        g = self.gn(a, p_drop=p, call_id=self.call_base)
        c = torch.cat((g, x_), dim=-1)
        C = ops.stacked_linear(c, self.comb_fns[1], self.comb_fns[0], stack.get("comb"))
        return ops.mix(C, mask, self.z_ratio, ACT_NONE)


class EmbZGConv(nn.Module):
    """Embedding of the integer node feature + label mask, then `num_layers` GLASSConv layers
    with GraphNorm / activation / dropout between them and optional JK concatenation."""
    def __init__(self, hidden_channels, output_channels, num_layers, max_deg, dropout=0, activation=nn.ReLU(),
                 conv=GLASSConv, gn=True, jk=False, **kwargs):
        super().__init__()
        self.input_emb = nn.Embedding(int(max_deg) + 1, hidden_channels, scale_grad_by_freq=False)
        self.emb_gn = GraphNorm(hidden_channels)
        self.convs = nn.ModuleList()
        self.jk = jk
        widths = [hidden_channels] * (num_layers - 1) + [output_channels]
        for l, w in enumerate(widths):
            layer = conv(in_channels=hidden_channels, out_channels=w, activation=activation, **kwargs)
            layer.call_base = 16 * (l + 1)
            self.convs.append(layer)
        self.activation = activation
        self.dropout = dropout
        if gn:
            self.gns = nn.ModuleList([GraphNorm(hidden_channels) for _ in range(num_layers - 1)])
            self.gns.append(GraphNorm(output_channels + (num_layers - 1) * hidden_channels if jk else output_channels))
        else:
            self.gns = None
        self.reset_parameters()

    def reset_parameters(self):
        self.input_emb.reset_parameters()
        self.emb_gn.reset_parameters()
        for conv in self.convs:
            conv.reset_parameters()
        if self.gns is not None:
            for gn in self.gns:
                gn.reset_parameters()

    def _selection(self, x_flat):
        # x is static per dataset: the selection CSR is cached per feature tensor, keyed on the storage it views (the
        # tensor is kept alive in the entry so the pointer cannot be recycled underneath us).  SEVERAL entries are kept:
        # train / validation / test sets hold their own copies of x, and a captured training step (hipGraph) keeps
        # launching K1 with the raw pointers of ITS entry — evicting that entry when an evaluation comes by with another
        # x would leave the graph reading freed memory.
        key = (x_flat.data_ptr(), x_flat.shape[0], self.input_emb.weight.shape[0])
        cache = self.__dict__.setdefault("_sel_cache", {})
        hit = cache.get(key)
        if hit is None:
            if len(cache) >= 16:
                raise RuntimeError("EmbZGConv: more than 16 distinct node-feature tensors seen by one model; the selection "
                                   "cache keeps them alive for captured graphs — reuse the dataset tensors")
            hit = cache[key] = (x_flat, Selection(x_flat, key[2]))
        return hit[1]

    def forward(self, x, edge_index, edge_weight, z=None):
        n = x.shape[0]
        if x.numel() != n:
            raise NotImplementedError("one integer feature per node (x of shape [N,1])")
        x_flat = x.reshape(n)
        if x_flat.dtype != torch.int64:
            x_flat = x_flat.to(torch.int64)
        if not x_flat.is_contiguous():  # a channel slice of a [N, C, 1] feature tensor: the kernels read a dense int64[N]
            x_flat = x_flat.contiguous()
        if z is not None:
            # reference: mask = (z > 0.5) for whatever z holds (impl/models.py:243-248); the kernels read int64
            if z.numel() != n:
                raise ValueError(f"z must hold one label per node ({n}), got {tuple(z.shape)}")
            z = z.reshape(n)
            if z.dtype != torch.int64:
                z = (z > 0.5).to(torch.int64)
            elif not z.is_contiguous():
                z = z.contiguous()
        if USE_STACK and stack.StackProgram.supported(self):
            # the whole stack as one autograd node (explicit forward / backward program, glass_amd/stack.py)
            return stack.run(self, x_flat, z, edge_index, edge_weight)
        p = self.dropout if self.training else 0.0
        if self.training and (p > 0 or any(c.dropout > 0 for c in self.convs)):
            ops.rng_advance(x.device)  # new dropout masks for this forward/backward pair
            # ... read from a private snapshot of the (seed, step) words by this forward's ops and by their backward, so
            # further training forwards may come before it (NodeEmb over several channels, gradient accumulation)
            with ops.rng_scope(x.device, ops.rng_snapshot(x.device) if torch.is_grad_enabled() else None):
                return self._forward_per_op(x_flat, z, edge_index, edge_weight, p)
        return self._forward_per_op(x_flat, z, edge_index, edge_weight, p)

    def _forward_per_op(self, x_flat, z, edge_index, edge_weight, p):
        n = x_flat.shape[0]
        arena = getattr(self, "_glass_arena", None)
        if arena is not None:
            arena.refresh_transposes()  # operand images (W, W^T) of the fused dense kernels follow the weights
        code = _act_code(self.activation)
        h, mask = ops.embed_label(self.input_emb.weight, x_flat, z, self._selection(x_flat))
        h = self.emb_gn(h, p_drop=p, call_id=1)
        xs = []
        # JK buffer written in place by the fused layers (no torch.cat) when every layer takes that path
        widths = [c.trans_fns[0].weight.shape[0] for c in self.convs if isinstance(c, GLASSConv)]
        jk_buf = None
        if (self.jk and len(widths) == len(self.convs) and len(set(widths)) == 1 and widths[0] == h.shape[1] and
                code is not None and all("comb" in getattr(c, "_stack", {}) and len(c._stack["comb"]) == 6
                                         for c in self.convs) and ops.dual_linear_supported(h.shape[1]) and
                ops.USE_FUSED_DENSE):
            jk_buf = torch.empty((n, sum(widths)), dtype=torch.float32, device=h.device)
        for layer, conv in enumerate(self.convs):
            if jk_buf is not None:
                h = conv(h, edge_index, edge_weight, mask, out=jk_buf[:, layer * widths[0]:(layer + 1) * widths[0]])
            else:
                h = conv(h, edge_index, edge_weight, mask)
            xs.append(h)
            if layer + 1 == len(self.convs):
                break
            if self.gns is not None:
                if code is not None:
                    h = self.gns[layer](h, act=code, p_drop=p, call_id=conv.call_base + 1)
                    continue
                h = self.gns[layer](h)
            h = self.activation(h)
            h = F.dropout(h, p=self.dropout, training=self.training)
        if jk_buf is not None:
            h = ops.join_cols(jk_buf, xs)
        else:
            h = torch.cat(xs, dim=-1) if self.jk else xs[-1]
        if self.gns is not None:
            h = self.gns[-1](h)
        return h


# ---------------------------------------------------------------------------------------------
class PoolModule(nn.Module):
    """Subgraph readout.  `forward(x, batch)` keeps the reference's gathered-rows + batch-vector
    calling convention; GLASS.Pool uses the fused padded-matrix kernel directly."""
    mode = None

    def __init__(self, pool_fn=None, trans_fn=None):
        super().__init__()
        self.pool_fn = pool_fn
        self.trans_fn = trans_fn

    def forward(self, x, batch):
        if self.trans_fn is not None:
            x = self.trans_fn(x)
        return ops.segment_pool(x, batch2pad(batch), self.mode)


class AddPool(PoolModule):
    mode = "sum"

    def __init__(self, trans_fn=None):
        super().__init__(None, trans_fn)


class MaxPool(PoolModule):
    mode = "max"

    def __init__(self, trans_fn=None):
        super().__init__(None, trans_fn)


class MeanPool(PoolModule):
    mode = "mean"

    def __init__(self, trans_fn=None):
        super().__init__(None, trans_fn)


class SizePool(AddPool):
    mode = "size"


class GLASS(nn.Module):
    """EmbZGConv + per-target pooling and prediction heads (`preds[id]`, `pools[id]`)."""
    def __init__(self, conv: EmbZGConv, preds: nn.ModuleList, pools: nn.ModuleList):
        super().__init__()
        self.conv = conv
        self.preds = preds
        self.pools = pools

    def __deepcopy__(self, memo):
        """copy.deepcopy(model) (an early-stopping snapshot, a twin): a PLAIN copy — own parameter storage, no arena, no captured
        graphs, no adopted optimizer (arena.strip_runtime); it gets its own runtime objects the first time it trains."""
        from .arena import deepcopy_plain
        return deepcopy_plain(self, memo)

    def _channels(self, x):
        """The feature channels of x [N, C, F] as dense tensors, made once per feature tensor: x is static per dataset, and
        the conv caches its selection CSR (and captured graphs hold its pointers) by the tensor it is handed."""
        if x.shape[1] == 1:
            return [x.reshape(x.shape[0], x.shape[-1])]
        cache = self.__dict__.setdefault("_chan_cache", {})
        key = (x.data_ptr(), tuple(x.shape), x._version)
        hit = cache.get(key)
        if hit is None:
            if len(cache) >= 8:
                cache.clear()
            hit = cache[key] = (x, [x[:, c, :].contiguous() for c in range(x.shape[1])])
        return hit[1]

    def NodeEmb(self, x, edge_index, edge_weight, z=None):
        embs = [self.conv(xc, edge_index, edge_weight, z) for xc in self._channels(x)]
        if len(embs) == 1:
            return embs[0]  # mean over a single feature channel is the identity
        return torch.stack(embs, dim=1).mean(dim=1)

    def Pool(self, emb, subG_node, pool):
        if isinstance(pool, PoolModule) and pool.trans_fn is None and pool.mode is not None:
            return ops.segment_pool(emb, subG_node, pool.mode)
        from .utils import pad2batch
        batch, pos = pad2batch(subG_node)
        return pool(emb[pos], batch)

    def forward(self, x, edge_index, edge_weight, subG_node, z=None, id=0):
        emb = self.NodeEmb(x, edge_index, edge_weight, z)
        emb = self.Pool(emb, subG_node, self.pools[id])
        return _head(self.preds[id], emb)


def _head(pred, emb):
    """The prediction head.  A bare nn.Linear under no_grad (evaluation: train.test) runs as one small kernel of this
    library (glass_head_linear_f32) instead of a library GEMM; anything else is the module itself."""
    if (type(pred) is nn.Linear and not torch.is_grad_enabled() and emb.is_cuda and emb.dim() == 2 and
            emb.dtype == torch.float32 and pred.weight.dtype == torch.float32):
        from . import _lib
        emb = emb if emb.stride(1) == 1 else emb.contiguous()
        w = pred.weight if pred.weight.is_contiguous() else pred.weight.contiguous()
        out = torch.empty((emb.shape[0], w.shape[0]), dtype=torch.float32, device=emb.device)
        rc = _lib.load().glass_head_linear_f32(emb.data_ptr(), emb.stride(0), w.data_ptr(),
                                               0 if pred.bias is None else pred.bias.data_ptr(), emb.shape[0], emb.shape[1],
                                               w.shape[0], out.data_ptr(), out.stride(0), torch.cuda.current_stream().cuda_stream)
        _lib.check(rc, "glass_head_linear_f32")
        return out
    return pred(emb)


# ---------------------------------------------------------------------------------------------
# SSL pre-training models (link prediction) — reference impl/models.py:361-509.  Same kernels:
# K1 for the aggregation, K6 GraphNorm, K3 embedding gather (+ K1 backward), K7 for the pair readout.
# ---------------------------------------------------------------------------------------------
class MyGCNConv(nn.Module):
    """Unlabeled message-passing layer: comb_fn([GraphNorm(adj @ act(trans_fn(x_))) || x_])."""
    def __init__(self, in_channels: int, out_channels: int, activation=nn.ReLU(inplace=True), aggr="mean"):
        super().__init__()
        self.trans_fn = Linear(in_channels, out_channels)
        self.comb_fn = Linear(in_channels + out_channels, out_channels)
        self.adj = None
        self.activation = activation
        self.aggr = aggr
        self.gn = GraphNorm(out_channels)

    def reset_parameters(self):
        self.trans_fn.reset_parameters()
        self.comb_fn.reset_parameters()
        self.gn.reset_parameters()

    def forward(self, x_, edge_index, edge_weight):
        if self.adj is None:
            self.adj = buildAdj(edge_index, edge_weight, x_.shape[0], self.aggr)
        x = self.activation(self.trans_fn(x_))
        x = self.gn(ops.spmm(self.adj, x))
        return self.comb_fn(torch.cat((x, x_), dim=-1))


class EmbGConv(nn.Module):
    """Embedding + `num_layers` unlabeled conv layers (GraphNorm / activation / dropout between them)."""
    def __init__(self, input_channels: int, hidden_channels: int, output_channels: int, num_layers: int, max_deg: int,
                 dropout=0, activation=nn.ReLU(inplace=True), conv=MyGCNConv, gn=True, jk=False, **kwargs):
        super().__init__()
        self.input_emb = nn.Embedding(int(max_deg) + 1, hidden_channels)
        self.jk = jk
        dims = [input_channels] + [hidden_channels] * (num_layers - 1) + [output_channels]
        self.convs = nn.ModuleList([conv(in_channels=dims[i], out_channels=dims[i + 1], **kwargs)
                                    for i in range(num_layers)])
        for l, layer in enumerate(self.convs):
            layer.call_base = 16 * (l + 1)  # dropout streams of the step program (stack.StackProgram, unlabeled mode)
        self.activation = activation
        self.dropout = dropout
        self.gns = nn.ModuleList([GraphNorm(hidden_channels) for _ in range(num_layers - 1)]) if gn else None
        self.reset_parameters()

    def reset_parameters(self):
        for conv in self.convs:
            conv.reset_parameters()
        if self.gns is not None:
            for gn in self.gns:
                gn.reset_parameters()

    _selection = EmbZGConv._selection

    def forward(self, x, edge_index, edge_weight, z=None):
        x_flat = x.reshape(-1)
        if x_flat.dtype != torch.int64:
            x_flat = x_flat.to(torch.int64)
        if USE_STACK and z is None and stack.StackProgram.supported(self):
            # the whole stack as one autograd node on the fused kernels (glass_amd/stack.py, unlabeled mode)
            if not x_flat.is_contiguous():
                x_flat = x_flat.contiguous()
            return stack.run(self, x_flat, None, edge_index, edge_weight)
        h, _mask = ops.embed_label(self.input_emb.weight, x_flat, None, self._selection(x_flat))
        h = F.dropout(h, p=self.dropout, training=self.training)
        # The reference appends the GraphNorm output and then applies the activation to that very tensor; with
        # the driver's nn.ReLU(inplace=True) (GNNEmb.py:90) the appended tensor is therefore the ACTIVATED one.
        inplace = bool(getattr(self.activation, "inplace", False))
        xs = []
        for layer, conv in enumerate(self.convs[:-1]):
            h = conv(h, edge_index, edge_weight)
            if self.gns is not None:
                h = self.gns[layer](h)
            pre = h
            h = self.activation(h)
            xs.append(h if inplace else pre)
            h = F.dropout(h, p=self.dropout, training=self.training)
        xs.append(self.convs[-1](h, edge_index, edge_weight))
        return torch.cat(xs, dim=-1) if self.jk else xs[-1]


class EdgeGNN(nn.Module):
    """Link-prediction wrapper: node embeddings -> mean over each node pair -> preds[id]."""
    def __init__(self, conv, preds: nn.ModuleList, pools: nn.ModuleList):
        super().__init__()
        self.conv = conv
        self.preds = preds
        self.pools = pools

    NodeEmb, _channels, __deepcopy__ = GLASS.NodeEmb, GLASS._channels, GLASS.__deepcopy__

    def Pool(self, emb, subG_node, pool):
        return ops.segment_pool(emb, subG_node, "mean")  # emb[subG_node].mean(dim=1); pairs carry no padding

    def forward(self, x, edge_index, edge_weight, subG_node, z=None, id=0):
        emb = self.NodeEmb(x, edge_index, edge_weight, z)
        emb = self.Pool(emb, subG_node, self.pools[id])
        return self.preds[id](emb)
