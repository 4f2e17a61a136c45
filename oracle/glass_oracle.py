"""CPU oracle for the GLASS labeled message-passing hot path.   *** TEST INFRASTRUCTURE ***

This module is the checker, never the product: only `tests/`, `__graft_entry__.smoke()` and
`bench.py`'s `cpu_baseline` leg may import it.  Nothing under `glass_amd/` or `impl/` does.

What it is: a PyG-free restatement, in plain CPU torch ops, of the reference algorithm
(`/root/reference/impl/models.py:83-355`, `/root/reference/impl/utils.py:5-45`).  It keeps
the reference's operator choices where they fix the arithmetic (uncoalesced COO
`torch.sparse.mm` for the aggregation, `nn.Linear`, `nn.Embedding`, `index_add_` pooling) so
that it doubles as the timed "reference CPU path" (`cpu_baseline.kind = "port"`).  It is
dtype-generic: `.double()` gives the fp64 ground truth used for the noise-floor comparisons
(SURVEY.md Appendix B).

How it is pinned: `tests/golden/make_golden.py` imports the reference itself (unmodified,
from `/root/reference`, with `tests/golden/pyg_stub.py` standing in for the five absent
PyTorch-Geometric symbols) and writes `tests/golden/*.npz`; `tests/test_oracle_golden.py`
checks this oracle against every one of those fixtures plus the three docstring examples that
are the reference's only executable specifications (`impl/utils.py:9,21`,
`impl/models.py:288-289`).  The reference's OWN code is therefore pinned by reference-run
vectors.  The arithmetic it borrows from PyTorch-Geometric 1.7.2 / torch_scatter (GraphNorm,
GraphSizeNorm, global_{add,mean,max}_pool — third-party, not under /root/reference, and the
reference holds no test pinning them) is restated from their published semantics:
**parity unpinned for that third-party arithmetic** (SURVEY.md §8c).

State-dict keys equal the reference's so weights interchange:
  conv.input_emb.weight, conv.emb_gn.{weight,bias,mean_scale},
  conv.convs.{l}.trans_fns.{0,1}.{weight,bias}, conv.convs.{l}.comb_fns.{0,1}.{weight,bias},
  conv.convs.{l}.gn.{...}, conv.gns.{l}.{...}, preds.0.{weight,bias}
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


# ----------------------------------------------------------------------------------------
# label / index helpers                                     (reference impl/utils.py:5-45)
# ----------------------------------------------------------------------------------------
def max_zero_one(x, pos):
    """`MaxZOZ` (impl/utils.py:32-45): z[n] = 1 iff n occurs in `pos` (padding -1 ignored)."""
    z = torch.zeros(x.shape[0], dtype=torch.int64)
    flat = pos.reshape(-1)
    z[flat[flat >= 0]] = 1
    return z


def pad_to_batch(pad):
    """`pad2batch` (impl/utils.py:18-29): row-major flatten of the padded matrix, -1 dropped.
    Returns (batch, pos): batch[k] = subgraph row of the k-th kept entry, pos[k] = node id."""
    rows = torch.arange(pad.shape[0]).reshape(-1, 1).expand(-1, pad.shape[1]).reshape(-1)
    flat = pad.reshape(-1)
    keep = flat >= 0
    return rows[keep], flat[keep]


def batch_to_pad(batch):
    """`batch2pad` (impl/utils.py:5-15): inverse layout; row i lists the positions j with
    batch[j] == i-th distinct non-negative value, padded with -1."""
    vals = [v for v in torch.unique(batch).tolist() if v >= 0]
    lists = [torch.nonzero(batch == v).reshape(-1) for v in vals]
    width = max(len(l) for l in lists)
    out = torch.full((len(lists), width), -1, dtype=torch.int64)
    for i, l in enumerate(lists):
        out[i, :len(l)] = l
    return out


# ----------------------------------------------------------------------------------------
# normalised adjacency                                   (reference impl/models.py:83-111)
# ----------------------------------------------------------------------------------------
def adjacency_values(edge_index, edge_weight, n_node, aggr):
    """Per-edge values of the normalised adjacency.  row = edge_index[0] is the DESTINATION
    (output row), col = edge_index[1] the source (models.py:90-92,164).  deg = weighted row
    sum, isolated rows (deg < 0.5) get +1 (models.py:93-94).  No self-loops are added."""
    row, col = edge_index[0], edge_index[1]
    deg = torch.zeros(n_node, dtype=edge_weight.dtype).index_add_(0, row, edge_weight)
    deg = torch.where(deg < 0.5, deg + 1.0, deg)
    if aggr == "mean":
        return (1.0 / deg)[row] * edge_weight
    if aggr == "sum":
        return edge_weight
    if aggr == "gcn":
        dinv = torch.pow(deg, -0.5)
        return dinv[row] * edge_weight * dinv[col]
    raise NotImplementedError(aggr)


def build_adj(edge_index, edge_weight, n_node, aggr):
    """`buildAdj`: UNCOALESCED COO [N,N]; duplicates act additively inside the matmul."""
    return torch.sparse_coo_tensor(edge_index, adjacency_values(edge_index, edge_weight, n_node, aggr),
                                   size=(n_node, n_node))


def dense_adj(edge_index, edge_weight, n_node, aggr):
    """Dense [N,N] of the same operator (tiny graphs only; used by fixtures/tests)."""
    vals = adjacency_values(edge_index, edge_weight, n_node, aggr)
    a = torch.zeros(n_node * n_node, dtype=vals.dtype)
    a.index_add_(0, edge_index[0] * n_node + edge_index[1], vals)
    return a.reshape(n_node, n_node)


# ----------------------------------------------------------------------------------------
# PyG 1.7.2 arithmetic used with batch=None                       (third-party; see header)
# ----------------------------------------------------------------------------------------
class GraphNorm(nn.Module):
    """Whole-graph GraphNorm (call sites models.py:165,249,257,266,271; all with batch=None):
    mu = mean_rows(x); o = x - mu*mean_scale; y = weight*o/sqrt(mean_rows(o^2)+eps) + bias."""
    def __init__(self, channels, eps=1e-5):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(channels))
        self.bias = nn.Parameter(torch.zeros(channels))
        self.mean_scale = nn.Parameter(torch.ones(channels))

    def forward(self, x):
        centred = x - x.mean(dim=0, keepdim=True) * self.mean_scale
        var = centred.pow(2).mean(dim=0, keepdim=True)
        return self.weight * centred / (var + self.eps).sqrt() + self.bias


def segment_pool(emb, batch, n_seg, mode):
    """global_{add,mean,max}_pool / SizePool (models.py:275-319) on rows already gathered.
    mode: sum | mean (sum / max(count,1)) | max (empty -> 0) | size (x * n_b^-1/2, then sum)."""
    cnt = torch.zeros(n_seg, dtype=emb.dtype).index_add_(0, batch, torch.ones(batch.shape[0], dtype=emb.dtype))
    if mode == "size":
        emb = emb * cnt.pow(-0.5)[batch].reshape(-1, 1)
    if mode in ("sum", "size", "mean"):
        out = torch.zeros(n_seg, emb.shape[1], dtype=emb.dtype).index_add_(0, batch, emb)
        if mode == "mean":
            out = out / cnt.clamp(min=1).reshape(-1, 1)
        return out
    if mode == "max":
        out = torch.full((n_seg, emb.shape[1]), float("-inf"), dtype=emb.dtype)
        out = out.scatter_reduce(0, batch.reshape(-1, 1).expand_as(emb), emb, reduce="amax")
        return torch.where(torch.isneginf(out), torch.zeros_like(out), out)
    raise NotImplementedError(mode)


# ----------------------------------------------------------------------------------------
# model                                                   (reference impl/models.py:114-355)
# ----------------------------------------------------------------------------------------
# Dropout (models.py:166, 251, 259: F.dropout with the YAML's p).  A test may FEED the masks: `mask_feed(list)` makes the
# next dropouts, in call order, multiply by the given keep-scale tensors (0 or 1/(1-p)) instead of drawing their own — how
# a dropout-on run of the HIP path is compared with this restatement on the very same masks.
_MASK_FEED = []


def mask_feed(scales):
    _MASK_FEED[:] = list(scales)


def _dropout(h, p, training):
    if training and p > 0 and _MASK_FEED:
        return h * _MASK_FEED.pop(0).to(h.dtype)
    return F.dropout(h, p=p, training=training)


# ReLU derivative masks may be FED the same way (`relu_mask_feed`): the next ReLUs, in call order, compute h * mask with the
# given 0/1 tensors instead of max(h, 0).  ReLU is not differentiable at 0, and an fp32 evaluation decides relu'(h) from
# the SIGN OF ITS OWN ROUNDING ERROR wherever |h| is below its noise (a handful of the 8.4 M head pre-activations of the
# pre-training step: each flip moves the flat gradient by ~1.4e-5 of its largest entry) — a gradient comparison against an
# fp64 evaluation is only meaningful on the same branch of every ReLU, exactly as a dropout run is only comparable on the
# same masks.  The forward value changes by at most the fed evaluation's rounding error at the flipped elements.
_RELU_FEED = []


def relu_mask_feed(masks):
    _RELU_FEED[:] = list(masks)


def _relu(h):
    if _RELU_FEED:
        return h * _RELU_FEED.pop(0).to(h.dtype)
    return F.relu(h)


def _mix(mask, zr, f1, f0):
    """models.py:161-162 / 172-173: labeled rows zr*f1+(1-zr)*f0, unlabeled zr*f0+(1-zr)*f1."""
    return torch.where(mask, zr * f1 + (1 - zr) * f0, zr * f0 + (1 - zr) * f1)


class OracleConv(nn.Module):
    """`GLASSConv` (models.py:114-174)."""
    def __init__(self, cin, cout, aggr, z_ratio, dropout):
        super().__init__()
        self.trans_fns = nn.ModuleList([nn.Linear(cin, cout), nn.Linear(cin, cout)])
        self.comb_fns = nn.ModuleList([nn.Linear(cin + cout, cout), nn.Linear(cin + cout, cout)])
        self.gn = GraphNorm(cout)
        self.aggr, self.z_ratio, self.dropout = aggr, z_ratio, dropout
        self.adj = None  # cached after first forward, like the reference (models.py:154-156)

    def forward(self, x_in, edge_index, edge_weight, mask, act):
        if self.adj is None:
            self.adj = build_adj(edge_index, edge_weight.to(x_in.dtype), x_in.shape[0], self.aggr)
        f1 = act(self.trans_fns[1](x_in))
        f0 = act(self.trans_fns[0](x_in))
        h = self.adj @ _mix(mask, self.z_ratio, f1, f0)
        h = _dropout(self.gn(h), self.dropout, self.training)
        h = torch.cat((h, x_in), dim=-1)
        return _mix(mask, self.z_ratio, self.comb_fns[1](h), self.comb_fns[0](h))


class OracleEmbZGConv(nn.Module):
    """`EmbZGConv` (models.py:177-272).  gn=False (models.py:194, 226-227): no `gns` at all — emb_gn and every conv's own
    GraphNorm stay.  act: "elu" = the driver's nn.ELU(inplace=True) (GLASSTest.py:143), "relu" = the constructor default
    nn.ReLU() (models.py:192; NOT in place, so the tensors JK keeps stay raw also without gns)."""
    def __init__(self, hidden, out, n_layers, max_deg, dropout, aggr, z_ratio, jk=True, gn=True, act="elu"):
        super().__init__()
        self.input_emb = nn.Embedding(int(max_deg) + 1, hidden)
        self.emb_gn = GraphNorm(hidden)
        dims = [hidden] * (n_layers - 1) + [out]
        self.convs = nn.ModuleList([OracleConv(hidden, d, aggr, z_ratio, dropout) for d in dims])
        self.gns = nn.ModuleList([GraphNorm(hidden) for _ in range(n_layers - 1)] +
                                 [GraphNorm(out + (n_layers - 1) * hidden if jk else out)]) if gn else None
        self.jk, self.dropout = jk, dropout
        self.act = {"elu": F.elu, "relu": F.relu}[act]

    def forward(self, x, edge_index, edge_weight, z=None):
        n = x.shape[0]
        mask = torch.ones(n, 1, dtype=torch.bool) if z is None else (z > 0.5).reshape(-1, 1)
        act = self.act
        h = self.emb_gn(self.input_emb(x).reshape(n, -1))
        h = _dropout(h, self.dropout, self.training)
        saved = []
        for l, conv in enumerate(self.convs):
            h = conv(h, edge_index, edge_weight, mask, act)
            saved.append(h)  # JK keeps the RAW conv outputs (models.py:254-255,260-264)
            if l + 1 < len(self.convs):
                h = _dropout(act(self.gns[l](h) if self.gns is not None else h), self.dropout, self.training)
        out = torch.cat(saved, dim=-1) if self.jk else saved[-1]
        return self.gns[-1](out) if self.gns is not None else out


class OracleGLASS(nn.Module):
    """`GLASS` (models.py:322-355) with one head (`preds.0`) and one pool."""
    def __init__(self, hidden, n_layers, max_deg, out_channels, aggr="mean", pool="sum", z_ratio=0.8,
                 dropout=0.0, jk=True, gn=True, act="elu"):
        super().__init__()
        self.conv = OracleEmbZGConv(hidden, hidden, n_layers, max_deg, dropout, aggr, z_ratio, jk, gn=gn, act=act)
        self.preds = nn.ModuleList([nn.Linear(hidden * n_layers if jk else hidden, out_channels)])
        self.pool = pool

    def node_emb(self, x, edge_index, edge_weight, z=None):
        # x is [N,1,1] int64 (datasets.py:52,56,60); one feature channel -> the mean over channels
        # (models.py:336-344) is the identity.
        assert x.shape[1] == 1
        return self.conv(x[:, 0, :].reshape(x.shape[0], -1), edge_index, edge_weight, z)

    def pool_emb(self, emb, subg_node):
        batch, pos = pad_to_batch(subg_node)
        return segment_pool(emb[pos], batch, int(batch.max()) + 1, self.pool)

    def forward(self, x, edge_index, edge_weight, subg_node, z=None, id=0):
        emb = self.node_emb(x, edge_index, edge_weight, z)
        return self.preds[id](self.pool_emb(emb, subg_node))


def train_step(model, optimizer, loss_fn, x, edge_index, edge_weight, pos, y):
    """One hot-loop iteration as in impl/train.py:10-16 + ZGDataloader (SubGDataset.py:92-96)."""
    z = max_zero_one(x, pos)
    optimizer.zero_grad()
    loss = loss_fn(model(x, edge_index, edge_weight, pos, z), y)
    loss.backward()
    val = loss.detach().item()
    optimizer.step()
    return val


# ----------------------------------------------------------------------------------------
# SSL pre-training path (link prediction)              (reference impl/models.py:361-509)
# ----------------------------------------------------------------------------------------
class OracleGCNConv(nn.Module):
    """`MyGCNConv` (models.py:361-397): one weight set, no labels:
    comb_fn([GraphNorm(adj @ act(trans_fn(x_))) || x_])."""
    def __init__(self, cin, cout, aggr):
        super().__init__()
        self.trans_fn = nn.Linear(cin, cout)
        self.comb_fn = nn.Linear(cin + cout, cout)
        self.gn = GraphNorm(cout)
        self.aggr = aggr
        self.adj = None

    def forward(self, x_in, edge_index, edge_weight, act):
        if self.adj is None:
            self.adj = build_adj(edge_index, edge_weight.to(x_in.dtype), x_in.shape[0], self.aggr)
        h = self.gn(self.adj @ act(self.trans_fn(x_in)))
        return self.comb_fn(torch.cat((h, x_in), dim=-1))


class OracleEmbGConv(nn.Module):
    """`EmbGConv` (models.py:400-473): embedding (+dropout), then per non-last layer conv -> GraphNorm -> save ->
    act -> dropout; the last conv output is saved raw; JK concatenates the saved tensors.
    Quirk kept on purpose: the driver's activation is nn.ReLU(inplace=True) (GNNEmb.py:90) and the reference
    applies it to the very tensor it has just appended (models.py:461-463), so the tensors JK sees are the
    ACTIVATED ones.  (GNNEmb.py always runs jk=False, where only the raw last output matters.)"""
    def __init__(self, cin, hidden, out, n_layers, max_deg, dropout, aggr, jk=False):
        super().__init__()
        self.input_emb = nn.Embedding(int(max_deg) + 1, hidden)
        dims = [cin] + [hidden] * (n_layers - 1) + [out]
        self.convs = nn.ModuleList([OracleGCNConv(dims[i], dims[i + 1], aggr) for i in range(n_layers)])
        self.gns = nn.ModuleList([GraphNorm(hidden) for _ in range(n_layers - 1)])
        self.jk, self.dropout = jk, dropout

    def forward(self, x, edge_index, edge_weight, z=None):
        act = _relu  # GNNEmb.py:90 nn.ReLU(inplace=True)  (F.relu unless a test feeds derivative masks)
        h = F.dropout(self.input_emb(x.reshape(-1)), p=self.dropout, training=self.training)
        saved = []
        for l, conv in enumerate(self.convs[:-1]):
            h = act(self.gns[l](conv(h, edge_index, edge_weight, act)))  # in-place ReLU aliases the saved tensor
            saved.append(h)
            h = F.dropout(h, p=self.dropout, training=self.training)
        saved.append(self.convs[-1](h, edge_index, edge_weight, act))
        return torch.cat(saved, dim=-1) if self.jk else saved[-1]


class OracleEdgeGNN(nn.Module):
    """`EdgeGNN` (models.py:476-509) with the 2-layer MLP head of GNNEmb.py:94-99 (Linear, [dropout], ReLU,
    Linear); Pool = mean over the node pair (models.py:501-504)."""
    def __init__(self, hidden, n_layers, max_deg, aggr="mean", dropout=0.0, jk=False):
        super().__init__()
        self.conv = OracleEmbGConv(hidden, hidden, hidden, n_layers, max_deg, dropout, aggr, jk)
        width = hidden * n_layers if jk else hidden

        class _Seq(nn.Module):  # key layout preds.0.seq.modlist.{i} of the reference's MLP/Seq
            def __init__(self, mods):
                super().__init__()
                self.modlist = nn.ModuleList(mods)

        class _ReLU(nn.Module):
            def forward(self, h):
                return _relu(h)

        class _MLP(nn.Module):
            def __init__(self):
                super().__init__()
                mods = [nn.Linear(width, hidden)] + ([nn.Dropout(dropout)] if dropout > 0 else []) + \
                       [_ReLU(), nn.Linear(hidden, 1)]
                self.seq = _Seq(mods)

            def forward(self, x):
                for m in self.seq.modlist:
                    x = m(x)
                return x

        self.preds = nn.ModuleList([_MLP()])

    def forward(self, x, edge_index, edge_weight, pairs, z=None, id=0):
        assert x.shape[1] == 1
        emb = self.conv(x[:, 0, :].reshape(x.shape[0], -1), edge_index, edge_weight, z)
        return self.preds[id](emb[pairs].mean(dim=1))
