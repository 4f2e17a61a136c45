"""Test-only stand-in for the PyTorch-Geometric symbols the reference imports.

PURPOSE: the reference (`/root/reference/impl/models.py:4-6`, `impl/SubGDataset.py:1`,
`datasets.py:5-6`) imports `torch_geometric`, which is absent from this image and cannot be
installed.  `install()` registers minimal modules in `sys.modules` so that the reference's own
files import and run UNMODIFIED on CPU; `tests/golden/make_golden.py` then dumps golden vectors.

This file is NOT part of the product and is never needed on the GPU box.  It restates the
published semantics of PyTorch-Geometric 1.7.2 (the version the reference pins in prose,
`README.md:19`) + torch_scatter for exactly the symbols used:

  torch_geometric.nn.norm.GraphNorm / GraphSizeNorm
  torch_geometric.nn.glob.glob.global_{add,mean,max}_pool
  torch_geometric.nn.GCNConv                (imported, never called on the GLASS path)
  torch_geometric.data.Data
  torch_geometric.utils.{is_undirected,to_undirected,negative_sampling,to_networkx}

They are written in the scatter-with-batch-vector form PyG itself uses (so the oracle, which
uses closed-form whole-graph reductions, is an independent formulation).
"""
import sys
import types

import torch
import torch.nn as nn


# ---- torch_scatter semantics -------------------------------------------------------------
def _scatter_add(src, index, dim_size):
    out = torch.zeros((dim_size, ) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
    return out.index_add_(0, index, src)


def _scatter_mean(src, index, dim_size):
    # torch_scatter.scatter_mean: sum / clamp(count, min=1)
    out = _scatter_add(src, index, dim_size)
    cnt = _scatter_add(torch.ones(index.shape[0], dtype=src.dtype, device=src.device), index,
                       dim_size).clamp_(min=1)
    return out / cnt.reshape((-1, ) + (1, ) * (src.dim() - 1))


def _scatter_max(src, index, dim_size):
    # torch_scatter.scatter_max: empty segments -> 0
    out = torch.full((dim_size, ) + tuple(src.shape[1:]), float("-inf"), dtype=src.dtype,
                     device=src.device)
    idx = index.reshape((-1, ) + (1, ) * (src.dim() - 1)).expand_as(src)
    out = out.scatter_reduce(0, idx, src, reduce="amax", include_self=True)
    return torch.where(torch.isinf(out) & (out < 0), torch.zeros_like(out), out)


def _degree(index, num_nodes, dtype):
    out = torch.zeros((num_nodes, ), dtype=dtype, device=index.device)
    return out.scatter_add_(0, index, torch.ones(index.shape[0], dtype=dtype, device=index.device))


# ---- torch_geometric.nn.norm -------------------------------------------------------------
class GraphNorm(nn.Module):
    """PyG 1.7.2 `GraphNorm(in_channels, eps=1e-5)`."""
    def __init__(self, in_channels, eps=1e-5):
        super().__init__()
        self.in_channels = in_channels
        self.eps = eps
        self.weight = nn.Parameter(torch.empty(in_channels))
        self.bias = nn.Parameter(torch.empty(in_channels))
        self.mean_scale = nn.Parameter(torch.empty(in_channels))
        self.reset_parameters()

    def reset_parameters(self):
        nn.init.ones_(self.weight)
        nn.init.zeros_(self.bias)
        nn.init.ones_(self.mean_scale)

    def forward(self, x, batch=None):
        if batch is None:
            batch = x.new_zeros(x.size(0), dtype=torch.long)
        batch_size = int(batch.max()) + 1
        mean = _scatter_mean(x, batch, batch_size)[batch]
        out = x - mean * self.mean_scale
        var = _scatter_mean(out.pow(2), batch, batch_size)
        std = (var + self.eps).sqrt()[batch]
        return self.weight * out / std + self.bias


class GraphSizeNorm(nn.Module):
    """PyG 1.7.2 `GraphSizeNorm`: x * degree(batch)^-0.5 [batch]."""
    def forward(self, x, batch=None):
        if batch is None:
            batch = torch.zeros(x.size(0), dtype=torch.long, device=x.device)
        inv_sqrt_deg = _degree(batch, int(batch.max()) + 1, x.dtype).pow(-0.5)
        return x * inv_sqrt_deg[batch].view(-1, 1)


# ---- torch_geometric.nn.glob.glob --------------------------------------------------------
def global_add_pool(x, batch, size=None):
    size = int(batch.max().item() + 1) if size is None else size
    return _scatter_add(x, batch, size)


def global_mean_pool(x, batch, size=None):
    size = int(batch.max().item() + 1) if size is None else size
    return _scatter_mean(x, batch, size)


def global_max_pool(x, batch, size=None):
    size = int(batch.max().item() + 1) if size is None else size
    return _scatter_max(x, batch, size)


class GCNConv(nn.Module):
    """Imported by the reference (`impl/models.py:4`) as a default argument only."""
    def __init__(self, *a, **k):
        super().__init__()

    def forward(self, *a, **k):
        raise NotImplementedError("GCNConv is off the GLASS path; stub only")


# ---- torch_geometric.data ----------------------------------------------------------------
class Data:
    def __init__(self, x=None, edge_index=None, edge_attr=None, y=None, pos=None, **kwargs):
        self.x = x
        self.edge_index = edge_index
        self.edge_attr = edge_attr
        self.y = y
        self.pos = pos
        for k, v in kwargs.items():
            setattr(self, k, v)


# ---- torch_geometric.utils ---------------------------------------------------------------
def _coalesce(edge_index, edge_attr, n):
    key = edge_index[0] * n + edge_index[1]
    uniq, inv = torch.unique(key, sorted=True, return_inverse=True)
    ei = torch.stack((uniq // n, uniq % n))
    if edge_attr is None:
        return ei, None
    ea = torch.zeros((uniq.shape[0], ) + tuple(edge_attr.shape[1:]), dtype=edge_attr.dtype)
    ea.index_add_(0, inv, edge_attr)  # reduce="add"
    return ei, ea


def to_undirected(edge_index, edge_attr=None, num_nodes=None, reduce="add"):
    n = int(edge_index.max()) + 1 if num_nodes is None else num_nodes
    row, col = edge_index
    ei = torch.stack((torch.cat((row, col)), torch.cat((col, row))))
    ea = None if edge_attr is None else torch.cat((edge_attr, edge_attr))
    ei, ea = _coalesce(ei, ea, n)
    return ei if edge_attr is None else (ei, ea)


def is_undirected(edge_index, edge_attr=None, num_nodes=None):
    n = int(edge_index.max()) + 1 if num_nodes is None else num_nodes
    ei, _ = _coalesce(edge_index, None, n)
    und = to_undirected(ei, num_nodes=n)
    return ei.size(1) == und.size(1)


def negative_sampling(edge_index, num_nodes=None, num_neg_samples=None):
    raise NotImplementedError("off the GLASS path; stub only")


def to_networkx(*a, **k):
    raise NotImplementedError("off the GLASS path; stub only")


def install():
    """Register the stub modules; idempotent."""
    if "torch_geometric" in sys.modules and getattr(sys.modules["torch_geometric"],
                                                    "__glass_stub__", False):
        return

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    norm = mod("torch_geometric.nn.norm", GraphNorm=GraphNorm, GraphSizeNorm=GraphSizeNorm)
    globglob = mod("torch_geometric.nn.glob.glob", global_add_pool=global_add_pool,
                   global_mean_pool=global_mean_pool, global_max_pool=global_max_pool)
    glob = mod("torch_geometric.nn.glob", glob=globglob, global_add_pool=global_add_pool,
               global_mean_pool=global_mean_pool, global_max_pool=global_max_pool)
    nnm = mod("torch_geometric.nn", GCNConv=GCNConv, norm=norm, glob=glob)
    data = mod("torch_geometric.data", Data=Data)
    utils = mod("torch_geometric.utils", is_undirected=is_undirected, to_undirected=to_undirected,
                negative_sampling=negative_sampling, to_networkx=to_networkx)
    mod("torch_geometric", nn=nnm, data=data, utils=utils, __glass_stub__=True)
