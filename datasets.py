"""Dataset container and loaders with the reference's top-level `datasets` interface
(/root/reference/datasets.py): `BaseGraph`, `load_dataset(name)`.  PyG-free and pickle-free.

  * the four shipped synthetic sets (density, cut_ratio, coreness, component) are read from
    `dataset_/<name>/graph.npz` (plain arrays converted once by tools/convert_datasets.py);
  * the real-world sets (ppi_bp, hpo_metab, hpo_neuro, em_user) are parsed from the SubGNN text
    format under `./dataset/<name>/` (`edge_list.txt`, `subgraphs.pth`) when the user has them
    (they ship with neither repo: reference README.md:26);
  * `synthetic:<workload>` builds a seeded graph shaped like a benchmark configuration
    (glass_amd/synth.py), e.g. `synthetic:ppi_bp`.

One-time CPU I/O; nothing here is on the accelerated path.
"""
import os

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))


def _coalesce_add(edge_index, edge_attr, n):
    """Sort by (row, col) and merge duplicates by summing their weights (PyG coalesce, reduce='add')."""
    key = edge_index[0] * n + edge_index[1]
    uniq, inv = torch.unique(key, sorted=True, return_inverse=True)
    ei = torch.stack((torch.div(uniq, n, rounding_mode="floor"), uniq % n))
    ea = torch.zeros(uniq.shape[0], dtype=edge_attr.dtype).index_add_(0, inv, edge_attr)
    return ei, ea


class BaseGraph:
    """x: node features ([N,1,0] until a set*Feature call); edge_index/edge_attr: symmetrised graph;
    pos: padded subgraph node matrix (-1 pad); y: subgraph targets; mask: 0/1/2 = train/valid/test."""
    def __init__(self, x, edge_index, edge_weight, subG_node, subG_label, mask):
        self.x, self.edge_index, self.edge_attr = x, edge_index, edge_weight
        self.pos, self.y, self.mask = subG_node, subG_label, mask
        self.to_undirected()

    @property
    def num_nodes(self):
        return self.x.shape[0]

    def _degree(self):
        n = self.x.shape[0]
        return torch.zeros(n, dtype=self.edge_attr.dtype).index_add_(0, self.edge_index[0].cpu(),
                                                                     self.edge_attr.cpu()).to(torch.int64)

    def setDegreeFeature(self, mod=1):
        """x[n] = rank of floor(deg(n)/mod) among the distinct values (reference datasets.py:45-52)."""
        deg = torch.div(self._degree(), mod, rounding_mode="floor")
        self.x = torch.unique(deg, return_inverse=True)[1].reshape(self.x.shape[0], 1, -1).to(self.edge_index.device)

    def setOneFeature(self):
        self.x = torch.ones((self.x.shape[0], 1, 1), dtype=torch.int64, device=self.edge_index.device)

    def setNodeIdFeature(self):
        self.x = torch.arange(self.x.shape[0], dtype=torch.int64,
                              device=self.edge_index.device).reshape(self.x.shape[0], 1, -1)

    def get_split(self, split: str):
        sel = self.mask == {"train": 0, "valid": 1, "test": 2}[split]
        return self.x, self.edge_index, self.edge_attr, self.pos[sel], self.y[sel]

    def to_undirected(self):
        """Symmetrise + coalesce (weights of duplicates add) unless the graph already is symmetric."""
        n = self.x.shape[0]
        ei, ea = self.edge_index, self.edge_attr
        fwd = torch.unique(ei[0] * n + ei[1])
        if fwd.shape[0] == ei.shape[1] and torch.equal(fwd, torch.unique(ei[1] * n + ei[0])):
            return  # already symmetric and duplicate-free
        both = torch.cat((ei, ei.flip(0)), dim=1)
        self.edge_index, self.edge_attr = _coalesce_add(both, torch.cat((ea, ea)), n)

    def get_LPdataset(self, use_loop=False):
        """Link-prediction dataset for SSL pre-training (reference datasets.py:73-91): every edge is a positive
        pair, an equal number of uniformly sampled non-edges are negatives; with use_loop one extra sample per node
        asks whether it has a self-loop.  -> (x, edge_index, edge_attr, pos [M,2], y [M])."""
        n = self.x.shape[0]
        ei = self.edge_index
        neg = negative_sampling(ei, n)
        pos = torch.cat((ei, neg), dim=1).t()
        y = torch.cat((torch.ones(ei.shape[1]), torch.zeros(neg.shape[1]))).to(ei.device)
        if use_loop:
            loops = ei[0][ei[0] == ei[1]]
            all_loops = torch.arange(n, device=ei.device).reshape(-1, 1)[:, [0, 0]]
            y_loop = torch.zeros(n, device=y.device)
            y_loop[loops] = 1
            pos = torch.cat((pos, all_loops), dim=0)
            y = torch.cat((y, y_loop), dim=0)
        return self.x, ei, self.edge_attr, pos, y

    def to(self, device):
        for name in ("x", "edge_index", "edge_attr", "pos", "y", "mask"):
            setattr(self, name, getattr(self, name).to(device))
        return self


def negative_sampling(edge_index, num_nodes, num_neg_samples=None):
    """Uniformly random node pairs that are NOT edges (PyG `negative_sampling`, sparse method): about as many
    as there are edges; may return slightly fewer on very dense graphs."""
    want = edge_index.shape[1] if num_neg_samples is None else num_neg_samples
    dev = edge_index.device
    existing = torch.unique(edge_index[0] * num_nodes + edge_index[1])
    out = torch.empty(0, dtype=torch.int64, device=dev)
    for _ in range(4):
        cand = torch.randint(0, num_nodes * num_nodes, (int(1.2 * (want - out.shape[0])) + 16, ), device=dev)
        hit = torch.searchsorted(existing, cand).clamp_(max=existing.shape[0] - 1)
        cand = cand[existing[hit] != cand]
        out = torch.unique(torch.cat((out, cand)))
        if out.shape[0] >= want:
            break
    out = out[torch.randperm(out.shape[0], device=dev)[:want]]
    return torch.stack((torch.div(out, num_nodes, rounding_mode="floor"), out % num_nodes))


def _split_mask(cnt):
    """50 % train / 25 % valid / 25 % test, shuffled with torch's global RNG (reference datasets.py:118-123)."""
    mask = torch.cat((torch.zeros(cnt - cnt // 2, dtype=torch.int64), torch.ones(cnt // 4, dtype=torch.int64),
                      2 * torch.ones(cnt // 2 - cnt // 4, dtype=torch.int64)))
    return mask[torch.randperm(mask.shape[0])]


def _load_shipped(name):
    path = os.path.join(_HERE, "dataset_", name, "graph.npz")
    if not os.path.exists(path):
        path = os.path.join("dataset_", name, "graph.npz")
    z = np.load(path)
    edge = torch.from_numpy(z["edges"].astype(np.int64)).t().contiguous()
    pos = torch.from_numpy(z["subG"].astype(np.int64))
    label = torch.from_numpy(z["label"])
    return BaseGraph(torch.empty((int(z["n_node"]), 1, 0)), edge, torch.ones(edge.shape[1]), pos, label,
                     _split_mask(pos.shape[0]))


def _load_subgnn_text(name):
    """SubGNN format: `subgraphs.pth` lines "n1-n2-...\\tlabel[-label...]\\ttrain|val|test";
    `edge_list.txt` one "u v" pair per line (reference datasets.py:131-227)."""
    root = os.path.join("dataset", name)
    sub_f, edge_f = os.path.join(root, "subgraphs.pth"), os.path.join(root, "edge_list.txt")
    if not (os.path.exists(sub_f) and os.path.exists(edge_f)):
        raise FileNotFoundError(f"{root}/subgraphs.pth and edge_list.txt not found: the real-world datasets are not "
                                "shipped (see README); use a shipped synthetic set or synthetic:<workload>")
    label_ids, rows = {}, {"train": [], "val": [], "test": []}
    with open(sub_f) as f:
        for line in f:
            parts = line.rstrip("\n").split("\t")
            nodes = [int(t) for t in parts[0].split("-") if t != ""]
            if not nodes:
                continue
            labs = parts[1].split("-")
            for lab in labs:
                label_ids.setdefault(lab, len(label_ids))
            rows[parts[2].strip()].append((nodes, [label_ids[lab] for lab in labs]))
    if len(rows["val"]) < len(rows["test"]):  # the reference swaps so that valid is the larger one
        rows["val"], rows["test"] = rows["test"], rows["val"]
    order = rows["train"] + rows["val"] + rows["test"]
    multilabel = any(len(labs) > 1 for _, labs in order)
    width = max(len(nodes) for nodes, _ in order)
    pos = torch.full((len(order), width), -1, dtype=torch.int64)
    for i, (nodes, _) in enumerate(order):
        pos[i, :len(nodes)] = torch.tensor(nodes)
    if multilabel:
        label = torch.zeros(len(order), len(label_ids))
        for i, (_, labs) in enumerate(order):
            label[i, labs] = 1
    else:
        label = torch.tensor([labs[0] for _, labs in order], dtype=torch.float)
    mask = torch.cat((torch.zeros(len(rows["train"]), dtype=torch.int64), torch.ones(len(rows["val"]), dtype=torch.int64),
                      2 * torch.ones(len(rows["test"]), dtype=torch.int64)))
    edges = np.loadtxt(edge_f, dtype=np.int64).reshape(-1, 2)
    # the reference reads the file through networkx.read_edgelist into an undirected simple Graph: a repeated line and
    # a reversed duplicate ("u v" and "v u") are ONE edge (they must not coalesce to weight 2 in to_undirected)
    lo, hi = edges.min(1), edges.max(1)
    _, first = np.unique(lo * (int(hi.max()) + 1 if edges.size else 1) + hi, return_index=True)
    edges = edges[np.sort(first)]
    edge_index = torch.from_numpy(edges).t().contiguous()
    n = int(max(pos.max(), edge_index.max())) + 1
    return BaseGraph(torch.empty((n, 1, 0)), edge_index, torch.ones(edge_index.shape[1]), pos, label.to(torch.float),
                     mask)


def _load_synthetic(workload, n_sub=None, seed=0):
    from glass_amd import synth
    w = synth.WORKLOADS[workload]
    n_batches = max(4, (n_sub or 16 * w.batch) // w.batch)
    w, ei, ew, _x, pos, y = synth.make_workload(workload, seed=seed, n_batches=n_batches)
    g = BaseGraph(torch.empty((w.n_node, 1, 0)), torch.from_numpy(ei), torch.from_numpy(ew), torch.from_numpy(pos),
                  torch.from_numpy(y).to(torch.float if w.multilabel else torch.int64), _split_mask(pos.shape[0]))
    return g


def load_dataset(name: str):
    """-> BaseGraph.  To use your own dataset, add a branch returning a BaseGraph here."""
    if name in ("coreness", "cut_ratio", "density", "component"):
        return _load_shipped(name)
    if name in ("ppi_bp", "hpo_metab", "hpo_neuro", "em_user"):
        return _load_subgnn_text(name)
    if name.startswith("synthetic:"):
        return _load_synthetic(name.split(":", 1)[1])
    raise NotImplementedError()
