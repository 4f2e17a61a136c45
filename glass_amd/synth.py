"""Synthetic graphs shaped like the benchmark configurations (BASELINE.json configs).

No dataset ships with the repo for the real-world sets and there is no network, so benchmarks and
large parity tests run on seeded synthetic data: numpy Generator(PCG64(seed)); DISTINCT undirected
pairs u != v, symmetrised (nnz = 2 * pairs exactly), sorted by (row, col); edge_weight = 1;
`use_deg`-style node features (rank of the node's degree among the distinct degrees — the
reference's setDegreeFeature, /root/reference/datasets.py:45-52); uniform random subgraphs
without replacement, padded with -1 to the widest one.
"""
from dataclasses import dataclass

import numpy as np


@dataclass
class Workload:
    name: str
    n_node: int
    n_pairs: int
    hidden: int
    layers: int
    aggr: str
    pool: str
    z_ratio: float
    dropout: float
    batch: int          # subgraphs per step (per rank)
    sub_size: int       # nodes per subgraph
    n_class: int
    multilabel: bool = False
    powerlaw: float = 0.0  # >0: endpoints ~ Zipf weights rank^-powerlaw
    lr: float = 1e-3


# BASELINE.json configs C1..C5.  C1 is the shipped density graph (dataset_/density/graph.npz: N = 4 998, 29 962 undirected
# edges, 250 labelled subgraphs x 20 nodes) at SURVEY.md §8d's model (hidden 64, 2 layers, config/density.yml otherwise);
# its n_pairs entry is informational — make_workload reads the real graph and its real subgraphs.
WORKLOADS = {
    "density": Workload("density", 4998, 29962, 64, 2, "sum", "size", 1.0, 0.0, 2, 20, 3),
    "ppi_bp": Workload("ppi_bp", 17080, 316951, 64, 2, "mean", "sum", 0.95, 0.5, 80, 10, 6, lr=0.0005),
    "hpo_neuro": Workload("hpo_neuro", 14587, 3238174, 64, 2, "gcn", "sum", 0.85, 0.5, 99, 15, 10, True, lr=0.002),
    "em_user": Workload("em_user", 50000, 500000, 128, 1, "gcn", "size", 0.75, 0.5, 6, 155, 2),
    "powerlaw": Workload("powerlaw", 1000000, 10000000, 256, 2, "mean", "sum", 0.9, 0.5, 64, 32, 6, powerlaw=0.8),
    # small shapes for tests
    "tiny": Workload("tiny", 300, 1500, 16, 2, "mean", "sum", 0.9, 0.0, 8, 6, 3),
}


def _distinct_pairs(rng, n, n_pairs, powerlaw=0.0, max_deg=50000):
    """Rejection-sample distinct undirected pairs (u<v) until exactly n_pairs are held."""
    if powerlaw > 0:
        w = np.arange(1, n + 1, dtype=np.float64)**(-powerlaw)
        cdf = np.cumsum(w / w.sum())
    keys = np.empty(0, dtype=np.int64)
    while keys.shape[0] < n_pairs:
        m = int((n_pairs - keys.shape[0]) * 1.2) + 1024
        if powerlaw > 0:
            u = np.searchsorted(cdf, rng.random(m)).astype(np.int64)
            v = np.searchsorted(cdf, rng.random(m)).astype(np.int64)
            u, v = np.minimum(u, n - 1), np.minimum(v, n - 1)
        else:
            u, v = rng.integers(0, n, m), rng.integers(0, n, m)
        ok = u != v
        lo, hi = np.minimum(u, v)[ok], np.maximum(u, v)[ok]
        new = np.unique(lo * n + hi)
        new = new[~np.isin(new, keys, assume_unique=True)] if keys.shape[0] else new
        rng.shuffle(new)
        keys = np.concatenate([keys, new[:n_pairs - keys.shape[0]]])
        if powerlaw > 0:  # cap the maximum degree: drop surplus edges of over-full hubs
            deg = np.bincount(np.concatenate([keys // n, keys % n]), minlength=n)
            hubs = np.nonzero(deg > max_deg)[0]
            for h in hubs:
                idx = np.nonzero((keys // n == h) | (keys % n == h))[0]
                keys = np.delete(keys, idx[max_deg:])
    return keys // n, keys % n


def make_graph(n_node, n_pairs, seed=0, powerlaw=0.0):
    """-> edge_index int64 [2, 2*n_pairs] sorted by (row, col), edge_weight float32 ones."""
    rng = np.random.Generator(np.random.PCG64(seed))
    lo, hi = _distinct_pairs(rng, n_node, n_pairs, powerlaw)
    row = np.concatenate([lo, hi])
    col = np.concatenate([hi, lo])
    order = np.argsort(row * n_node + col, kind="stable")
    ei = np.stack([row[order], col[order]])
    return ei, np.ones(ei.shape[1], dtype=np.float32)


def degree_feature(edge_index, n_node):
    """use_deg features: x[n] = index of deg(n) among the sorted distinct degrees; shape [N,1,1]."""
    deg = np.bincount(edge_index[0], minlength=n_node)
    x = np.unique(deg, return_inverse=True)[1].astype(np.int64)
    return x.reshape(n_node, 1, 1)


def make_subgraphs(n_node, n_sub, size, n_class, seed=1, multilabel=False):
    """-> pos int64 [n_sub, size] (uniform node sets without replacement), labels."""
    rng = np.random.Generator(np.random.PCG64(seed))
    pos = np.stack([rng.choice(n_node, size, replace=False) for _ in range(n_sub)]).astype(np.int64)
    if multilabel:
        y = (rng.random((n_sub, n_class)) < 0.3).astype(np.float32)
    else:
        y = rng.integers(0, n_class, n_sub).astype(np.int64)
    return pos, y


def _shipped_density(w, n_batches):
    """The real C1 graph: symmetrised, sorted by (row, col), use_deg features, the dataset's own subgraphs and labels
    (cycled when more batches are asked for than the 250 subgraphs give)."""
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dataset_", "density", "graph.npz"))
    e = z["edges"].astype(np.int64)
    row = np.concatenate([e[:, 0], e[:, 1]])
    col = np.concatenate([e[:, 1], e[:, 0]])
    order = np.argsort(row * w.n_node + col, kind="stable")
    ei = np.stack([row[order], col[order]])
    ew = np.ones(ei.shape[1], dtype=np.float32)
    idx = np.arange(w.batch * n_batches) % z["subG"].shape[0]
    return ei, ew, degree_feature(ei, w.n_node), z["subG"].astype(np.int64)[idx], z["label"].astype(np.int64)[idx]


def make_workload(name, seed=0, n_batches=4):
    """Graph + features + n_batches*batch subgraphs for a named workload (numpy arrays)."""
    w = WORKLOADS[name]
    if name == "density":
        return (w, *_shipped_density(w, n_batches))
    ei, ew = make_graph(w.n_node, w.n_pairs, seed, w.powerlaw)
    x = degree_feature(ei, w.n_node)
    pos, y = make_subgraphs(w.n_node, w.batch * n_batches, w.sub_size, w.n_class, seed + 1, w.multilabel)
    return w, ei, ew, x, pos, y
