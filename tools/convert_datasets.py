"""One-time conversion of the reference's shipped synthetic datasets to a plain-array format.

    python tools/convert_datasets.py            (authoring container only: reads /root/reference)

The reference stores each synthetic dataset as a pickled dict {G: networkx.Graph, subG, subGLabel,
mask} (`/root/reference/dataset_/<name>/tmp.npy`, loaded at datasets.py:105-126).  This writes the
same DATA — undirected edge list in networkx edge order, padded subgraph node lists, integer labels —
as `dataset_/<name>/graph.npz` so that this repo's PyG-/pickle-free `datasets.load_dataset` can read
it on any box.  No reference source is copied; only data arrays."""
import os
import sys

import numpy as np

REF = "/root/reference/dataset_"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dataset_")

for name in ("density", "cut_ratio", "coreness", "component"):
    obj = np.load(os.path.join(REF, name, "tmp.npy"), allow_pickle=True).item()
    G = obj["G"]
    nodes = list(G.nodes)
    assert nodes == list(range(len(nodes))), "node ids must be contiguous"
    edges = np.array(list(G.edges), dtype=np.int32)  # [E,2] in networkx order (datasets.py:108-109)
    subg = obj["subG"]
    width = max(len(s) for s in subg)
    pad = np.full((len(subg), width), -1, dtype=np.int32)
    for i, s in enumerate(subg):
        pad[i, :len(s)] = s
    labels = np.array([ord(c) - ord("A") for c in obj["subGLabel"]], dtype=np.int64)  # datasets.py:116
    os.makedirs(os.path.join(OUT, name), exist_ok=True)
    path = os.path.join(OUT, name, "graph.npz")
    np.savez_compressed(path, n_node=len(nodes), edges=edges, subG=pad, label=labels)
    print(name, "nodes", len(nodes), "edges", edges.shape[0], "subgraphs", pad.shape, "classes", len(set(labels)),
          f"{os.path.getsize(path)/1024:.0f} KiB")
