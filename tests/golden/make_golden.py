"""Generate the golden fixtures in this directory by RUNNING THE REFERENCE ITSELF.

Run here (authoring container) only:   python tests/golden/make_golden.py
It imports `/root/reference/impl/{models,utils,metrics}.py` and `/root/reference/datasets.py`
unmodified (PyG symbols supplied by `pyg_stub.py`), feeds them seeded inputs and writes
`*.npz` files holding INPUTS and EXPECTED OUTPUTS only (no reference source, no pickles).
`/root/reference` does not exist on the GPU box; tests read only the committed `.npz`.

Fixtures:
  g1_buildadj.npz   buildAdj dense A for mean/sum/gcn: isolated node, duplicate edge, self-loop, weights
  g2_conv_*.npz     GLASSConv forward + all grads, N=32 H=8, per aggr
  g3_emb_*.npz      EmbZGConv forward for L in {1,2,3} x jk in {0,1}
  g4_pool.npz       Add/Mean/Max/Size pool on padded pos with a shared node (+ grads)
  g5_density_*.npz  full GLASS loss + every param grad on the shipped density graph, H=64 L=2 (use_deg)
  g6_utils.npz      MaxZOZ / pad2batch / batch2pad incl. the docstring examples
  g7_metrics.npz    binaryf1 / microf1
  g8_adam.npz       3 Adam steps' losses on a small graph
  g9_keys.npz       state_dict key/shape list for the GLASSTest.buildModel construction
  g10_edgegnn_*.npz SSL pre-training path: EdgeGNN (EmbGConv of MyGCNConv layers + MLP head), pred / loss / grads
  g11_defaults_*.npz the reference's CONSTRUCTOR DEFAULTS the fused kernels do not cover (VERDICT r3 missing #3): activation
                    nn.ReLU() (impl/models.py:125,192), gn=False (:194), MaxPool (:300-303), a width outside the fused
                    family (48) — full GLASS loss + every gradient, fp32 and fp64
"""
import functools
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
sys.dont_write_bytecode = True
sys.path.insert(0, HERE)
sys.path.insert(0, REF)

import pyg_stub  # noqa: E402

pyg_stub.install()
os.chdir(REF)

import torch  # noqa: E402
import torch.nn as nn  # noqa: E402
from impl import models, utils, metrics  # noqa: E402  (the reference)
import datasets as ref_datasets  # noqa: E402  (the reference's top-level datasets.py)

torch.set_num_threads(1)  # deterministic summation order for the fixtures


def save(name, **arrs):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in arrs.items()})
    print(f"{name}: {os.path.getsize(path)/1024:.1f} KiB  keys={len(arrs)}")


def sd_arrays(model, prefix="sd/"):
    return {prefix + k: v.detach().numpy().copy() for k, v in model.state_dict().items()}


def grad_arrays(model, prefix="grad/"):
    return {prefix + k: p.grad.detach().numpy().copy() for k, p in model.named_parameters()}


def randomize_(model, gen, scale_gn=True):
    """Non-trivial values for every parameter (incl. GraphNorm weight/bias/mean_scale) so the
    fixtures exercise all terms."""
    with torch.no_grad():
        for k, p in model.named_parameters():
            if k.endswith("mean_scale") or (k.endswith("weight") and p.dim() == 1):
                p.copy_(1.0 + 0.3 * torch.randn(p.shape, generator=gen))
            elif p.dim() == 1:
                p.copy_(0.2 * torch.randn(p.shape, generator=gen))
            # Linear / Embedding weights keep their default init


def small_graph(rng, n, n_pairs, weighted=False):
    pairs = set()
    while len(pairs) < n_pairs:
        u, v = rng.integers(0, n, 2)
        if u != v:
            pairs.add((min(u, v), max(u, v)))
    pairs = np.array(sorted(pairs), dtype=np.int64)
    ei = np.concatenate([pairs, pairs[:, ::-1]], 0).T
    ew = np.ones(ei.shape[1], np.float32)
    if weighted:
        w = rng.uniform(0.5, 2.0, len(pairs)).astype(np.float32)
        ew = np.concatenate([w, w])
    return ei, ew


def build_glass(hidden, layers, max_deg, out_ch, aggr, pool, z_ratio, dropout=0.0, jk=True):
    """Same construction as GLASSTest.py:129-175 (buildModel)."""
    conv = models.EmbZGConv(hidden, hidden, layers, max_deg=max_deg, activation=nn.ELU(inplace=True), jk=jk,
                            dropout=dropout,
                            conv=functools.partial(models.GLASSConv, aggr=aggr, z_ratio=z_ratio, dropout=dropout),
                            gn=True)
    mlp = nn.Linear(hidden * layers if jk else hidden, out_ch)
    pool_fn = {"mean": models.MeanPool, "max": models.MaxPool, "sum": models.AddPool, "size": models.SizePool}[pool]()
    return models.GLASS(conv, nn.ModuleList([mlp]), nn.ModuleList([pool_fn]))


# ------------------------------------------------------------------------------------------
def g1():
    # 12 nodes; node 11 isolated; edge (0,1) duplicated; self-loop on 3; non-unit weights
    ei = np.array([[0, 1, 0, 1, 2, 3, 3, 4, 5, 6, 7, 8, 9, 10, 2, 5, 0, 1],
                   [1, 0, 1, 0, 3, 2, 3, 5, 4, 7, 6, 9, 8, 9, 10, 10, 4, 7]], dtype=np.int64)
    ew = np.array([1.0, 1.0, 0.5, 0.5, 2.0, 2.0, 1.5, 1.0, 1.0, 0.25, 0.25, 3.0, 3.0, 1.0, 1.0, 0.75, 1.0, 1.25],
                  dtype=np.float32)
    out = {"edge_index": ei, "edge_weight": ew, "n_node": 12}
    for aggr in ("mean", "sum", "gcn"):
        adj = models.buildAdj(torch.from_numpy(ei), torch.from_numpy(ew), 12, aggr)
        out["A_" + aggr] = adj.to_dense().numpy()
    save("g1_buildadj.npz", **out)


def g2():
    rng = np.random.default_rng(2)
    n, h = 32, 8
    ei, ew = small_graph(rng, n, 70, weighted=True)
    x = rng.standard_normal((n, h)).astype(np.float32)
    mask = rng.random(n) < 0.3
    gout = rng.standard_normal((n, h)).astype(np.float32)
    for aggr in ("mean", "sum", "gcn"):
        gen = torch.Generator().manual_seed(20)
        torch.manual_seed(20)
        conv = models.GLASSConv(h, h, activation=nn.ELU(inplace=True), aggr=aggr, z_ratio=0.8, dropout=0.0)
        randomize_(conv, gen)
        xt = torch.from_numpy(x).requires_grad_(True)
        y = conv(xt, torch.from_numpy(ei), torch.from_numpy(ew), torch.from_numpy(mask).reshape(-1, 1))
        (y * torch.from_numpy(gout)).sum().backward()
        # the same reference module evaluated in float64 (ground truth for its own fp32 rounding noise)
        conv64 = models.GLASSConv(h, h, activation=nn.ELU(inplace=True), aggr=aggr, z_ratio=0.8, dropout=0.0).double()
        conv64.load_state_dict({k: v.double() for k, v in conv.state_dict().items()})
        xt64 = torch.from_numpy(x).double().requires_grad_(True)
        y64 = conv64(xt64, torch.from_numpy(ei), torch.from_numpy(ew).double(), torch.from_numpy(mask).reshape(-1, 1))
        (y64 * torch.from_numpy(gout).double()).sum().backward()
        save(f"g2_conv_{aggr}.npz", edge_index=ei, edge_weight=ew, x=x, mask=mask, gout=gout, z_ratio=0.8,
             y=y.detach().numpy(), grad_x=xt.grad.numpy(), y64=y64.detach().numpy(), grad_x64=xt64.grad.numpy(),
             **sd_arrays(conv), **grad_arrays(conv), **grad_arrays(conv64, "grad64/"))


def g3():
    rng = np.random.default_rng(3)
    n, h = 40, 8
    ei, ew = small_graph(rng, n, 90)
    xfeat = rng.integers(0, 6, (n, 1)).astype(np.int64)
    z = (rng.random(n) < 0.25).astype(np.int64)
    for layers in (1, 2, 3):
        for jk in (0, 1):
            torch.manual_seed(30 + layers * 2 + jk)
            gen = torch.Generator().manual_seed(31)
            emb = models.EmbZGConv(h, h, layers, max_deg=5, activation=nn.ELU(inplace=True), jk=bool(jk), dropout=0.0,
                                   conv=functools.partial(models.GLASSConv, aggr="mean", z_ratio=0.7, dropout=0.0),
                                   gn=True)
            randomize_(emb, gen)
            emb.eval()
            y = emb(torch.from_numpy(xfeat), torch.from_numpy(ei), torch.from_numpy(ew), torch.from_numpy(z))
            y_noz = emb(torch.from_numpy(xfeat), torch.from_numpy(ei), torch.from_numpy(ew), None)
            emb64 = models.EmbZGConv(h, h, layers, max_deg=5, activation=nn.ELU(inplace=True), jk=bool(jk),
                                     dropout=0.0, conv=functools.partial(models.GLASSConv, aggr="mean", z_ratio=0.7,
                                                                         dropout=0.0), gn=True).double()
            emb64.load_state_dict({k: v.double() for k, v in emb.state_dict().items()})
            emb64.eval()
            y64 = emb64(torch.from_numpy(xfeat), torch.from_numpy(ei), torch.from_numpy(ew).double(),
                        torch.from_numpy(z))
            save(f"g3_emb_L{layers}_jk{jk}.npz", edge_index=ei, edge_weight=ew, x=xfeat, z=z, z_ratio=0.7, aggr="mean",
                 layers=layers, jk=jk, hidden=h, y=y.detach().numpy(), y_noz=y_noz.detach().numpy(),
                 y64=y64.detach().numpy(), **sd_arrays(emb))


def g4():
    rng = np.random.default_rng(4)
    n, c = 20, 6
    emb = rng.standard_normal((n, c)).astype(np.float32)
    pos = np.array([[0, 2, 3, 7], [1, 4, 5, -1], [6, 7, -1, -1], [7, -1, -1, -1], [8, 9, 10, 11]], dtype=np.int64)
    gout = rng.standard_normal((pos.shape[0], c)).astype(np.float32)
    out = {"emb": emb, "pos": pos, "gout": gout}
    glass = models.GLASS(None, nn.ModuleList(), nn.ModuleList())
    for name, cls in (("sum", models.AddPool), ("mean", models.MeanPool), ("max", models.MaxPool),
                      ("size", models.SizePool)):
        e = torch.from_numpy(emb).requires_grad_(True)
        y = glass.Pool(e, torch.from_numpy(pos), cls())
        (y * torch.from_numpy(gout)).sum().backward()
        out["y_" + name] = y.detach().numpy()
        out["grad_" + name] = e.grad.numpy()
    save("g4_pool.npz", **out)


def density_graph():
    torch.manual_seed(0)
    g = ref_datasets.load_dataset("density")  # BaseGraph: symmetrised + coalesced (datasets.py:28,68-71)
    g.setDegreeFeature()
    return g


def g5():
    g = density_graph()
    n = g.x.shape[0]
    ei = g.edge_index.numpy()
    assert n < 32768
    # store one direction only (u<v); the symmetrised, (row,col)-sorted list is rebuilt by the test
    und = ei[:, ei[0] < ei[1]].astype(np.int16)
    max_deg = int(g.x.max())
    # (aggr, pool, z_ratio, batch) variants; labels: 3-class CE as GLASSTest.py:66-71
    variants = [("sum", "size", 1.0, [3, 117]), ("mean", "sum", 0.95, [0, 5, 9, 200, 249, 31, 77, 123]),
                ("gcn", "mean", 0.85, [10, 11, 12, 13, 14, 15])]
    for aggr, pool, zr, sel in variants:
        torch.manual_seed(5)
        model = build_glass(64, 2, max_deg, 3, aggr, pool, zr)
        gen = torch.Generator().manual_seed(55)
        randomize_(model, gen)
        pos = g.pos[sel]
        y = g.y[sel].to(torch.int64)
        z = utils.MaxZOZ(g.x, pos)
        model.train()  # dropout=0 -> identical to eval
        emb = model.NodeEmb(g.x, g.edge_index, g.edge_attr, z)
        pred = model.preds[0](model.Pool(emb, pos, model.pools[0]))
        loss = nn.CrossEntropyLoss()(pred, y)
        loss.backward()
        embn = emb.detach().numpy()
        # the same reference model evaluated in float64: the reference's own fp32 rounding noise is
        # |fp32 - fp64| (1e-5..3e-5 rel-inf here: PyG's scatter_mean sums the N rows sequentially in fp32)
        m64 = build_glass(64, 2, max_deg, 3, aggr, pool, zr).double()
        m64.load_state_dict({k: v.double() for k, v in model.state_dict().items()})
        m64.train()
        emb64 = m64.NodeEmb(g.x, g.edge_index, g.edge_attr.double(), z)
        pred64 = m64.preds[0](m64.Pool(emb64, pos, m64.pools[0]))
        loss64 = nn.CrossEntropyLoss()(pred64, y)
        loss64.backward()
        g64 = {k: p.grad.detach().numpy() for k, p in m64.named_parameters()}
        extra64 = {"grad64/" + k: v.astype(np.float32) for k, v in g64.items()}  # fp64 values, stored rounded
        extra64["gnorm64_keys"] = np.array(sorted(g64))
        extra64["gnorm64"] = np.array([np.sqrt((g64[k]**2).sum()) for k in sorted(g64)])  # float64 exact pins
        extra64["pred64"] = pred64.detach().numpy()
        extra64["loss64"] = loss64.item()
        extra64["emb_rows64"] = emb64.detach().numpy()[:16]
        extra64["emb_colsum64"] = emb64.detach().numpy().sum(0)
        save(f"g5_density_{aggr}.npz", **extra64, n_node=n, und_pairs=und, x=g.x.numpy().reshape(-1).astype(np.int16),
             pos=pos.numpy().astype(np.int16), y=y.numpy(), z=z.numpy().astype(np.uint8), aggr=aggr, pool=pool,
             z_ratio=zr, hidden=64, layers=2, max_deg=max_deg, pred=pred.detach().numpy(), loss=loss.item(),
             emb_rows=embn[:16], emb_absmax=np.abs(embn).max(),
             **{k: v.astype(np.float32) for k, v in sd_arrays(model).items()}, **grad_arrays(model))


def g6():
    batch = torch.tensor([0, 1, 0, 0, 1, 1, 2, 2])
    pad = utils.batch2pad(batch)
    b2, p2 = utils.pad2batch(pad)
    x = torch.zeros(9, 1, 1, dtype=torch.int64)
    pos = torch.tensor([[0, 2, 3], [1, 4, 5], [6, 2, -1]])
    save("g6_utils.npz", batch=batch.numpy(), pad=pad.numpy(), p2b_batch=b2.numpy(), p2b_pos=p2.numpy(),
         mz_pos=pos.numpy(), mz_n=9, mz_z=utils.MaxZOZ(x, pos).numpy())


def g7():
    rng = np.random.default_rng(7)
    pred_b = rng.standard_normal((50, 4)).astype(np.float32)
    lab_b = (rng.random((50, 4)) < 0.4).astype(np.float32)
    pred_m = rng.standard_normal((60, 5)).astype(np.float32)
    lab_m = rng.integers(0, 5, 60)
    save("g7_metrics.npz", pred_b=pred_b, lab_b=lab_b, f1_b=metrics.binaryf1(pred_b, lab_b), pred_m=pred_m,
         lab_m=lab_m, f1_m=metrics.microf1(pred_m, lab_m))


def g8():
    rng = np.random.default_rng(8)
    n = 60
    ei, ew = small_graph(rng, n, 150)
    # use_deg-style features
    deg = np.bincount(ei[0], minlength=n)
    xfeat = np.unique(deg, return_inverse=True)[1].reshape(n, 1, 1).astype(np.int64)
    pos = np.full((12, 5), -1, dtype=np.int64)
    for i in range(12):
        k = rng.integers(2, 6)
        pos[i, :k] = rng.choice(n, k, replace=False)
    y = rng.integers(0, 3, 12)
    torch.manual_seed(8)
    model = build_glass(16, 2, int(xfeat.max()), 3, "mean", "sum", 0.9)
    sd0 = sd_arrays(model)
    opt = torch.optim.Adam(model.parameters(), lr=1e-2)
    losses = []
    xt, eit, ewt = torch.from_numpy(xfeat), torch.from_numpy(ei), torch.from_numpy(ew)
    model.train()
    for step in range(3):
        sel = torch.arange(step * 4, step * 4 + 4)
        p = torch.from_numpy(pos)[sel]
        z = utils.MaxZOZ(xt, p)
        opt.zero_grad()
        loss = nn.CrossEntropyLoss()(model(xt, eit, ewt, p, z, id=0), torch.from_numpy(y)[sel])
        loss.backward()
        losses.append(loss.item())
        opt.step()
    save("g8_adam.npz", edge_index=ei, edge_weight=ew, x=xfeat, pos=pos, y=y, losses=np.array(losses), lr=1e-2,
         hidden=16, layers=2, aggr="mean", pool="sum", z_ratio=0.9, **sd0,
         **{"sd_end/" + k[3:]: v for k, v in sd_arrays(model).items()})


def g9():
    model = build_glass(64, 2, 1, 3, "mean", "sum", 0.8)
    keys = list(model.state_dict().keys())
    shapes = [list(v.shape) for v in model.state_dict().values()]
    save("g9_keys.npz", keys=np.array(keys), shapes=np.array([str(s) for s in shapes]),
         n_params=sum(p.numel() for p in model.parameters()))


def main():
    fns = {"g1": g1, "g2": g2, "g3": g3, "g4": g4, "g5": g5, "g6": g6, "g7": g7, "g8": g8, "g9": g9, "g10": g10, "g11": g11}
    only = [a for a in sys.argv[1:] if a in fns]
    for name, fn in fns.items():
        if not only or name in only:
            fn()


def g10():
    """SSL pre-training path: EdgeGNN = EmbGConv(MyGCNConv layers) + MLP head, link prediction
    with Pool = mean of the two endpoint embeddings (reference impl/models.py:361-509, GNNEmb.py:76-105)."""
    rng = np.random.default_rng(10)
    n, h = 40, 8
    ei, ew = small_graph(rng, n, 90)
    deg = np.bincount(ei[0], minlength=n)
    xfeat = np.unique(deg, return_inverse=True)[1].reshape(n, 1, 1).astype(np.int64)
    pairs = rng.integers(0, n, (24, 2)).astype(np.int64)
    y = (rng.random(24) < 0.5).astype(np.float32)
    for layers, jk, aggr in ((2, 0, "mean"), (3, 1, "gcn"), (1, 0, "sum")):
        torch.manual_seed(100 + layers)
        gen = torch.Generator().manual_seed(101)

        def build():
            conv = models.EmbGConv(h, h, h, layers, max_deg=int(xfeat.max()), activation=nn.ReLU(inplace=True),
                                   jk=bool(jk), dropout=0.0,
                                   conv=functools.partial(models.MyGCNConv, aggr=aggr), gn=True)
            head = models.MLP(h * layers if jk else h, h, 1, 2, dropout=0.0, activation=nn.ReLU(inplace=True))
            return models.EdgeGNN(conv, nn.ModuleList([head]), nn.ModuleList([models.MeanPool()]))

        m32 = build()
        randomize_(m32, gen)
        m64 = build().double()
        m64.load_state_dict({k: v.double() for k, v in m32.state_dict().items()})
        outs = {}
        for tag, m, dt in (("", m32, torch.float32), ("64", m64, torch.float64)):
            m.train()
            pred = m(torch.from_numpy(xfeat), torch.from_numpy(ei), torch.from_numpy(ew).to(dt), torch.from_numpy(pairs))
            loss = nn.BCEWithLogitsLoss()(pred.flatten(), torch.from_numpy(y).to(dt))
            loss.backward()
            outs["pred" + tag] = pred.detach().numpy()
            outs["loss" + tag] = loss.item()
            outs.update(grad_arrays(m, "grad" + tag + "/"))
        save(f"g10_edgegnn_L{layers}_jk{jk}_{aggr}.npz", edge_index=ei, edge_weight=ew, x=xfeat, pairs=pairs, y=y,
             layers=layers, jk=jk, aggr=aggr, hidden=h, **sd_arrays(m32), **outs)


def g11():
    """The reference with its own constructor defaults where the product's fused path has holes: ReLU (not the driver's
    ELU), gn=False, MaxPool, hidden 48 (no MFMA family, not narrow): N = 72, 2 layers, jk, 9 subgraphs of <= 7 nodes
    (ragged, one node shared by three subgraphs)."""
    rng = np.random.default_rng(11)
    n, h, layers = 72, 48, 2
    ei, ew = small_graph(rng, n, 260)
    deg = np.bincount(ei[0], minlength=n)
    xfeat = np.unique(deg, return_inverse=True)[1].reshape(n, 1, 1).astype(np.int64)
    pos = np.full((9, 7), -1, dtype=np.int64)
    for b in range(9):
        k = int(rng.integers(3, 8))
        pos[b, :k] = rng.choice(n, size=k, replace=False)
    pos[0, 0] = pos[3, 1] = pos[7, 2] = 5
    y = rng.integers(0, 3, 9).astype(np.int64)
    for tag, gn, pool, aggr, zr in (("relu_gn_max_mean", True, "max", "mean", 0.8), ("relu_nogn_sum_gcn", False, "sum", "gcn", 0.9),
                                   ("relu_gn_size_sum", True, "size", "sum", 0.95)):
        torch.manual_seed(110)
        gen = torch.Generator().manual_seed(111)

        def build():
            conv = models.EmbZGConv(h, h, layers, max_deg=int(xfeat.max()), activation=nn.ReLU(), jk=True, dropout=0.0,
                                    conv=functools.partial(models.GLASSConv, aggr=aggr, z_ratio=zr, dropout=0.0), gn=gn)
            pool_fn = {"mean": models.MeanPool, "max": models.MaxPool, "sum": models.AddPool, "size": models.SizePool}[pool]()
            return models.GLASS(conv, nn.ModuleList([nn.Linear(h * layers, 3)]), nn.ModuleList([pool_fn]))

        m32 = build()
        randomize_(m32, gen)
        m64 = build().double()
        m64.load_state_dict({k: v.double() for k, v in m32.state_dict().items()})
        outs = {}
        x_t, pos_t = torch.from_numpy(xfeat), torch.from_numpy(pos)
        z = utils.MaxZOZ(x_t, pos_t)
        for t, m, dt in (("", m32, torch.float32), ("64", m64, torch.float64)):
            m.train()
            pred = m(x_t, torch.from_numpy(ei), torch.from_numpy(ew).to(dt), pos_t, z)
            loss = nn.CrossEntropyLoss()(pred, torch.from_numpy(y))
            loss.backward()
            outs["pred" + t] = pred.detach().numpy()
            outs["loss" + t] = loss.item()
            if t == "64":
                outs.update(grad_arrays(m, "grad64/"))
        save(f"g11_defaults_{tag}.npz", edge_index=ei, edge_weight=ew, x=xfeat, pos=pos, y=y, z=z.numpy(), gn=int(gn), pool=pool,
             aggr=aggr, z_ratio=zr, hidden=h, layers=layers, **sd_arrays(m32), **outs)


if __name__ == "__main__":
    main()
