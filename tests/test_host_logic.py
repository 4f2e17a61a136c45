"""CPU tests of the host-side mirror of the reference interface (no GPU compute): module surface,
index helpers, loaders, metrics, config, synthetic generators, gradient bucket."""
import os

import numpy as np
import pytest
import torch
import torch.nn as nn

from helpers import load, build_glass

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_drop_in_surface():
    """`from impl import models, SubGDataset, train, metrics, utils, config` (GLASSTest.py:1) and every
    name a driver touches."""
    from impl import models, SubGDataset, train, metrics, utils, config
    for name in ("Seq", "MLP", "buildAdj", "GLASSConv", "EmbZGConv", "PoolModule", "AddPool", "MaxPool", "MeanPool",
                 "SizePool", "GLASS"):
        assert hasattr(models, name)
    for name in ("MaxZOZ", "pad2batch", "batch2pad"):
        assert hasattr(utils, name)
    for name in ("GDataset", "GDataloader", "ZGDataloader"):
        assert hasattr(SubGDataset, name)
    assert callable(train.train) and callable(train.test)
    assert callable(metrics.binaryf1) and callable(metrics.microf1) and callable(metrics.auroc)
    config.set_device(-1)
    assert config.device == torch.device("cpu")
    config.set_device("cpu")
    assert config.device == torch.device("cpu")
    config.set_device(0)
    assert config.device.type in ("cuda", "cpu")


def test_state_dict_keys_match_reference():
    g = load("g9_keys.npz")
    model = build_glass(64, 2, 1, 3, "mean", "sum", 0.8)
    sd = model.state_dict()
    assert list(sd.keys()) == [str(k) for k in g["keys"]]
    assert [str(list(v.shape)) for v in sd.values()] == [str(s) for s in g["shapes"]]
    assert sum(p.numel() for p in model.parameters()) == int(g["n_params"])


def test_constructor_contract():
    """max_deg may be a 0-dim tensor (GLASSTest.py:99,142); input_emb is re-assignable (GLASSTest.py:157);
    GLASSConv works under functools.partial; unknown pool raises NotImplementedError in buildModel style."""
    import functools
    from impl import models
    conv = models.EmbZGConv(8, 8, 2, max_deg=torch.tensor(5), activation=nn.ELU(inplace=True), jk=True, dropout=0.1,
                            conv=functools.partial(models.GLASSConv, aggr="gcn", z_ratio=0.7, dropout=0.1), gn=True)
    assert conv.input_emb.weight.shape == (6, 8)
    conv.input_emb = nn.Embedding.from_pretrained(torch.randn(11, 8), freeze=False)
    assert conv.state_dict()["input_emb.weight"].shape == (11, 8)
    assert conv.gns[-1].weight.shape == (16, )  # JK: H * L channels
    assert conv.convs[0].comb_fns[0].weight.shape == (8, 16)  # input order [aggregated || x_]
    nojk = models.EmbZGConv(8, 4, 3, max_deg=2, jk=False)
    assert nojk.gns[-1].weight.shape == (4, ) and len(nojk.gns) == 3


def test_mlp_and_seq_structure():
    from impl import models
    m = models.MLP(4, 8, 2, 3, dropout=0.5, tail_activation=False, gn=False)
    kinds = [type(x).__name__ for x in m.seq.modlist]
    assert kinds == ["Linear", "Dropout", "ReLU", "Linear", "Dropout", "ReLU", "Linear"]
    m1 = models.MLP(4, 8, 2, 1, tail_activation=True, gn=True)
    assert [type(x).__name__ for x in m1.seq.modlist] == ["Linear", "GraphNorm", "ReLU"]
    y = models.MLP(4, 8, 2, 2)(torch.randn(5, 4))  # plain torch modules: runs anywhere
    assert y.shape == (5, 2)
    s = models.Seq([nn.Linear(3, 3), nn.ReLU()])
    assert s(torch.randn(2, 3)).shape == (2, 3)


def test_product_model_refuses_cpu_tensors():
    """No CPU fallback in the product path: the HIP path is the only path."""
    from glass_amd._lib import GlassHipError
    model = build_glass(8, 1, 3, 2, "mean", "sum", 0.8)
    x = torch.zeros(6, 1, 1, dtype=torch.int64)
    ei = torch.tensor([[0, 1], [1, 0]])
    with pytest.raises(GlassHipError):
        model(x, ei, torch.ones(2), torch.tensor([[0, 1]]), torch.zeros(6, dtype=torch.int64))


def test_pad2batch_batch2pad_docstring_examples():
    from impl import utils
    pad = utils.batch2pad(torch.tensor([0, 1, 0, 0, 1, 1, 2, 2]))  # impl/utils.py:9
    assert pad.tolist() == [[0, 2, 3], [1, 4, 5], [6, 7, -1]]
    b, p = utils.pad2batch(pad)  # impl/utils.py:21 (row-major order)
    assert b.tolist() == [0, 0, 0, 1, 1, 1, 2, 2] and p.tolist() == [0, 2, 3, 1, 4, 5, 6, 7]
    g = load("g6_utils.npz")
    assert np.array_equal(utils.batch2pad(torch.from_numpy(g["batch"])).numpy(), g["pad"])
    b, p = utils.pad2batch(torch.from_numpy(g["pad"]))
    assert np.array_equal(b.numpy(), g["p2b_batch"]) and np.array_equal(p.numpy(), g["p2b_pos"])
    # negative batch entries are dropped, empty rows impossible, ragged widths padded
    pad = utils.batch2pad(torch.tensor([3, -1, 3, 7, 7, 7]))
    assert pad.tolist() == [[0, 2, -1], [3, 4, 5]]


def test_metrics_golden():
    from impl import metrics
    g = load("g7_metrics.npz")
    assert metrics.binaryf1(g["pred_b"], g["lab_b"]) == pytest.approx(float(g["f1_b"]), abs=1e-12)
    assert metrics.microf1(g["pred_m"], g["lab_m"]) == pytest.approx(float(g["f1_m"]), abs=1e-12)
    assert 0.0 <= metrics.auroc(np.array([0.1, 0.9, 0.4, 0.8]), np.array([0, 1, 0, 1])) <= 1.0


def _dataset(n_sub=23, n=50):
    from impl import SubGDataset
    g = torch.Generator().manual_seed(0)
    x = torch.zeros(n, 1, 1, dtype=torch.int64)
    ei = torch.randint(0, n, (2, 100), generator=g)
    pos = torch.randint(-1, n, (n_sub, 5), generator=g)
    return SubGDataset.GDataset(x, ei, torch.ones(100), pos, torch.arange(n_sub))


def test_loader_tuple_layout_and_batching():
    from impl import SubGDataset
    ds = _dataset()
    assert len(ds) == 23 and ds.num_nodes == 50 and ds[3][1].item() == 3
    ld = SubGDataset.GDataloader(ds, batch_size=5, shuffle=False, drop_last=True)
    batches = list(ld)
    assert len(batches) == len(ld) == 4
    x, ei, ea, pos, y = batches[0]
    assert x is ds.x and ei is ds.edge_index and ea is ds.edge_attr
    assert pos.shape == (5, 5) and y.tolist() == [0, 1, 2, 3, 4]
    ld = SubGDataset.GDataloader(ds, batch_size=5, shuffle=True, drop_last=False)
    ys = torch.cat([b[-1] for b in ld])
    assert sorted(ys.tolist()) == list(range(23))  # a permutation, last partial batch kept
    seen = {}
    zl = SubGDataset.ZGDataloader(ds, 4, shuffle=False, drop_last=False, z_fn=lambda x, p: seen.setdefault("p", p))
    b = next(iter(zl))
    assert len(b) == 6 and b[4] is seen["p"] and torch.equal(b[3], ds.pos[:4])  # (x, ei, ea, pos, z, y)
    default = SubGDataset.ZGDataloader(ds, 4, shuffle=False)
    assert next(iter(default))[4].shape == (50, 1)  # default z_fn: zeros [N, x.shape[1]]


def test_synthetic_generators_are_seeded_and_well_formed():
    from glass_amd import synth
    w, ei, ew, x, pos, y = synth.make_workload("tiny", seed=0, n_batches=2)
    w2, ei2, *_ = synth.make_workload("tiny", seed=0, n_batches=2)
    assert np.array_equal(ei, ei2)
    assert ei.shape == (2, 2 * w.n_pairs) and np.all(ei[0] != ei[1])
    key = ei[0] * w.n_node + ei[1]
    assert np.all(np.diff(key) > 0)  # sorted by (row, col), no duplicates
    assert set(map(tuple, ei.T)) == set(map(tuple, ei[::-1].T))  # symmetric
    assert x.shape == (w.n_node, 1, 1) and pos.shape == (2 * w.batch, w.sub_size)
    for row in pos:
        assert len(set(row.tolist())) == w.sub_size  # without replacement
    ei_p, _ = synth.make_graph(3000, 20000, seed=1, powerlaw=0.9)
    deg = np.bincount(ei_p[0], minlength=3000)
    assert deg.max() > 20 * np.median(deg[deg > 0])  # skewed


def test_flat_grad_bucket_aliases_grads():
    from glass_amd.dist import FlatGradBucket
    lin = nn.Sequential(nn.Linear(4, 3), nn.Linear(3, 2))
    b = FlatGradBucket(list(lin.parameters()))
    assert b.flat.numel() == sum(p.numel() for p in lin.parameters()) and b.attached()
    lin(torch.randn(5, 4)).sum().backward()
    ref = torch.cat([p.grad.reshape(-1) for p in lin.parameters()])
    assert torch.equal(ref, b.flat) and b.flat.abs().sum() > 0  # autograd accumulated INTO the bucket
    b.zero()
    assert all(float(p.grad.abs().sum()) == 0 for p in lin.parameters())
    for p in lin.parameters():
        p.grad = None
    assert not b.attached()
    b.zero()  # re-attaches
    assert b.attached()


def test_datasets_loader_reproduces_reference_density():
    """datasets.load_dataset('density') (plain-array file, no pickle / networkx / PyG) vs the graph the
    REFERENCE's loader produced when fixture g5 was generated: same symmetrised edges, same degree feature,
    same padded subgraphs; split mask sizes 50/25/25 %."""
    import datasets
    g5 = load("g5_density_mean.npz")
    torch.manual_seed(0)
    g = datasets.load_dataset("density")
    g.setDegreeFeature()
    n = int(g5["n_node"])
    und = g5["und_pairs"].astype(np.int64)
    ref_keys = np.sort(np.concatenate([und[0] * n + und[1], und[1] * n + und[0]]))
    mine = (g.edge_index[0] * n + g.edge_index[1]).numpy()
    assert np.array_equal(mine, ref_keys)  # (row,col)-sorted, symmetric, coalesced
    assert torch.all(g.edge_attr == 1) and g.x.shape == (n, 1, 1)
    assert np.array_equal(g.x.reshape(-1).numpy(), g5["x"].astype(np.int64))
    sel = [0, 5, 9, 200, 249, 31, 77, 123]  # the subgraphs of the g5 'mean' variant
    assert np.array_equal(g.pos[sel].numpy(), g5["pos"].astype(np.int64))
    assert np.array_equal(g.y[sel].numpy(), g5["y"])
    assert g.mask.bincount().tolist() == [125, 62, 63]
    x, ei, ea, pos, y = g.get_split("valid")
    assert pos.shape == (62, 20) and y.shape == (62, )


def test_datasets_other_sources():
    import datasets
    for name, n, e2 in (("cut_ratio", 5000, 2 * 84295), ("coreness", 5000, 2 * 119205), ("component", 17260, 2 * 91619)):
        g = datasets.load_dataset(name)
        assert g.num_nodes == n and g.edge_index.shape == (2, e2)
    g = datasets.load_dataset("synthetic:tiny")
    g.setOneFeature()
    assert g.x.unique().tolist() == [1] and g.edge_index.shape[1] == 3000
    g.setNodeIdFeature()
    assert g.x.reshape(-1).tolist() == list(range(g.num_nodes))
    with pytest.raises(FileNotFoundError):
        datasets.load_dataset("ppi_bp")  # real-world sets are not shipped
    with pytest.raises(NotImplementedError):
        datasets.load_dataset("no_such_dataset")
    # to_undirected: directed input with a duplicate -> symmetrised, duplicate weights add
    bg = datasets.BaseGraph(torch.empty(4, 1, 0), torch.tensor([[0, 0, 2], [1, 1, 3]]), torch.tensor([1.0, 2.0, 1.0]),
                            torch.tensor([[0, 1]]), torch.tensor([0]), torch.tensor([0]))
    assert bg.edge_index.tolist() == [[0, 1, 2, 3], [1, 0, 3, 2]] and bg.edge_attr.tolist() == [3.0, 3.0, 1.0, 1.0]


def test_link_prediction_dataset():
    """get_LPdataset (SSL pre-training): positives = all edges, negatives = sampled non-edges, optional loop probes."""
    import datasets
    torch.manual_seed(0)
    g = datasets.load_dataset("synthetic:tiny")
    g.setDegreeFeature()
    x, ei, ea, pos, y = g.get_LPdataset()
    nnz, n = ei.shape[1], g.num_nodes
    assert pos.shape == (2 * nnz, 2) and y.shape == (2 * nnz, ) and y[:nnz].min() == 1 and y[nnz:].max() == 0
    assert torch.equal(pos[:nnz], ei.t())
    edges = set((ei[0] * n + ei[1]).tolist())
    assert not edges & set((pos[nnz:, 0] * n + pos[nnz:, 1]).tolist())  # negatives are non-edges
    assert len(set((pos[nnz:, 0] * n + pos[nnz:, 1]).tolist())) == nnz  # distinct
    x, ei, ea, pos2, y2 = g.get_LPdataset(use_loop=True)
    assert pos2.shape[0] == 2 * nnz + n and torch.equal(pos2[-n:, 0], pos2[-n:, 1]) and y2[-n:].sum() == 0


def test_subgnn_text_loader_deduplicates_undirected_edges(tmp_path, monkeypatch):
    """Real-world format (reference datasets.py:131-227 reads edge_list.txt through networkx.read_edgelist into an
    undirected simple graph): repeated and reversed lines are one edge; valid / test are swapped so valid is larger."""
    import datasets
    root = tmp_path / "dataset" / "toy"
    root.mkdir(parents=True)
    (root / "edge_list.txt").write_text("0 1\n1 0\n1 2\n1 2\n2 3\n3 3\n")
    (root / "subgraphs.pth").write_text("0-1\tA\ttrain\n1-2-3\tB\ttrain\n2-3\tA\tval\n0\tB\ttest\n3\tA\ttest\n")
    monkeypatch.chdir(tmp_path)
    g = datasets._load_subgnn_text("toy")
    # BaseGraph symmetrises on construction (coalesce with add): every undirected edge once per direction with weight 1
    # (2 would mean a duplicate line survived); the self-loop meets itself and gets 2, as in the reference
    got = {tuple(p): w for p, w in zip(g.edge_index.t().tolist(), g.edge_attr.tolist())}
    assert got == {(0, 1): 1.0, (1, 0): 1.0, (1, 2): 1.0, (2, 1): 1.0, (2, 3): 1.0, (3, 2): 1.0, (3, 3): 2.0}
    assert g.pos.shape == (5, 3)
    assert g.mask.tolist() == [0, 0, 1, 1, 2]  # the larger of val / test becomes valid


def test_loss_callables_are_recognised_by_what_they_compute():
    """glass_amd.losses.fusable_mode: the reference driver's binary loss is a plain function around BCEWithLogitsLoss
    (GLASSTest.py:57-58) — recognised by evaluating it; anything that computes something else is not."""
    import torch.nn as nn
    from torch.nn import BCEWithLogitsLoss, CrossEntropyLoss
    from glass_amd import losses

    def loss_fn(x, y):  # verbatim GLASSTest.py:57-58
        return BCEWithLogitsLoss()(x.flatten(), y.flatten())

    assert losses.fusable_mode(loss_fn) == 1 and losses.fusable_mode(loss_fn) == 1  # (second call: the cached verdict)
    assert losses.fusable_mode(CrossEntropyLoss()) == 0
    assert losses.fusable_mode(lambda p, t: nn.functional.cross_entropy(p, t)) == 0
    assert losses.fusable_mode(losses.BCEWithLogits()) == 1 and losses.fusable_mode(losses.CrossEntropy()) == 0
    for other in (CrossEntropyLoss(label_smoothing=0.1), CrossEntropyLoss(reduction="sum"), CrossEntropyLoss(ignore_index=0),
                  CrossEntropyLoss(weight=torch.tensor([1.0, 2.0, 3.0])), nn.MSELoss(),
                  lambda p, t: BCEWithLogitsLoss(reduction="sum")(p.flatten(), t.flatten()),
                  lambda p, t: BCEWithLogitsLoss(pos_weight=torch.tensor(2.0))(p.flatten(), t.flatten()),
                  lambda p, t: 0.5 * nn.functional.cross_entropy(p, t), lambda p, t: 1.0, None, 3):
        assert losses.fusable_mode(other) is None, other


def test_only_a_plain_adam_over_the_whole_model_is_adopted():
    """glass_amd.optim.adoptable names the reason a caller's optimizer keeps the eager loop."""
    import torch.nn as nn
    from glass_amd import optim
    m = nn.Linear(4, 3)
    assert "SGD" in optim.adoptable(torch.optim.SGD(m.parameters(), lr=0.1), m)
    assert "AdamW" in optim.adoptable(torch.optim.AdamW(m.parameters()), m)
    assert "amsgrad" in optim.adoptable(torch.optim.Adam(m.parameters(), amsgrad=True), m)
    assert "group" in optim.adoptable(torch.optim.Adam([{"params": [m.weight]}, {"params": [m.bias], "lr": 0.1}]), m)
    assert "exactly" in optim.adoptable(torch.optim.Adam([m.weight]), m)
    assert "GPU" in optim.adoptable(torch.optim.Adam(m.parameters()), m)  # (CPU parameters: the reference's --device -1 case)


@pytest.mark.skipif(not os.path.exists("/root/reference/GLASSTest.py"), reason="authoring container only: needs /root/reference")
def test_reference_driver_file_runs_over_this_repos_modules_up_to_the_first_kernel():
    """The drop-in claim at its source: /root/reference/GLASSTest.py itself — unmodified, executed with this repo's root first
    on sys.path — imports `impl.*` and `datasets` from HERE, parses its flags, loads and splits the shipped density set, runs
    its own `buildModel`, builds its loaders, `Adam(gnn.parameters(), lr)` and `ReduceLROnPlateau`, and calls
    `impl.train.train`.  Without a GPU that call stops at the first kernel with GlassHipError (there is deliberately no CPU
    fallback) — everything before it is the module surface working as the reference expects.  On a GPU box the
    same objects take the captured step program: tests/test_gpu_reference_caller.py."""
    import subprocess
    import sys
    code = (
        "import sys, runpy\n"
        "sys.dont_write_bytecode = True\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "sys.argv = ['GLASSTest.py', '--use_one', '--use_seed', '--use_maxzeroone', '--repeat', '1', '--device', '-1', '--dataset', 'density']\n"
        "try:\n"
        "    runpy.run_path('/root/reference/GLASSTest.py', run_name='__main__')\n"
        "except BaseException as e:\n"
        "    import impl.models, impl.train, datasets\n"
        "    print('ENDED', type(e).__name__, '|', impl.models.__file__, '|', impl.train.__file__, '|', datasets.__file__)\n"
        "    import traceback; traceback.print_exc()\n")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, cwd=ROOT,
                         env=dict(os.environ, PYTHONDONTWRITEBYTECODE="1"))
    text = out.stdout + out.stderr
    assert "ENDED GlassHipError" in text, text[-3000:]
    ended = [ln for ln in out.stdout.splitlines() if ln.startswith("ENDED")][0]
    assert ended.count(os.path.join(ROOT, "glass_amd")) == 2 and os.path.join(ROOT, "datasets.py") in ended, ended
    assert "repeat 0" in out.stdout                      # reached the training loop of the reference's test()
    assert "glass_amd/train.py" in text and "in train" in text  # ... and died inside impl.train.train, at a kernel call


def test_deepcopy_of_a_model_is_a_plain_independent_copy():
    """copy.deepcopy(model) — an early-stopping snapshot in a caller's loop — gives a plain module: same weights, own storage,
    none of the runtime attachments the training loops hang on a model (those hold device pointers and captured graphs;
    glass_amd.arena.strip_runtime).  The GPU side — a copy taken AFTER impl.train.train adopted the optimizer and built the
    arena — is tests/test_gpu_reference_caller.py::test_deepcopy_after_training_gives_an_independent_plain_model."""
    import copy
    from glass_amd import arena
    from glass_amd.utils import RuntimeCache
    torch.manual_seed(0)
    m = build_glass(16, 2, 5, 3, "mean", "sum", 0.8)
    m.__dict__["_glass_train_steps"] = RuntimeCache(a=object())   # what train.train would leave behind
    m.conv.__dict__["_glass_arena"] = object()
    c = copy.deepcopy(m)
    assert all(torch.equal(a, b) and a.data_ptr() != b.data_ptr() for a, b in zip(m.state_dict().values(), c.state_dict().values()))
    assert not any(k in mod.__dict__ for mod in c.modules() for k in arena._RUNTIME_ATTRS)
    assert "_glass_train_steps" in m.__dict__ and m.conv.__dict__["_glass_arena"] is not None   # the original keeps its own
    assert copy.deepcopy(RuntimeCache(x=1)) == {} and isinstance(copy.deepcopy(RuntimeCache(x=1)), RuntimeCache)


def test_opaque_loss_probe_can_be_declined():
    """losses.fusable_mode classifies an opaque callable by CALLING it on synthetic tensors: a callable marked
    `_glass_no_fuse` is never called by the probe and never fused; an unmarked one that computes the reference's binary loss is
    recognised (GLASSTest.py:57-58)."""
    import torch
    from glass_amd import losses

    calls = []

    def counting_bce(x, y):
        calls.append(1)
        return torch.nn.BCEWithLogitsLoss()(x.flatten(), y.flatten())

    assert losses.fusable_mode(counting_bce) == 1 and len(calls) > 0
    n = len(calls)
    assert losses.fusable_mode(counting_bce) == 1 and len(calls) == n   # cached verdict: no further calls

    def stateful(x, y):
        calls.append(2)
        return torch.nn.BCEWithLogitsLoss()(x.flatten(), y.flatten())

    stateful._glass_no_fuse = True
    n = len(calls)
    assert losses.fusable_mode(stateful) is None and len(calls) == n


def test_scaling_forecast_model():
    """dist.predict_scaling: the forecast every bench line prints — efficiency 1 at one rank, falling with the
    ring's depth, the one-shot form flat in N and above the ring from 2 ranks on; an embedding-sized bucket that overlaps the
    backward tail hides the small all-reduce."""
    from glass_amd import dist
    small = {"small_allreduce": 200_000, "big_reduce_scatter": 0, "big_all_gather": 0}
    fc = dist.predict_scaling(small, 0.232)
    eff = {r["world"]: r for r in fc["per_world"]}
    assert eff[1]["ring_efficiency"] == 1.0 and eff[1]["oneshot_efficiency"] == 1.0
    assert 1.0 > eff[2]["ring_efficiency"] > eff[4]["ring_efficiency"] > eff[8]["ring_efficiency"] > 0.8
    assert abs(eff[8]["ring_exposed_us"] - (12.0 + 14 * 1.5 + 2 * 200_000 / 8 / (153.0 * 0.7 * 1e3))) < 1e-6
    assert all(eff[n]["oneshot_efficiency"] > eff[n]["ring_efficiency"] for n in (2, 4, 8))
    assert abs(eff[2]["oneshot_exposed_us"] - eff[8]["oneshot_exposed_us"]) < 1e-9
    big = {"small_allreduce": 200_000, "big_reduce_scatter": 25_600_000, "big_all_gather": 25_600_000}
    a = dist.predict_scaling(big, 0.6, overlaps_small=False)["per_world"][3]
    b = dist.predict_scaling(big, 0.6, overlaps_small=True)["per_world"][3]
    assert b["ring_exposed_us"] < a["ring_exposed_us"] and b["ring_efficiency"] > a["ring_efficiency"]
