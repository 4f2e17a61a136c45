"""Data-parallel path on CPU with the gloo backend, world_size = 2.

The N>1 path is: identical shuffle on every rank -> each rank takes perm[rank::world] of every
batch -> full-graph forward/backward on its slice -> ONE all-reduce of the flat gradient bucket,
divided by world size.  The collective, the sharding and the bucket are host logic and run here on
CPU tensors; the per-rank gradients come from the CPU oracle (the product model has no CPU path).
Checked: rank slices are disjoint and cover the batch; every rank ends with the same averaged
gradient; it equals the mean of single-process gradients on each rank's sub-batch (and NOT the
gradient at the global batch — labels are per batch)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as td
import torch.multiprocessing as mp
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
WORLD = 2


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _setup(rank, port):
    for p in (ROOT, HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(WORLD))
    torch.set_num_threads(2)
    td.init_process_group("gloo", rank=rank, world_size=WORLD)


def _oracle_model_and_data():
    from glass_amd import synth
    from oracle import glass_oracle as O
    w, ei, ew, x, pos, y = synth.make_workload("tiny", seed=0, n_batches=2)
    ei, ew, x, pos, y = (torch.from_numpy(a) for a in (ei, ew, x, pos, y))
    torch.manual_seed(0)
    model = O.OracleGLASS(w.hidden, w.layers, int(x.max()), w.n_class, aggr=w.aggr, pool=w.pool, z_ratio=w.z_ratio)
    return w, model, (x, ei, ew, pos, y)


def _grads_on(model, data, sel):
    from oracle import glass_oracle as O
    x, ei, ew, pos, y = data
    for c in model.conv.convs:
        c.adj = None
    model.zero_grad()
    p = pos[sel]
    loss = nn.CrossEntropyLoss()(model(x, ei, ew, p, O.max_zero_one(x, p)), y[sel])
    loss.backward()
    return torch.cat([q.grad.reshape(-1) for q in model.parameters()]).clone()


def _worker(rank, port, out_dir):
    _setup(rank, port)
    from glass_amd import dist as gdist
    from glass_amd.SubGDataset import GDataset, ZGDataloader
    from oracle import glass_oracle as O
    assert gdist.is_distributed() and gdist.rank() == rank and gdist.world_size() == WORLD
    w, model, data = _oracle_model_and_data()
    x, ei, ew, pos, y = data
    # --- loader: same permutation everywhere (rank 0's), disjoint strided shards --------------------
    torch.manual_seed(100 + rank)  # different local RNG per rank on purpose
    ds = GDataset(x, ei, ew, pos, torch.arange(pos.shape[0]))
    loader = ZGDataloader(ds, batch_size=8, shuffle=True, drop_last=True, z_fn=O.max_zero_one, shard=True)
    shards = [b[-1].clone() for b in loader]
    # evaluation loaders are NOT sharded: every rank sees the full batches (same scores / early stop everywhere);
    # a sharded loader skips, on every rank, a tail batch smaller than the world
    full = [b[-1].tolist() for b in ZGDataloader(ds, batch_size=8, shuffle=False, z_fn=O.max_zero_one)]
    assert full == [list(range(0, 8)), list(range(8, 16))]
    ds17 = GDataset(x, ei, ew, pos[:1].repeat(17, 1), torch.arange(17))
    tail = [b[-1].tolist() for b in ZGDataloader(ds17, batch_size=8, shuffle=False, z_fn=O.max_zero_one, shard=True)]
    assert [len(t) for t in tail] == [4, 4] and all(len(t) > 0 for t in tail)
    gathered = [None] * WORLD
    td.all_gather_object(gathered, [s.tolist() for s in shards])
    # --- one data-parallel step through the flat bucket -------------------------------------------
    bucket = gdist.FlatGradBucket(list(model.parameters()))
    bucket.zero()
    batch = torch.arange(8)
    mine = gdist.shard(batch)
    p = pos[mine]
    loss = nn.CrossEntropyLoss()(model(x, ei, ew, p, O.max_zero_one(x, p)), y[mine])
    loss.backward()
    local = bucket.flat.clone()
    bucket.all_reduce_mean()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), local=local.numpy(), reduced=bucket.flat.numpy(),
             mine=mine.numpy())
    if rank == 0:
        import json
        with open(os.path.join(out_dir, "shards.json"), "w") as f:
            json.dump(gathered, f)
    td.barrier()
    td.destroy_process_group()


def test_two_rank_data_parallel_step(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(port, str(tmp_path)), nprocs=WORLD, join=True)
    r = [np.load(tmp_path / f"rank{k}.npz") for k in range(WORLD)]
    # shards: rank r takes batch[r::2]
    assert r[0]["mine"].tolist() == [0, 2, 4, 6] and r[1]["mine"].tolist() == [1, 3, 5, 7]
    import json
    shards = json.load(open(tmp_path / "shards.json"))
    assert len(shards[0]) == len(shards[1]) == 2  # 16 subgraphs / batch 8, drop_last
    for b0, b1 in zip(*shards):
        assert len(b0) == len(b1) == 4 and not set(b0) & set(b1)
    assert sorted(sum(shards[0] + shards[1], [])) == list(range(16))  # one shared permutation, fully covered
    # every rank holds the same averaged gradient = mean of the local ones
    assert np.array_equal(r[0]["reduced"], r[1]["reduced"])
    assert np.allclose(r[0]["reduced"], (r[0]["local"] + r[1]["local"]) / 2, rtol=0, atol=1e-7)
    # == mean of single-process gradients on each rank's sub-batch
    for p in (ROOT, HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    w, model, data = _oracle_model_and_data()
    g0 = _grads_on(model, data, torch.tensor([0, 2, 4, 6]))
    g1 = _grads_on(model, data, torch.tensor([1, 3, 5, 7]))
    ref = ((g0 + g1) / 2).numpy()
    assert np.abs(r[0]["reduced"] - ref).max() <= 1e-6 * np.abs(ref).max()
    # ... and is NOT the global-batch gradient (max-zero-one labels are per batch)
    gg = _grads_on(model, data, torch.arange(8)).numpy()
    assert np.abs(gg - ref).max() > 1e-4 * np.abs(ref).max()


def test_collective_prediction_model():
    """dist.predict_collective_us: the stated model every N > 1 bench line carries — zero on one rank,
    latency-bound and growing with the ring depth for the 0.2 MB bucket of hidden 64, bandwidth-bound for an embedding-sized
    bucket; the constants travel with the prediction."""
    from glass_amd import dist
    small = {"small_allreduce": 200_000, "big_reduce_scatter": 0, "big_all_gather": 0}
    assert dist.predict_collective_us(small, 1)["total_us"] == 0.0
    t = [dist.predict_collective_us(small, n)["total_us"] for n in (2, 4, 8)]
    assert 10.0 < t[0] < t[1] < t[2] < 60.0
    big = dist.predict_collective_us({"small_allreduce": 3_200_000, "big_reduce_scatter": 10**9, "big_all_gather": 10**9}, 8)
    assert big["big_reduce_scatter_us"] > 1000.0 and big["total_us"] == big["small_allreduce_us"] + big["big_reduce_scatter_us"] + big["big_all_gather_us"]
    assert big["model"]["status"].startswith("assumed") and big["model"]["t_hop_us"] == dist.MODEL_T_HOP_US
