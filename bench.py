"""Headline benchmark: aggregated edges/sec of the GLASS labeled message-passing step.

    python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run)

A step = MaxZOZ + GLASS.forward + loss + backward + (gradient all-reduce) + Adam.step on one batch
of subgraphs per rank, over the WHOLE graph (the reference propagates over the full graph every
step, /root/reference/impl/train.py:10-16).  metric = nnz * L * steps * world / wall_seconds
(SURVEY.md §8d), inputs resident in HBM before the timed region.  At N=1 the workload is BASELINE
config[1]: the ppi_bp-shaped synthetic graph, hidden=64 (config/ppi_bp.yml hyper-parameters incl.
dropout 0.5).  Multi-GPU = subgraph-batch data parallelism: replicated graph, per-rank batch fixed
("weak"), one flat-bucket RCCL all-reduce per step.

The JSON line also carries
  roofline     : the CSR aggregation kernel (K1) — algorithmic bytes nnz*(4H+8)+N*(4H+4) per launch
                 divided by its average duration, measured with HIP events on the launch stream in a
                 second, instrumented pass of the same step loop — against the 8 TB/s HBM peak.
  cpu_baseline : the oracle (CPU restatement of the reference path, torch ops on host cores) timed
                 on a bounded sample of the same workload on rank 0 at N=1.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="ppi_bp", help="ppi_bp | hpo_neuro | em_user | powerlaw | tiny")
    ap.add_argument("--dropout", type=float, default=None, help="override the workload's YAML dropout")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=0, help="CPU baseline steps (0 = size to ~15 s)")
    ap.add_argument("--graph", type=int, default=1, help="1: replay the step from a captured hipGraph when possible")
    return ap.parse_args()


def loss_fn_for(w):
    """The driver's losses (GLASSTest.py:57-58, 69); glass_amd.losses' versions compute the same values and
    let TrainStep fuse them with the Linear head."""
    from glass_amd import losses
    return losses.BCEWithLogits() if w.multilabel else losses.CrossEntropy()


def cpu_baseline(w, ei, ew, x, pos, y, steps):
    """The reference CPU path (--device -1) as restated by oracle/: same step, host cores."""
    from oracle import glass_oracle as O
    torch.manual_seed(0)
    model = O.OracleGLASS(w.hidden, w.layers, int(x.max()), w.n_class, aggr=w.aggr, pool=w.pool, z_ratio=w.z_ratio,
                          dropout=w.dropout)
    model.train()
    opt = torch.optim.Adam(model.parameters(), lr=w.lr)
    loss_fn = (lambda p, t: nn.BCEWithLogitsLoss()(p.flatten(), t.flatten())) if w.multilabel else nn.CrossEntropyLoss()
    nb = pos.shape[0] // w.batch

    def step(i):
        sl = slice((i % nb) * w.batch, (i % nb + 1) * w.batch)
        return O.train_step(model, opt, loss_fn, x, ei, ew, pos[sl], y[sl])

    t0 = time.perf_counter()
    step(0)  # warm-up (builds the COO adjacency, like the reference's first forward)
    first = time.perf_counter() - t0
    t0 = time.perf_counter()
    step(1)
    one = time.perf_counter() - t0
    if steps <= 0:
        steps = max(3, min(50, int(15.0 / max(one, 1e-3))))
    t0 = time.perf_counter()
    for i in range(steps):
        step(2 + i)
    dt = time.perf_counter() - t0
    nnz = ei.shape[1]
    return {"value": nnz * w.layers * steps / dt, "unit": "edges/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{steps} steps of the same workload (oracle/glass_oracle.py, torch CPU ops, {dt:.1f} s, "
                      f"{dt / steps * 1e3:.0f} ms/step, first step {first:.1f} s)",
            "ms_per_step": dt / steps * 1e3}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path is the only product path)")
    n_dev = torch.cuda.device_count()
    backend = os.environ.get("GLASS_BENCH_BACKEND", "nccl")  # "gloo": smoke-test the N>1 path on a 1-GPU box
    if backend == "gloo":
        local_rank = local_rank % n_dev
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as td
        if backend == "nccl":
            td.init_process_group("nccl", device_id=dev)  # RCCL over xGMI
        else:
            td.init_process_group(backend)

    from glass_amd import synth, ops, graph as ggraph, dist as gdist
    from impl import utils
    from glass_amd.factory import build_glass

    n_batches = 16
    w, ei_np, ew_np, x_np, pos_np, y_np = synth.make_workload(args.workload, seed=0, n_batches=n_batches * world)
    if args.dropout is not None:
        w.dropout = args.dropout
    ei, ew, x, pos, y = (torch.from_numpy(a) for a in (ei_np, ew_np, x_np, pos_np, y_np))
    nnz, N, H, L = ei.shape[1], w.n_node, w.hidden, w.layers

    torch.manual_seed(0)
    model = build_glass(H, L, int(x.max()), w.n_class, w.aggr, w.pool, w.z_ratio, dropout=w.dropout).to(dev)
    model.train()
    from glass_amd.arena import ParamArena
    from glass_amd.optim import FlatAdam
    bucket = ParamArena(model)  # flat params + grads: stacked weight views, fused Adam, one all-reduce
    opt = FlatAdam(bucket, lr=w.lr)
    loss_fn = loss_fn_for(w)
    xg, eig, ewg = x.to(dev), ei.to(dev), ew.to(dev)
    # this rank's batches: rank r owns batches r, r+world, ...  (disjoint subgraphs; weak scaling)
    pos_g = pos.to(dev).reshape(n_batches * world, w.batch, -1)[rank::world].contiguous()
    y_g = y.to(dev).reshape(n_batches * world, w.batch, *y.shape[1:])[rank::world].contiguous()
    ops.rng_seed(1234 + rank, dev)

    from glass_amd.step import TrainStep
    stepper = TrainStep(model, opt, loss_fn, xg, eig, ewg, bucket, use_graph=bool(args.graph))

    def run(k, offset):
        for i in range(k):
            b = (offset + i) % n_batches
            stepper(pos_g[b], y_g[b])

    def barrier():
        if world > 1:
            import torch.distributed as td
            td.barrier(device_ids=[local_rank]) if backend == "nccl" else td.barrier()
        torch.cuda.synchronize()

    run(args.warmup, 0)
    barrier()
    t0 = time.perf_counter()
    run(args.steps, args.warmup)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as td
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        td.all_reduce(t, op=td.ReduceOp.MAX)
        dt = t.item()
    last_loss = stepper.last_loss()

    # ---- roofline of the dominant kernel (K1), measured in the same step loop with HIP events ----
    # A second, instrumented pass of the same steps: every K1 launch is bracketed by HIP events recorded
    # on the launch stream (graph.K1_EVENT_HOOK).  The pass is eager (events cannot sit inside the replayed
    # graph), and an eager step is host-bound here (~2.5 ms of launches for ~0.6 ms of GPU work), which would
    # count host starvation between the two records as kernel time; so each step is queued behind a GPU-side
    # spin long enough for the host to run ahead — the commands then execute back to back, as in the graph.
    k1_steps = min(args.steps, 50)
    events = []
    eager = TrainStep(model, opt, loss_fn, xg, eig, ewg, bucket, use_graph=False)
    eager(pos_g[0], y_g[0])
    torch.cuda.synchronize()
    t_host = time.perf_counter()
    eager(pos_g[0], y_g[0])  # host time of one eager step (no sync inside)
    t_host = time.perf_counter() - t_host
    torch.cuda.synchronize()
    spin_cycles = int(max(t_host * 1.5, 2e-3) * 2.4e9)
    ggraph.K1_EVENT_HOOK = events
    null_pairs = []  # two back-to-back records with nothing between them: the bracket's own cost
    for i in range(k1_steps):
        torch.cuda._sleep(spin_cycles)
        eager(pos_g[i % n_batches], y_g[i % n_batches])
        n0, n1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n0.record()
        n1.record()
        null_pairs.append((n0, n1))
        torch.cuda.synchronize()
    ggraph.K1_EVENT_HOOK = None
    torch.cuda.synchronize()
    bracket_ms = sorted(a.elapsed_time(b) for a, b in null_pairs)[len(null_pairs) // 2]
    # adjacency launches only (the embedding backward also runs on K1, with its own tiny matrix)
    k1_ms = [a.elapsed_time(b) for a, b, _nr, nz, _h in events if nz == nnz]
    k1_raw = sum(k1_ms) / len(k1_ms) * 1e-3
    k1_avg = max(k1_raw - bracket_ms * 1e-3, 1e-7)  # launch duration = bracket reading - empty-bracket reading
    alg_bytes = nnz * (4 * H + 8) + N * (4 * H + 4)
    roofline = {"bound": "hbm", "achieved": alg_bytes / k1_avg / 1e9, "peak": 8000.0, "unit": "GB/s",
                "frac": alg_bytes / k1_avg / 8e12, "traffic": None, "kernel": "glass_spmm_csr_f32 (spmm_sweep_kernel)",
                "alg_bytes_per_launch": alg_bytes, "avg_launch_us": k1_avg * 1e6, "launches_timed": len(k1_ms),
                "event_bracket_raw_us": k1_raw * 1e6, "empty_bracket_us": bracket_ms * 1e3}
    prof = os.path.join(ROOT, "profiles", "r01_k1_traffic.json")
    if os.path.exists(prof):
        try:
            with open(prof) as f:
                roofline["traffic"] = json.load(f).get(args.workload, {}).get("hbm_bytes_per_launch")
        except Exception:
            pass
    if roofline["traffic"]:
        # frac above counts ALGORITHMIC bytes (X is L2 / Infinity-Cache resident on the small BASELINE graphs, so it can
        # exceed 1); this is the same launch priced with the L2-miss traffic the PMC counters saw
        roofline["traffic_frac"] = roofline["traffic"] / k1_avg / 8e12
        roofline["traffic_source"] = "profiles/r01_k1_traffic.json (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, separate passes)"

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(w, ei, ew, x, pos, y, args.cpu_steps)

    if rank == 0:
        out = {
            "metric": "aggregated edges/sec (GLASSConv fwd+bwd)", "value": nnz * L * args.steps * world / dt,
            "unit": "edges/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{w.name}-shaped synthetic graph (BASELINE config[1] family): N={N}, nnz={nnz}, "
                                   f"hidden={H}, layers={L}, aggr={w.aggr}, pool={w.pool}, z_ratio={w.z_ratio}, "
                                   f"dropout={w.dropout}, batch={w.batch}x{w.sub_size} per rank, use_deg features, Adam",
                       "parallelism": f"subgraph-batch dp{world}, replicated graph, one flat all-reduce/step",
                       "step": "MaxZOZ+fwd+loss+bwd+allreduce+Adam", "hip_graph": bool(stepper.graphed),
                       "final_loss": last_loss},
            "roofline": roofline, "cpu_baseline": cpu,
        }
        print(json.dumps(out), flush=True)
    if world > 1:
        import torch.distributed as td
        td.destroy_process_group()


if __name__ == "__main__":
    main()
