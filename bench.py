"""Headline benchmark: aggregated edges/sec of the GLASS labeled message-passing step.

    python bench.py --gpus N --steps K --warmup W

With --gpus N > 1 and no WORLD_SIZE in the environment, this process starts N ranks itself
(`python -m torch.distributed.run --nproc-per-node N bench.py ...` as a child, before anything here
touches a GPU), forwards rank 0's JSON line and exits with the children's return code.  Launched by
torch.distributed.run directly (RANK / LOCAL_RANK / WORLD_SIZE set) it is one rank of that job.

A step = MaxZOZ + GLASS.forward + loss + backward + (gradient all-reduce) + Adam.step on one batch
of subgraphs per rank, over the WHOLE graph (the reference propagates over the full graph every
step, /root/reference/impl/train.py:10-16).  metric = nnz * L * steps * world / wall_seconds
(SURVEY.md §8d), inputs resident in HBM before the timed region.  At N=1 the workload is BASELINE
config[1]: the ppi_bp-shaped synthetic graph, hidden=64 (config/ppi_bp.yml hyper-parameters incl.
dropout 0.5).  Multi-GPU = subgraph-batch data parallelism: replicated graph, per-rank batch fixed
("weak"), gradient all-reduce per step (--features nodeid: bucketed, see glass_amd/dist.py).

Timing: after W warm-up steps, BLOCKS of exactly K steps are timed, each bracketed by a barrier +
torch.cuda.synchronize() on both sides and taken as the MAX over ranks; at least 25 blocks and at least
0.25 s in all (a K = 20 block of this step is 6 ms: one block alone is at the mercy of a single host
hiccup).  ms_per_step / value come from the MEDIAN block; min / max / blocks are printed beside it.

The JSON line also carries
  roofline       : the CSR aggregation kernel (K1) inside the step — algorithmic bytes
                   nnz*(4H+8)+N*(4H+4) per launch / its average duration.  Durations come from HIP events
                   recorded on the launch stream around every K1 launch of an instrumented pass, minus the
                   bracket's own cost, which is CALIBRATED in the same process on the same kernel and shape
                   (bracketed average - back-to-back average of a 50-launch hipGraph).  Two utilisations, both
                   <= 1 by construction: every gathered byte passes an XCD L2 (algorithmic rate / 34.5 TB/s
                   aggregate), and the L2-miss traffic passes the memory side — the Infinity Cache while X fits
                   its 256 MiB (the guide's gathered-row rate, 8.6 TB/s chip-wide), HBM (8 TB/s) beyond.
                   A third one prices what binds a row gather: the gathered neighbour rows (nnz * 4H bytes) against
                   the guide's measured whole-row gather rates (18.8 TB/s from L2, 8.6 TB/s from the Infinity Cache),
                   blended by the measured share of rows that missed L2.  `frac` = the largest, `bound` names it.  `traffic` = memory-side bytes per launch from
                   PMC counters collected IN THIS RUN (a child `rocprofv3 --kernel-trace --pmc FETCH_SIZE`,
                   then WRITE_SIZE, on tools/bin/spmm_bench at the same shape: separate passes, KiB units,
                   FETCH_SIZE doubled per the gfx950 rule) or null.
  roofline_hbm   : K1 alone, same process, on shapes whose X cannot be cache-resident (N = 4 M permutation
                   = no reuse at all; N = 2 M uniform, mean degree 6), hidden 64, against the 8 TB/s HBM peak.
  step_breakdown : device time per C-ABI entry point per step (same bracket method); the dense (MFMA)
                   calls with the FLOPs they EXECUTE (effective-weight kernels run one product where the
                   reference formulation has two) next to the reference formulation's.
  cpu_baseline   : the oracle (CPU restatement of the reference path, torch ops on host cores) timed
                   on a bounded sample of the same workload on rank 0 at N=1.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

L2_PEAK_GBPS = 34500.0   # MI355X_MICROARCH.md §L2: aggregate of the 8 XCD L2s
HBM_PEAK_GBPS = 8000.0   # spec (6.29 TB/s measured achievable copy)
HBM_COPY_GBPS = 6290.0   # MI355X_MICROARCH.md §HBM: what a float4 copy achieves; an "HBM" rate above it was cache-assisted
IC_GATHER_GBPS = 8600.0  # MI355X_MICROARCH.md §Indexed rows: uniformly random rows of a 38 MB table (Infinity Cache), chip-wide
L2_GATHER_GBPS = 18800.0  # same table: rows of a table shared by every workgroup and served by the XCD's L2 (16.8-18.8 TB/s; the upper end)
MFMA_F32_TFLOPS = 157.3  # dense fp32 matrix-core peak (f32-input MFMA)
MFMA_BF16_TFLOPS = 2500.0  # dense bf16 matrix-core peak (MI355X_MICROARCH.md); the split product form issues 6 bf16 MFMAs per fp32 product
L2_XCD_BYTES = 4 << 20
IC_BYTES = 256 << 20


# which BASELINE.json `configs` entry a workload is
BASELINE_CONFIG = {"density": "BASELINE config[0]: the reference's own CPU-runnable case, here on the GPU", "ppi_bp": "BASELINE config[1]",
                   "hpo_neuro": "BASELINE config[2]", "em_user": "BASELINE config[3], one rank's share", "powerlaw": "BASELINE config[4], one rank's share"}


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="ppi_bp", help="ppi_bp | hpo_neuro | em_user | powerlaw | density (the shipped C1 graph) | tiny")
    ap.add_argument("--features", default="deg", choices=["deg", "nodeid"],
                    help="deg: use_deg-style small table (default); nodeid: V = N embedding table (dense [N,H] "
                         "gradient -> the bucketed collective matters)")
    ap.add_argument("--dropout", type=float, default=None, help="override the workload's YAML dropout")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline-hbm", action="store_true", help="skip the stand-alone HBM-bound K1 measurements")
    ap.add_argument("--no-pmc", action="store_true",
                    help="skip the child rocprofv3 --pmc passes that measure K1's memory-side traffic (roofline.traffic = null); "
                         "use it when this command itself runs under rocprofv3")
    ap.add_argument("--min-blocks", type=int, default=25, help="timed blocks of --steps steps each (median reported)")
    ap.add_argument("--cpu-steps", type=int, default=0, help="CPU baseline steps (0 = size to ~15 s)")
    ap.add_argument("--graph", type=int, default=1, help="1: replay the step from a captured hipGraph when possible")
    ap.add_argument("--no-floor", action="store_true", help="skip the launch-chain floor of the step (a child rocprofv3 --kernel-trace pass)")
    ap.add_argument("--mode", default="train", choices=["train", "eval", "trace"],
                    help="eval: the evaluation loop's forward passes (impl/train.py:20-34) over the same batches — one batch per "
                         "step, K batches side by side as parallel branches of one hipGraph (glass_amd/evalstep.py)")
    ap.add_argument("--eval-parallel", type=int, default=8, help="--mode eval: batches per replay (the sequential form is timed too)")
    ap.add_argument("--head-labels", type=int, default=1,
                    help="1: the label launch of the step's batch rides in the graph's head launch (prologue || labels, "
                         "glass_step_head_f32, batch named by a device-resident cursor); 0: eager label launch in front of every replay")
    ap.add_argument("--caller", default="step", choices=["step", "reference"],
                    help="step: bench.py drives glass_amd.step.TrainStep itself (flat arena + FlatAdam built here).  reference: "
                         "the step is reached the way /root/reference/GLASSTest.py reaches it — buildModel's constructions, "
                         "torch.optim.Adam(model.parameters()), the driver's own loss callable, ZGDataloader(z_fn=MaxZOZ, "
                         "shuffle, drop_last) — through impl.train.train, one epoch of exactly --steps batches per timed block")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher / process-group plumbing only (no GPU work; value is null) — CPU-box smoke")
    return ap.parse_args(argv)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(args):
    """--gpus N without a torchrun environment: start the N ranks as a child job.  Nothing in this process has
    touched the GPU yet (no torch.cuda call, libglass_hip not loaded); the child processes are fresh interpreters."""
    if os.environ.get("GLASS_BENCH_BACKEND", "nccl") == "nccl" and args.gpus > torch.cuda.device_count() and not args.dry_run:
        # (device_count() does not initialise the GPU) — refuse before any rank can hang in init_process_group
        print(f"bench.py: --gpus {args.gpus} but this node has {torch.cuda.device_count()} GPU(s); RCCL needs one GPU per "
              "rank (GLASS_BENCH_BACKEND=gloo shares one GPU between ranks for a plumbing smoke test)", file=sys.stderr)
        return 2
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    return subprocess.run(cmd, env=env).returncode


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


# ---- device-time instrumentation -----------------------------------------------------------------------------
class _BracketLib:
    """Stand-in for the ctypes library during an instrumented pass: every C-ABI call that takes a stream is bracketed
    by two HIP events recorded on the launch stream (torch's current stream is the stream every launch here uses)."""
    def __init__(self, lib, sink):
        self._lib, self._sink = lib, sink

    def __getattr__(self, name):
        fn = getattr(self._lib, name)
        if not name.endswith("_f32") and name not in ("glass_maxzoz_i64", "glass_copy_pair", "glass_rng_advance",
                                                      "glass_batch_labels"):
            return fn
        sink = self._sink

        def call(*a):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = fn(*a)
            e1.record()
            sink.append((name, e0, e1, a))
            return rc
        return call


def _k1_alg_bytes(nnz, n_rows, H):
    return nnz * (4 * H + 8) + n_rows * (4 * H + 4)


def k1_back_to_back(op, H, launches=50, replays=5):
    """Average duration of one K1 launch when `launches` of them run back to back from a hipGraph (no host gaps,
    no event between launches): what rocprofv3's kernel trace reports per dispatch, up to the launch boundary."""
    dev = op.rowptr.device
    x = torch.randn(op.n_cols, H, device=dev)
    y = torch.empty(op.n_rows, H, device=dev)
    for _ in range(3):
        op.spmm(x, out=y)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    import torch.distributed as td
    # thread_local: a process group's watchdog thread may query events while this thread captures (glass_amd/step.py)
    mode = "thread_local" if (td.is_available() and td.is_initialized()) else "global"
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side, capture_error_mode=mode):
            for _ in range(launches):
                op.spmm(x, out=y)
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    # one event pair per replay, each replay queued behind a GPU-side spin so that the host has submitted it before the GPU
    # gets there (a host thread delayed between two replays — e.g. by the CPU baseline's worker threads still spinning — once
    # put 2.3x into a single-region average of this function); the median replay is reported
    times = []
    for _ in range(max(replays, 5)):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda._sleep(int(1e-3 * 2.4e9))
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        times.append(e0.elapsed_time(e1) * 1e-3 / launches)
    times.sort()
    return times[len(times) // 2], (x, y)


def k1_bracketed(op, x, y, launches=50):
    """The same launch measured the way the in-step launches are: one event pair around each, the stream kept fed
    (each batch queued behind a GPU-side spin so that the host runs ahead)."""
    evs = []
    torch.cuda._sleep(int(3e-3 * 2.4e9))
    for _ in range(launches):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        op.spmm(x, out=y)
        e1.record()
        evs.append((e0, e1))
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return ts[len(ts) // 2] * 1e-3


def roofline_hbm_entries(dev):
    """K1 alone on shapes where X (and Y) cannot be cache-resident: the regime the >= 0.60 target is about."""
    import numpy as np
    from glass_amd import graph as ggraph, synth
    out = []
    H = 64
    # (a) permutation matrix, N = 4 M: every X row read exactly once, no reuse anywhere (X = Y = 1.02 GB)
    n = 4_000_000
    rng = np.random.Generator(np.random.PCG64(7))
    col = torch.from_numpy(rng.permutation(n).astype(np.int32)).to(dev)
    rowptr = torch.arange(n + 1, dtype=torch.int32, device=dev)
    op = ggraph.CSROperand(rowptr, col, torch.ones(n, device=dev), n, n)
    t, keep = k1_back_to_back(op, H, launches=10, replays=3)
    b = _k1_alg_bytes(n, n, H)
    out.append({"shape": "permutation N=4000000 (degree 1, no reuse)", "H": H, "bound": "hbm", "avg_launch_us": t * 1e6,
                "alg_bytes_per_launch": b, "achieved": b / t / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": b / t / 1e9 / HBM_PEAK_GBPS, "cache_assisted": False, "evidence_for_hbm_target": True,
                "label": "HBM-bound, no reuse: every X row is fetched exactly once and X (1 GB) is 4x the Infinity Cache — THE "
                         "figure the >= 0.60-of-HBM-roofline target is answered with"})
    del op, keep, col, rowptr
    # (b) uniform random graph, N = 2 M, 6 M undirected pairs (mean degree 6): X = 512 MB > Infinity Cache
    n, pairs = 2_000_000, 6_000_000
    rng = np.random.Generator(np.random.PCG64(11))
    u, v = rng.integers(0, n, pairs), rng.integers(0, n, pairs)   # duplicates / self-loops are harmless for a timing shape
    row = torch.from_numpy(np.concatenate([u, v])).to(dev)
    colt = torch.from_numpy(np.concatenate([v, u])).to(dev)
    order = torch.argsort(row * n + colt)
    row, colt = row[order], colt[order]
    rowptr = ggraph._csr_from_sorted(row, n)
    op = ggraph.CSROperand(rowptr, colt.to(torch.int32).contiguous(),
                           torch.full((2 * pairs, ), 1.0 / 6.0, device=dev), n, n)
    t, keep = k1_back_to_back(op, H, launches=10, replays=3)
    b = _k1_alg_bytes(2 * pairs, n, H)
    rate = b / t / 1e9
    out.append({"shape": "uniform N=2000000 nnz=12000000 (mean degree 6)", "H": H, "bound": "hbm", "avg_launch_us": t * 1e6,
                "alg_bytes_per_launch": b, "achieved": rate, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": rate / HBM_PEAK_GBPS, "cache_assisted": rate > HBM_COPY_GBPS, "evidence_for_hbm_target": False,
                "label": "memory-side (HBM + Infinity Cache): X is 512 MB, half of it fits the 256 MiB Infinity Cache and each row "
                         "is gathered 6 times, so part of the algorithmic bytes never reaches HBM — a rate above the 6.3 TB/s an HBM "
                         "copy achieves is cache-assisted, not HBM evidence"})
    del op, keep
    torch.cuda.empty_cache()
    return out


_PROFILER_ENV_PREFIXES = ("ROCP_", "ROCPROF", "ROCPROFILER_", "ROCTRACER_", "HSA_TOOLS_")


def under_profiler(env=None):
    """True when this process was started by a profiler (rocprofv3 preloads its tool library and exports ROCP_* /
    ROCPROF* variables).  A child rocprofv3 started from here would inherit them: its launcher would initialise the GPU
    before it execs the program (the exec-after-GPU-init this pool forbids) and the outer trace would be instrumented
    twice."""
    env = os.environ if env is None else env
    if "rocprof" in env.get("LD_PRELOAD", "").lower():
        return True
    return any(k.startswith(_PROFILER_ENV_PREFIXES) for k in env)


def child_env_without_profiler(env=None):
    """The environment for a child process that must not inherit an outer profiler's hooks."""
    env = dict(os.environ if env is None else env)
    for k in list(env):
        if k.startswith(_PROFILER_ENV_PREFIXES):
            del env[k]
    if "rocprof" in env.get("LD_PRELOAD", "").lower():
        del env["LD_PRELOAD"]
    return env


def k1_pmc_traffic(workload, H, timeout=240, csr=None):
    """Memory-side bytes per K1 launch at this workload's shape, measured NOW: two child runs of tools/bin/spmm_bench under
    `rocprofv3 --kernel-trace --pmc <counter>` — FETCH_SIZE and WRITE_SIZE in separate passes, KiB units, FETCH_SIZE
    doubled (gfx950 reports half the bytes of wide loads; MI355X_MICROARCH.md §HBM).  None when the shape has no
    stand-alone generator, a tool is missing, or a pass fails — never a stale number."""
    import csv
    import glob
    import shutil
    import tempfile
    exe = os.path.join(ROOT, "tools", "bin", "spmm_bench")
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof):
        return None, "rocprofv3 missing"
    if under_profiler():
        return None, "running under a profiler (LD_PRELOAD / ROCP_* / ROCPROF_* set): the child PMC passes are skipped"
    # the stand-alone K1 program where it has a generator for the shape; otherwise (the shipped density graph) the K1 launches
    # of a few eager steps of this file itself (--mode trace --graph 0): same kernels, same matrix, counted per dispatch
    standalone = workload in ("ppi_bp", "hpo_neuro", "em_user", "powerlaw") and os.path.exists(exe)
    csr_file = None
    if not standalone and csr is not None and os.path.exists(exe):
        # a graph without a generator in the stand-alone program (the shipped density graph): its CSR as this run holds it,
        # dumped for tools/bin/spmm_bench file:<path> — same kernel, same matrix, K1 alone under the counters
        import numpy as np
        rowptr, col, val = (t.detach().cpu().numpy() for t in csr)
        fd, csr_file = tempfile.mkstemp(prefix="glass_csr_", suffix=".bin", dir="/tmp")
        with os.fdopen(fd, "wb") as f:
            np.array([rowptr.shape[0] - 1, col.shape[0]], dtype=np.int64).tofile(f)
            rowptr.astype(np.int32).tofile(f)
            col.astype(np.int32).tofile(f)
            val.astype(np.float32).tofile(f)
        workload, standalone = "file:" + csr_file, True
    child = [exe, workload, str(H), "10"] if standalone else \
        [sys.executable, os.path.join(ROOT, "bench.py"), "--mode", "trace", "--graph", "0", "--workload", workload, "--steps", "4",
         "--warmup", "1"]
    vals = {}
    d = None
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            d = tempfile.mkdtemp(prefix="glass_pmc_", dir="/tmp")
            env = child_env_without_profiler()
            env["TMPDIR"] = "/tmp"
            subprocess.run([rocprof, "--kernel-trace", "--pmc", ctr, "--output-format", "csv", "-d", d, "--", *child],
                           cwd="/tmp", env=env, timeout=timeout, stdout=subprocess.DEVNULL,
                           stderr=subprocess.DEVNULL, check=True)
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            rows = [r for r in csv.DictReader(open(files[0])) if r["Counter_Name"] == ctr and "spmm_" in r["Kernel_Name"]]
            names = sorted({r["Kernel_Name"].split("(")[0] for r in rows})
            lead = [n for n in names if "sweep" in n] or [n for n in names if "long" in n]
            launches = sum(1 for r in rows if r["Kernel_Name"].split("(")[0] == lead[0])
            vals[ctr] = sum(float(r["Counter_Value"]) for r in rows) / launches
            shutil.rmtree(d, ignore_errors=True)
            d = None
    except Exception as e:  # noqa: BLE001 — any failure means "not measured"
        return None, f"PMC pass failed: {type(e).__name__}"
    finally:
        if d is not None:
            shutil.rmtree(d, ignore_errors=True)
        if csr_file is not None and os.path.exists(csr_file):
            os.remove(csr_file)
    return int((2 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024), (
        "this run: child rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes, KiB, FETCH_SIZE x2 per the "
        "gfx950 rule) on " + ("tools/bin/spmm_bench at the same shape" if standalone else
                              "the K1 launches of four eager steps of bench.py --mode trace on the same graph"))


def trailing_period(names):
    """A kernel trace that ends with identical replays of one step: the shortest period p >= 2 with the last three windows of p
    names equal (None when there is none)."""
    return next((p for p in range(2, len(names) // 3 + 1) if names[-p:] == names[-2 * p:-p] == names[-3 * p:-2 * p]), None)


def step_floor(args, n_replays=2000, timeout=300):
    """Latency floor of the step's chain of launches ON THIS BOX: (1) a child `rocprofv3 --kernel-trace` run of this file in
    --mode trace lists the kernels of one replayed step with their grid, block and dynamic-LDS sizes; (2) the same chain —
    the label launch eager, the rest captured in one hipGraph, one stream, every node depending on its predecessor — is
    replayed with a kernel that does nothing (glass_empty_launch) at the same geometries.  What that costs per step is what
    the step would cost if every kernel were free: launch latency, workgroup dispatch and the graph's node-to-node
    hand-over.  -> dict, or (None, reason)."""
    import csv
    import glob
    import shutil
    import tempfile
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof):
        return None, "rocprofv3 missing"
    if under_profiler():
        return None, "running under a profiler: the child kernel-trace pass is skipped"
    d = tempfile.mkdtemp(prefix="glass_floor_", dir="/tmp")
    try:
        env = child_env_without_profiler()
        env["TMPDIR"] = "/tmp"
        cmd = [rocprof, "--kernel-trace", "--output-format", "csv", "-d", d, "--", sys.executable, os.path.join(ROOT, "bench.py"),
               "--mode", "trace", "--workload", args.workload, "--features", args.features, "--caller", args.caller, "--steps", "6",
               "--warmup", "2", "--graph", str(args.graph)] + (["--dropout", str(args.dropout)] if args.dropout is not None else [])
        subprocess.run(cmd, cwd="/tmp", env=env, timeout=timeout, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
        files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
        rows = sorted(csv.DictReader(open(files[0])), key=lambda r: int(r["Start_Timestamp"]))
    except Exception as e:  # noqa: BLE001 — any failure means "not measured"
        return None, f"kernel-trace pass failed: {type(e).__name__}"
    finally:
        shutil.rmtree(d, ignore_errors=True)
    names = [r["Kernel_Name"].split("(")[0] for r in rows]
    period = trailing_period(names)
    if period is None:
        return None, "no periodic step found in the kernel trace"
    step_rows = rows[-period:]

    def geom(r):
        wg = [max(int(r[f"Workgroup_Size_{a}"]), 1) for a in "XYZ"]
        grid = [max(int(r[f"Grid_Size_{a}"]), 1) // w for a, w in zip("XYZ", wg)]  # (rocprofv3 reports the grid in work-items)
        return grid, wg[0] * wg[1] * wg[2], int(r.get("LDS_Block_Size", 0) or 0)
    chain = [geom(r) for r in step_rows]
    # the step starts with the label launch (eager, outside the graph): rotate the period so that it comes first
    first = next((i for i, r in enumerate(step_rows) if "batch_labels" in r["Kernel_Name"]), 0)
    chain = chain[first:] + chain[:first]
    kernel_us = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in step_rows) / 1e3
    from glass_amd import _lib
    lib = _lib.load()

    def launch(g):
        (gx, gy, gz), block, lds = g
        _lib.check(lib.glass_empty_launch(gx, gy, gz, block, min(lds, 160 * 1024), torch.cuda.current_stream().cuda_stream), "glass_empty_launch")
    eager_head = chain[:1] if first is not None and "batch_labels" in step_rows[first]["Kernel_Name"] else []
    if os.environ.get("GLASS_FLOOR_ALL_IN_GRAPH") == "1":  # what-if: the label launch captured with the rest
        eager_head = []
    body = chain[len(eager_head):]
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for g in chain:
            launch(g)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for g in body:
            launch(g)

    def one_step():
        for g in eager_head:
            launch(g)
        graph.replay()
    for _ in range(50):
        one_step()
    torch.cuda.synchronize()
    blocks = []
    per = max(n_replays // 10, 1)
    for _ in range(10):
        t0 = time.perf_counter()
        for _ in range(per):
            one_step()
        torch.cuda.synchronize()
        blocks.append((time.perf_counter() - t0) / per)
    blocks.sort()
    return {"us": blocks[len(blocks) // 2] * 1e6, "us_min": blocks[0] * 1e6, "launches": len(chain), "in_graph": len(body),
            "kernel_time_us_in_trace": kernel_us,
            "method": "the step's launch chain (child rocprofv3 --kernel-trace of --mode trace: kernel order, grids, block sizes, dynamic "
                      "LDS) replayed with glass_empty_launch at the same geometries — label launch eager, the rest one hipGraph on one "
                      "stream; median of 10 blocks"}, None


def dense_flop_model(N, H, L, pos_batches, lab_cap):
    """FLOPs per step of the fused Linear-pair kernels: the reference formulation's (two products per row of every pair,
    impl/models.py:158-173) and the ones the kernels EXECUTE.  The comb pair has no activation before the label mix, so a
    row needs one product with an effective weight: hidden 64 — every row tile runs one product, the listed labeled rows a
    second time (glass_comb_eff_*); hidden 256 / 512 forward and 128 / 256 / 512 data gradient — row tiles holding at most
    3 / 7 labeled rows run one product (dense_tiled.hip kMaxFix); weight gradient — one all-rows product plus the labeled
    rows (S / L forms).  The tile census is taken from the benchmark's own batches."""
    import numpy as np
    pair = 2.0 * N * H * (2 * H)          # one [N,H] x [H,2H] product = 4 N H^2
    ref = {"trans_fwd": pair, "comb_fwd": 2 * pair, "trans_dgrad": pair, "comb_dgrad": 2 * pair, "trans_wgrad": pair,
           "comb_wgrad": 2 * pair}
    ex = dict(ref)
    n_lab, frac_one_fwd, frac_one_dg = [], [], []
    tile = 64 if H == 128 else 128
    for pos in pos_batches:
        ids = np.unique(pos[pos >= 0])
        n_lab.append(len(ids))
        per_tile = np.bincount(ids // tile, minlength=-(-N // tile))
        frac_one_fwd.append(float((per_tile <= 3).mean()))
        frac_one_dg.append(float((per_tile <= 7).mean()))
    lab = float(np.mean(n_lab)) / N
    if H == 64 and lab_cap:
        ex["comb_fwd"] = ex["comb_dgrad"] = ex["comb_wgrad"] = pair * (1.0 + lab)
    elif H in (128, 256, 512):
        f1, d1 = float(np.mean(frac_one_fwd)), float(np.mean(frac_one_dg))
        if H >= 256:
            ex["comb_fwd"] = pair * (f1 + 2 * (1 - f1))
        ex["comb_dgrad"] = pair * (d1 + 2 * (1 - d1))
        ex["comb_wgrad"] = pair * (1.0 + min(1.0, 16 * lab))   # S + the 16-row stages (or the rows) that hold a label
    return {k: v * L for k, v in ref.items()}, {k: v * L for k, v in ex.items()}


def shared_workload(name, n_batches, world, rank):
    """The synthetic workload, made BEFORE init_process_group: rank 0 generates it once (config 5: a 20 M-edge rejection
    sampler, ~a minute) and publishes it atomically as a temp .npz; the other ranks wait for the FILE, not inside a
    collective, so no process-group timeout is running meanwhile, and a rank-0 failure shows up as their own timeout with
    a message.  Returns (workload, arrays, share path or None); rank 0 removes the file (`finally`) once a
    barrier has told it that every rank holds the data."""
    import numpy as np
    from glass_amd import synth
    if world == 1:
        w, *arrs = synth.make_workload(name, seed=0, n_batches=n_batches * world)
        return w, tuple(arrs), None
    # one file per LAUNCH: the ranks of one launch share their parent (the torch.distributed.run agent, or bench.py's own
    # spawner), so its pid is a nonce a SIGKILLed earlier run with the same port cannot have left behind; rank 0 also
    # removes whatever sits at the path before it generates, and the real run checks the ranks' CRCs agree (main())
    share = f"/tmp/glass_bench_{os.environ.get('MASTER_PORT', '0')}_{os.getppid()}_{name}_{world}x{n_batches}.npz"
    if rank == 0:
        for stale in (share, share + ".tmp.npz"):
            if os.path.exists(stale):
                os.remove(stale)
        w, *arrs = synth.make_workload(name, seed=0, n_batches=n_batches * world)
        np.savez(share + ".tmp.npz", **dict(zip(("ei", "ew", "x", "pos", "y"), arrs)))
        os.replace(share + ".tmp.npz", share)
        return w, tuple(arrs), share
    w = synth.WORKLOADS[name]
    deadline = time.time() + float(os.environ.get("GLASS_BENCH_WORKLOAD_TIMEOUT", "1800"))
    while not os.path.exists(share):
        if time.time() > deadline:
            raise SystemExit(f"bench.py rank {rank}: rank 0 did not publish the workload ({share}) in time")
        time.sleep(0.2)
    with np.load(share) as z:
        arrs = tuple(z[k] for k in ("ei", "ew", "x", "pos", "y"))
    return w, arrs, share


def dry_run(args, world, rank):
    """Process-group plumbing without a GPU (gloo): the workload hand-over (rank 0 generates, the others load the file before
    any collective), the same barrier / max-over-ranks / rank-0-prints protocol, and the `collective` block every N > 1 line
    carries.  The CPU suite runs it with --gpus 8 (tests/test_bench_launcher.py)."""
    import zlib
    import torch.distributed as td
    name = args.workload if args.workload in ("tiny", "density", "ppi_bp") else "tiny"  # (no minute-long samplers in a dry run)
    w, arrs, share = shared_workload(name, 2, world, rank)
    try:
        if world > 1:
            td.init_process_group("gloo")
            td.barrier()
    finally:
        if share is not None and rank == 0 and os.path.exists(share):
            os.remove(share)
    crc = float(zlib.crc32(arrs[0].tobytes()) ^ zlib.crc32(arrs[3].tobytes()))
    same = True
    if world > 1:
        lo, hi = torch.tensor([crc], dtype=torch.float64), torch.tensor([crc], dtype=torch.float64)
        td.all_reduce(lo, op=td.ReduceOp.MIN)
        td.all_reduce(hi, op=td.ReduceOp.MAX)
        same = bool(lo.item() == hi.item())
    # this rank's batches exactly as main() slices them: rank r owns batches r, r + world, ...
    pos = arrs[3].reshape(2 * world, w.batch, -1)[rank::world]
    t0 = time.perf_counter()
    for _ in range(args.steps):
        pass
    if world > 1:
        td.barrier()
    dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    if world > 1:
        td.all_reduce(dt, op=td.ReduceOp.MAX)
    if rank == 0:
        print(json.dumps({"metric": "aggregated edges/sec (GLASSConv fwd+bwd)", "value": None, "unit": "edges/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": None,
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
                          "data": "synthetic", "dry_run": True,
                          "config": {"workload": name, "parallelism": f"subgraph-batch dp{world}",
                                     "batches_per_rank": int(pos.shape[0]), "workload_identical_on_all_ranks": same},
                          "collective": {"world_size": td.get_world_size() if world > 1 else 1, "backend": "gloo" if world > 1 else None,
                                         "rccl_version": None, "in_graph": False, "capture_error": None, "exposed_us": None,
                                         "payload_bytes": None}}), flush=True)
    if world > 1:
        td.destroy_process_group()
    if not same:
        sys.exit(3)


def eval_mode(args, w, model, xg, eig, ewg, pos_g, nnz, N, H, L, world, rank, local_rank, backend, dev):
    """--mode eval: forward-only passes over the batches (train.test's loop: utils.MaxZOZ + GLASS.forward per batch, no
    dropout), a "step" = one batch.  Timed twice with the same block protocol as the training line: K = 1 (one batch per
    hipGraph replay) and K = --eval-parallel batches as parallel branches of one graph; `value` is the parallel form's."""
    from glass_amd.evalstep import EvalGraph
    model.eval()
    n_batches = pos_g.shape[0]

    def barrier():
        if world > 1:
            import torch.distributed as td
            td.barrier(device_ids=[local_rank]) if backend == "nccl" else td.barrier()
        torch.cuda.synchronize()

    results = {}
    with torch.no_grad():
        from glass_amd.utils import MaxZOZ
        kmax = max(1, args.eval_parallel)
        refs = [model(xg, eig, ewg, pos_g[b % n_batches], MaxZOZ(xg, pos_g[b % n_batches])).clone() for b in range(kmax)]
        for k in sorted({1, kmax}):
            g = EvalGraph(model, xg, eig, ewg, pos_g[0].shape, k).capture()
            # EVERY branch on a real batch of its own (branches sharing scratch would show here): each replayed branch = the
            # eager forward of its batch, bit for bit
            outs = g([pos_g[b % n_batches] for b in range(k)])
            same = all(bool(torch.equal(o, refs[b])) for b, o in enumerate(outs))

            def run(steps, offset):
                for i in range(0, steps, k):
                    g([pos_g[(offset + i + j) % n_batches] for j in range(min(k, steps - i))])
            run(args.warmup, 0)
            blocks = []
            for b in range(max(args.min_blocks, 5)):
                barrier()
                t0 = time.perf_counter()
                run(args.steps, b * args.steps)
                barrier()
                blocks.append(time.perf_counter() - t0)
            blocks.sort()
            results[k] = {"ms_per_batch": blocks[len(blocks) // 2] / args.steps * 1e3, "ms_min": blocks[0] / args.steps * 1e3,
                          "ms_max": blocks[-1] / args.steps * 1e3, "matches_eager_bitwise": same}
    kk = max(results)
    dt_ms = results[kk]["ms_per_batch"]
    if rank == 0:
        print(json.dumps({
            "metric": "aggregated edges/sec (GLASSConv fwd, evaluation loop)", "value": nnz * L * world / (dt_ms * 1e-3),
            "unit": "edges/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt_ms,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "mode": "eval",
            "config": {"workload": f"{w.name}-shaped synthetic graph: N={N}, nnz={nnz}, hidden={H}, layers={L}, aggr={w.aggr}, "
                                   f"pool={w.pool}, batch={w.batch}x{w.sub_size}, use_deg features, model.eval()",
                       "step": "MaxZOZ + GLASS.forward of ONE batch (impl/train.py:20-34); K batches per hipGraph replay as "
                               "parallel branches", "parallel_batches": kk, "hip_graph": True},
            "sequential": results[1], "parallel": results[kk],
            "speedup_parallel_over_sequential": results[1]["ms_per_batch"] / dt_ms}), flush=True)
    if world > 1:
        import torch.distributed as td
        td.destroy_process_group()


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.dry_run:
        return dry_run(args, world, rank)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path is the only product path)")
    n_dev = torch.cuda.device_count()
    backend = os.environ.get("GLASS_BENCH_BACKEND", "nccl")  # "gloo": smoke-test the N>1 path on a 1-GPU box
    if backend == "nccl" and world > n_dev:
        if rank == 0:
            print(f"bench.py: world size {world} but this node has {n_dev} GPU(s); RCCL needs one GPU per rank",
                  file=sys.stderr)
        sys.exit(2)  # every rank leaves before init_process_group: nobody waits for a rank that cannot start
    if backend == "gloo":
        local_rank = local_rank % n_dev
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from glass_amd import synth, ops, graph as ggraph, _lib
    from glass_amd.factory import build_glass

    ref_caller = args.caller == "reference"
    n_batches = args.steps if ref_caller else 16  # (reference caller: one epoch of the loader = one timed block)
    w, (ei_np, ew_np, x_np, pos_np, y_np), share = shared_workload(args.workload, n_batches, world, rank)
    try:
        if world > 1:
            import torch.distributed as td
            if backend == "nccl":
                td.init_process_group("nccl", device_id=dev)  # RCCL over xGMI
            else:
                td.init_process_group(backend)
            td.barrier(device_ids=[local_rank]) if backend == "nccl" else td.barrier()  # every rank holds the workload
    finally:
        if share is not None and rank == 0 and os.path.exists(share):
            os.remove(share)
    if world > 1:
        # every rank must hold the SAME workload (a stale or half-written share file would otherwise surface as a late
        # reshape error, or as ranks training on different graphs): CRC of the edge list and the subgraphs, MIN == MAX
        import zlib
        crc = float(zlib.crc32(ei_np.tobytes()) ^ zlib.crc32(pos_np.tobytes()))
        lo, hi = (torch.tensor([crc], dtype=torch.float64, device=dev if backend == "nccl" else "cpu") for _ in range(2))
        td.all_reduce(lo, op=td.ReduceOp.MIN)
        td.all_reduce(hi, op=td.ReduceOp.MAX)
        if lo.item() != hi.item():
            raise SystemExit(f"bench.py rank {rank}: the ranks hold different workloads (CRC {lo.item():.0f} .. {hi.item():.0f})")
    if args.dropout is not None:
        w.dropout = args.dropout
    if args.features == "nodeid":
        import numpy as np
        x_np = np.arange(w.n_node, dtype=np.int64).reshape(-1, 1, 1)  # use_nodeid: identity gather, V = N
    ei, ew, x, pos, y = (torch.from_numpy(a) for a in (ei_np, ew_np, x_np, pos_np, y_np))
    nnz, N, H, L = ei.shape[1], w.n_node, w.hidden, w.layers

    torch.manual_seed(0)
    model = build_glass(H, L, int(x.max()), w.n_class, w.aggr, w.pool, w.z_ratio, dropout=w.dropout).to(dev)
    model.train()
    from glass_amd.arena import ParamArena
    from glass_amd.optim import FlatAdam
    if ref_caller:
        # exactly the reference driver's objects (GLASSTest.py:57-58 / 69, 213): nothing of glass_amd is named here
        torch_opt = torch.optim.Adam(model.parameters(), lr=w.lr)
        loss_fn = (lambda p, t: nn.BCEWithLogitsLoss()(p.flatten(), t.flatten())) if w.multilabel else nn.CrossEntropyLoss()
        bucket = opt = None  # what impl.train.train builds underneath is picked up after the first epoch
    else:
        bucket = ParamArena(model)  # flat params + grads: stacked weight views, fused Adam, bucketed all-reduce
        opt = FlatAdam(bucket, lr=w.lr)
        loss_fn = loss_fn_for(w)
    xg, eig, ewg = x.to(dev), ei.to(dev), ew.to(dev)
    # this rank's batches: rank r owns batches r, r+world, ...  (disjoint subgraphs; weak scaling)
    pos_g = pos.to(dev).reshape(n_batches * world, w.batch, -1)[rank::world].contiguous()
    y_g = y.to(dev).reshape(n_batches * world, w.batch, *y.shape[1:])[rank::world].contiguous()
    ops.rng_seed(1234 + rank, dev)

    if args.mode == "eval":
        return eval_mode(args, w, model, xg, eig, ewg, pos_g, nnz, N, H, L, world, rank, local_rank, backend, dev)

    from glass_amd.step import TrainStep
    if ref_caller:
        from impl import SubGDataset as rSub, train as rtrain, utils as rutils
        ds = rSub.GDataset(xg, eig, ewg, pos_g.reshape(-1, pos_g.shape[-1]), y_g.reshape(-1, *y_g.shape[2:]))
        loader = rSub.ZGDataloader(ds, w.batch, z_fn=rutils.MaxZOZ, shuffle=True, drop_last=True)  # GLASSTest.py:107-113
        assert len(loader) == args.steps
        rtrain.train(torch_opt, model, loader, loss_fn)  # first epoch: adoption, warm-up, capture
        steps_built = model.__dict__.get("_glass_train_steps") or {}
        if len(steps_built) != 1:
            raise SystemExit("bench.py --caller reference: impl.train.train did not take the TrainStep path")
        stepper = next(iter(steps_built.values()))
        opt, bucket = stepper.opt, stepper.bucket
        if not (stepper._program_step() and (stepper.graphed or not args.graph)):
            raise SystemExit("bench.py --caller reference: the reference caller is not on the captured step program")

        def run(k, offset):
            for _ in range(-(-k // args.steps)):  # whole epochs (the warm-up count is rounded up to one)
                rtrain.train(torch_opt, model, loader, loss_fn)
    else:
        stepper = TrainStep(model, opt, loss_fn, xg, eig, ewg, bucket, use_graph=bool(args.graph))
        # the pre-selected batches as one "data set" + index batches: the step's head launch then labels its batch itself
        # through a device-resident cursor that cycles over them (TrainStep.begin_epoch: prologue || labels as ONE launch,
        # nothing in front of the replay) — what impl.train.train does per epoch with the loader's permutation
        head = bool(args.graph) and args.head_labels and stepper.begin_epoch(
            pos_g.reshape(-1, pos_g.shape[-1]), y_g.reshape(-1, *y_g.shape[2:]),
            torch.arange(n_batches * w.batch, device=dev).reshape(n_batches, w.batch), wrap=True)

        def run(k, offset):
            if head:
                for _ in range(k):
                    stepper.next_step()
                return
            for i in range(k):
                b = (offset + i) % n_batches
                stepper(pos_g[b], y_g[b])
    stepper.time_collective = world > 1  # HIP events around the exchange / optimizer part of every step
    if args.mode == "trace":
        # child of step_floor(): a few replays of the step under rocprofv3 --kernel-trace, nothing else
        run(max(args.warmup, 1), 0)
        run(args.steps, 0)
        torch.cuda.synchronize()
        return

    def barrier():
        if world > 1:
            import torch.distributed as td
            td.barrier(device_ids=[local_rank]) if backend == "nccl" else td.barrier()
        torch.cuda.synchronize()

    def timed_block(offset):
        barrier()
        t0 = time.perf_counter()
        run(args.steps, offset)
        barrier()
        return time.perf_counter() - t0

    def max_over_ranks(values):
        if world == 1:
            return list(values)
        import torch.distributed as td
        t = torch.tensor(values, device=dev, dtype=torch.float64)
        td.all_reduce(t, op=td.ReduceOp.MAX)
        return t.tolist()

    run(args.warmup, 0)
    # blocks of EXACTLY --steps steps, each bracketed by barrier + synchronize; the first one sizes the run (same count on
    # every rank: it is derived from the max over ranks)
    first = max_over_ranks([timed_block(args.warmup)])[0]
    # at least --min-blocks blocks and about THREE seconds of timed region (the r04 driver run rested on 0.25 s of GPU time, the
    # r05 one on 1 s: both too short for its 1 Hz utilisation samples to see the GPU phase), at most 1 200 blocks
    n_blocks = int(min(max(args.min_blocks, -(-3.0 // max(first, 1e-6))), 1200))
    times = [first] + [timed_block(args.warmup + (1 + b) * args.steps) for b in range(n_blocks - 1)]
    times = sorted(max_over_ranks(times))
    dt = times[len(times) // 2]  # the median block
    last_loss = stepper.last_loss()
    collective = stepper.collective_share() if world > 1 else None
    if collective is not None:
        # exposed share of the exchange = step time with it - step time without it (same blocks protocol; the ranks'
        # parameters diverge from here on, which only the timing below and the instrumented pass see)
        if stepper.graphed and not stepper.collective_in_graph:
            stepper.exchange_enabled = False
            off = args.warmup + n_blocks * args.steps
            t_wo = sorted(max_over_ranks([timed_block(off + b * args.steps) for b in range(max(5, n_blocks // 3))]))
            stepper.exchange_enabled = True
            collective["exposed_us"] = (dt - t_wo[len(t_wo) // 2]) / args.steps * 1e6
            collective["exposed_method"] = "median block with the exchange - median block with the collectives skipped"
        else:
            collective["exposed_us"] = None
            collective["exposed_method"] = "the exchange is captured inside the step's graph: no separate timing"
        # what the model expects of this line: the exchange sits between the last gradient and Adam, so unless it overlaps the
        # backward tail (embedding-sized bucket only) all of it is exposed — predicted weak-scaling efficiency of the step
        ms = dt / args.steps * 1e3
        pred = collective.get("predicted_us")
        if pred is not None:
            hidden = collective["predicted"]["small_allreduce_us"] if collective.get("small_bucket_overlaps_backward_tail") else 0.0
            collective["predicted_exposed_us"] = pred - hidden
            collective["predicted_share_of_step"] = (pred - hidden) * 1e-3 / ms
            collective["predicted_note"] = ("ms_per_step includes the exchange; a 1-GPU step of the same workload + predicted_exposed_us "
                                            "is what this line should read if the model's constants hold")

    # ---- device time per C-ABI call inside the step, K1 roofline (rank 0 reports; every rank runs the same code) ----
    # A second, instrumented pass of the same steps, eager (events cannot sit inside the replayed graph).  An eager
    # step is host-bound here, which would count host starvation between two records as kernel time, so each step
    # is queued behind a GPU-side spin long enough for the host to run ahead.
    adj = model.conv.convs[0].adj.fwd
    adj_ptrs = {model.conv.convs[0].adj.fwd.rowptr.data_ptr(), model.conv.convs[0].adj.bwd.rowptr.data_ptr()}
    t_b2b, (kx, ky) = k1_back_to_back(adj, H)
    t_brk = k1_bracketed(adj, kx, ky)
    bracket_cost = max(t_brk - t_b2b, 0.0)  # what one event pair adds to the reading of this kernel on this box
    del kx, ky
    k1_steps = min(args.steps, 50 if N < 200000 else 10)
    calls = []
    eager = TrainStep(model, opt, loss_fn, xg, eig, ewg, bucket, use_graph=False)
    eager(pos_g[0], y_g[0])
    torch.cuda.synchronize()
    t_host = time.perf_counter()
    eager(pos_g[0], y_g[0])  # host time of one eager step (no sync inside)
    t_host = time.perf_counter() - t_host
    torch.cuda.synchronize()
    spin_cycles = int(max(t_host * 2.0, 2e-3) * 2.4e9)
    real_lib = _lib.load()
    _lib._lib = _BracketLib(real_lib, calls)
    try:
        for i in range(k1_steps):
            torch.cuda._sleep(spin_cycles)
            eager(pos_g[i % n_batches], y_g[i % n_batches])
            torch.cuda.synchronize()
    finally:
        _lib._lib = real_lib
    per_call = {}
    k1_us = []
    for name, e0, e1, a in calls:
        t_us = max(e0.elapsed_time(e1) * 1e3 - bracket_cost * 1e6, 0.0)
        per_call.setdefault(name, []).append(t_us)
        # adjacency launches only (the CSR of A or of A^T by its row pointer): not the selection product of the embedding
        # backward, which has the same (n_rows, H) under --features nodeid
        if name == "glass_spmm_csr_f32" and a[0] in adj_ptrs:
            k1_us.append(t_us)
    if not k1_us:
        raise SystemExit("bench.py: no adjacency launch of glass_spmm_csr_f32 seen in the instrumented pass")
    k1_avg = sum(k1_us) / len(k1_us) * 1e-6
    alg_bytes = _k1_alg_bytes(nnz, N, H)
    x_bytes = N * H * 4
    traffic, traffic_source = (None, "--no-pmc") if (args.no_pmc or rank != 0 or world != 1) else k1_pmc_traffic(args.workload, H, csr=(adj.rowptr, adj.col, adj.val))
    # Two levels, two utilisations (each <= 1 by construction): every gathered byte passes an XCD L2; what misses there
    # passes the memory side — Infinity Cache while X fits it, HBM beyond.
    mem_level, mem_peak = ("infinity_cache", IC_GATHER_GBPS) if x_bytes <= IC_BYTES else ("hbm", HBM_PEAK_GBPS)
    alg_rate = alg_bytes / k1_avg / 1e9
    u_l2 = alg_rate / L2_PEAK_GBPS
    if traffic is not None:
        mem_rate = min(traffic, alg_bytes) / k1_avg / 1e9 if mem_level == "hbm" else traffic / k1_avg / 1e9
        u_mem = mem_rate / mem_peak
    elif mem_level == "hbm":
        mem_rate, u_mem = alg_rate, alg_rate / mem_peak   # compulsory bytes: a lower bound of the traffic
        if u_mem > 1.0:
            # more algorithmic bytes per second than HBM can deliver: part of the gathers were served by the Infinity
            # Cache (skewed graphs); without the counters the memory-side rate is unknown — no fraction above 1 is printed
            mem_rate, u_mem = None, None
    else:
        mem_rate, u_mem = None, None
    # Third utilisation — the one that binds a ROW GATHER: every neighbour row (4H bytes per edge) is gathered through the CU's
    # vector L1 from an XCD L2 or, when it misses there, from the Infinity Cache, and the guide measured what whole-row
    # gathers reach from either level (§Indexed rows: 16.8-18.8 TB/s from L2, 8.6 TB/s from the Infinity Cache — far below
    # the 34.5 TB/s a streaming read gets from L2).  The share of gathered bytes that missed L2 comes from the counters
    # (memory-side fetches minus the streamed col / val / rowptr reads and the Y writes); the peak is the harmonic blend.
    gather_bytes = nnz * 4 * H
    gather_rate = gather_bytes / k1_avg / 1e9
    u_gather, gather_peak, miss_share = None, None, None
    if traffic is not None and x_bytes <= IC_BYTES:
        miss_bytes = min(max(traffic - (nnz * 8 + N * 4 + N * 4 * H), 0), gather_bytes)
        miss_share = miss_bytes / gather_bytes
        gather_peak = 1.0 / ((1.0 - miss_share) / L2_GATHER_GBPS + miss_share / IC_GATHER_GBPS)
        u_gather = gather_rate / gather_peak
    # ONE definition of `frac`, whatever the numbers turn out to be: the kernel's MEMORY-SIDE traffic per launch (counters: what
    # missed the XCD L2s) / launch time, against the peak of the level that serves those misses — the Infinity Cache's measured
    # whole-row gather rate while X fits it, HBM's 8 TB/s beyond.  Without counters the compulsory (algorithmic) bytes stand in at
    # the HBM level only (a lower bound of the traffic, never above 1); at the Infinity-Cache level there is no stand-in and
    # frac is null.  The other utilisations are reported beside it under their own names (frac_algorithmic_hbm = SURVEY §8(d)'s
    # figure, cache-assisted whenever X is cache-resident; frac_gather_path; frac_l2_algorithmic) — never as `frac`.
    bound, achieved, peak, frac = mem_level, mem_rate, mem_peak, u_mem
    roofline = {"bound": bound, "achieved": achieved, "peak": peak, "unit": "GB/s", "frac": frac, "traffic": traffic,
                "kernel": "glass_spmm_csr_f32 (spmm_sweep_kernel) inside the training step",
                "regime": f"X = {x_bytes / 2**20:.1f} MiB: " + (
                    "fits one XCD's 4 MiB L2" if x_bytes <= L2_XCD_BYTES else
                    "beyond one XCD's L2, inside the 256 MiB Infinity Cache (L2 misses are gathered rows from it: 8.6 TB/s "
                    "chip-wide, MI355X_MICROARCH.md §Indexed rows)" if x_bytes <= IC_BYTES else
                    "beyond the Infinity Cache: L2 misses go to HBM (8 TB/s spec)"),
                "utilisation": {"l2_algorithmic": u_l2, "memory_side": u_mem, "memory_level": mem_level,
                                "memory_side_label": "memory-side (HBM + Infinity Cache): FETCH_SIZE / WRITE_SIZE count the L2's "
                                                     "fabric requests, Infinity-Cache hits included (MI355X_MICROARCH.md §HBM) — "
                                                     "not HBM-only traffic",
                                "gather_path": u_gather, "gather_rate_GBps": gather_rate, "gather_peak_GBps": gather_peak,
                                "gathered_bytes_per_launch": gather_bytes, "gather_l2_miss_share": miss_share,
                                "gather_path_label": "neighbour-row gathers (nnz * 4H bytes) / launch time, against the guide's measured "
                                                     "whole-row gather rates blended by where the rows came from: 18.8 TB/s for the share "
                                                     "served by an XCD L2, 8.6 TB/s for the share that missed to the Infinity Cache "
                                                     "(MI355X_MICROARCH.md §Indexed rows); the miss share is measured (counters)",
                                "note": "frac = memory_side, always (memory-side traffic per launch / launch time / the serving level's "
                                        "peak); l2_algorithmic = algorithmic bytes / 34.5 TB/s (the L2 streaming rate — a row gather "
                                        "cannot reach it); gather_path = gathered bytes / the blended gather rate"},
                "frac_definition": "memory-side traffic (rocprofv3 counters, per launch) / avg launch time / peak of the level serving "
                                   "the L2 misses (infinity_cache: 8.6 TB/s whole-row gathers; hbm: 8.0 TB/s)",
                "frac_algorithmic_hbm": alg_rate / HBM_PEAK_GBPS, "frac_gather_path": u_gather, "frac_l2_algorithmic": u_l2,
                "frac_of_hbm_peak_algorithmic": alg_rate / HBM_PEAK_GBPS,
                "algorithmic_cache_assisted": alg_rate > HBM_COPY_GBPS,
                "hbm_evidence": "algorithmic bytes / 8 TB/s above 6.3 / 8 = 0.79 cannot have come from HBM alone (cache-assisted); "
                                "the HBM-bound figure of this kernel is roofline_hbm[0] (no-reuse permutation)",
                "alg_bytes_per_launch": alg_bytes, "avg_launch_us": k1_avg * 1e6, "launches_timed": len(k1_us),
                "back_to_back_us": t_b2b * 1e6, "bracketed_standalone_us": t_brk * 1e6,
                "bracket_cost_us": bracket_cost * 1e6, "traffic_source": traffic_source,
                "timing": "HIP events on the launch stream around each in-step launch, minus the bracket cost calibrated "
                          "on the same kernel/shape (bracketed - back-to-back hipGraph average)"}

    n_prof = max(k1_steps, 1)
    breakdown = sorted(((name, sum(v) / n_prof, len(v) / n_prof) for name, v in per_call.items()), key=lambda r: -r[1])
    step_breakdown = {"unit": "us per step (device time, calibrated event brackets, eager instrumented pass)",
                      "calls": {name: {"us": round(us, 2), "launch_groups": round(cnt, 2)} for name, us, cnt in breakdown},
                      "total_us": round(sum(us for _, us, _ in breakdown), 1)}
    # dense (MFMA) calls: FLOPs of the reference formulation and FLOPs executed, per C-ABI entry point
    lab_cap = pos_g.shape[1] * pos_g.shape[2] if getattr(stepper, "_labels", None) is not None else 0
    f_ref, f_ex = dense_flop_model(N, H, L, [pos_g[b].cpu().numpy() for b in range(min(n_batches, 4))], lab_cap)
    calls_of = {"glass_dual_linear_fwd_f32": ("trans_fwd", ) if "glass_comb_eff_fwd_f32" in per_call else ("trans_fwd", "comb_fwd"),
                "glass_comb_eff_fwd_f32": ("comb_fwd", ),
                "glass_comb_eff_bwd_f32": ("comb_dgrad", "comb_wgrad"),
                "glass_dual_linear_bwd_f32": ("trans_dgrad", "trans_wgrad") if "glass_comb_eff_bwd_f32" in per_call else
                                             ("trans_dgrad", "trans_wgrad", "comb_dgrad", "comb_wgrad"),
                "glass_dual_linear_dgrad_f32": ("trans_dgrad", "comb_dgrad"),
                "glass_dual_linear_wgrad_f32": ("trans_wgrad", "comb_wgrad")}
    # the peak an fp32 FLOP of the dense kernels is priced against: the f32-input MFMA peak, or — where the LDS-tiled family
    # forms an fp32 product from six bf16 partial products (glass_dense_caps.product_form; GLASS_DENSE_SPLIT=0 opts every call out) — a sixth of the bf16 peak
    from glass_amd import _lib as _glib
    caps = _glib.dense_caps(H)
    split_form = caps.product_form == 1 and not ops.DENSE_F32_PRODUCTS
    mfma_peak = MFMA_BF16_TFLOPS / 6.0 if split_form else MFMA_F32_TFLOPS
    peak_note = ("fp32 products as six bf16 partial products of 3-way split operands (v_mfma_f32_32x32x16_bf16 / _16x16x32_bf16): peak = bf16 dense "
                 "peak / 6; the f32-input MFMA peak is 157.3 TFLOP/s" if split_form else "f32-input MFMA (v_mfma_f32_*_f32)")
    top = breakdown[0]
    dominant = {"kernel": top[0], "us_per_step": round(top[1], 2), "share_of_step": round(top[1] / step_breakdown["total_us"], 3)}
    if top[0] in calls_of:
        ex = sum(f_ex[k] for k in calls_of[top[0]]) / (top[1] * 1e-6) / 1e12
        rf = sum(f_ref[k] for k in calls_of[top[0]]) / (top[1] * 1e-6) / 1e12
        dominant.update({"bound": "mfma", "achieved": ex, "peak": mfma_peak, "unit": "TFLOP/s", "frac": ex / mfma_peak,
                         "frac_reference_formulation": rf / mfma_peak, "product_form": peak_note})
    elif top[0] == "glass_spmm_csr_f32":
        dominant.update({"bound": roofline["bound"], "frac": roofline["frac"]})
    step_breakdown["dominant"] = dominant
    dense_us = sum(us for name, us, _ in breakdown if name in calls_of)
    if dense_us > 0:
        seen = [k for name in per_call if name in calls_of for k in calls_of[name]]
        ex = sum(f_ex[k] for k in seen) / (dense_us * 1e-6) / 1e12
        rf = sum(f_ref[k] for k in seen) / (dense_us * 1e-6) / 1e12
        step_breakdown["dense_mfma"] = {"us_per_step": round(dense_us, 1), "achieved": ex, "peak": mfma_peak,
                                        "unit": "TFLOP/s", "frac": ex / mfma_peak,
                                        "frac_reference_formulation": rf / mfma_peak, "product_form": peak_note,
                                        "vs_f32_mfma_peak": ex / MFMA_F32_TFLOPS,
                                        "gflop_executed_per_step": sum(f_ex[k] for k in seen) / 1e9,
                                        "gflop_reference_per_step": sum(f_ref[k] for k in seen) / 1e9,
                                        "flops": "frac counts EXECUTED FLOPs (effective-weight kernels run one product per row of the "
                                                 "comb pair where the reference formulation, impl/models.py:169-173, has two); "
                                                 "frac_reference_formulation divides the reference's FLOPs by the same time"}

    hbm = None
    if rank == 0 and world == 1 and not args.no_roofline_hbm:
        del eager
        hbm = roofline_hbm_entries(dev)

    floor = None
    if rank == 0 and world == 1 and not args.no_floor:
        floor, why = step_floor(args)
        if floor is None:
            floor = {"us": None, "reason": why}
        else:
            floor["share_of_step"] = floor["us"] * 1e-3 / (dt / args.steps * 1e3)

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(w, ei, ew, x, pos, y, args.cpu_steps)

    if rank == 0:
        out = {
            "metric": "aggregated edges/sec (GLASSConv fwd+bwd)", "value": nnz * L * args.steps * world / dt,
            "unit": "edges/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "ms_per_step_min": times[0] / args.steps * 1e3,
            "ms_per_step_max": times[-1] / args.steps * 1e3, "blocks": len(times), "timed_region_s": sum(times),
            "timing": f"median of {len(times)} blocks of exactly {args.steps} steps, each bracketed by barrier + "
                      "torch.cuda.synchronize(), max over ranks per block",
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "shipped graph + subgraphs, random-init weights" if w.name == "density" else "synthetic",
            "config": {"workload": f"{'the shipped density graph' if w.name == 'density' else w.name + '-shaped synthetic graph'} "
                                   f"({BASELINE_CONFIG.get(w.name, 'not a BASELINE config')}): N={N}, nnz={nnz}, "
                                   f"hidden={H}, layers={L}, aggr={w.aggr}, pool={w.pool}, z_ratio={w.z_ratio}, "
                                   f"dropout={w.dropout}, batch={w.batch}x{w.sub_size} per rank, "
                                   f"{'use_nodeid (V=N)' if args.features == 'nodeid' else 'use_deg'} features, Adam",
                       "parallelism": f"subgraph-batch dp{world}, replicated graph, bucketed gradient all-reduce per step",
                       "step": "MaxZOZ+fwd+loss+bwd+allreduce+Adam", "hip_graph": bool(stepper.graphed),
                       "caller": ("reference: GLASSTest.py's own objects (buildModel constructions, torch.optim.Adam, its loss callable, "
                                  "ZGDataloader shuffle+drop_last) through impl.train.train — one epoch (incl. its host sync, the "
                                  "shuffle and the batch selection) per timed block") if ref_caller else
                                 "step: bench.py calls glass_amd.step.TrainStep on pre-selected batches",
                       "labels_in_head_launch": bool(getattr(getattr(stepper, "_labels", None), "in_head", False)),
                       "final_loss": last_loss},
            "roofline": roofline, "roofline_hbm": hbm, "step_breakdown": step_breakdown, "step_floor": floor, "cpu_baseline": cpu,
            "step_floor_us": None if floor is None else floor["us"],
        }
        if collective is not None:
            out["collective"] = collective
        # the forecast a SCALE run is to be held against: per world size the exposed exchange time under
        # the stated model (glass_amd/dist.py: assumed constants) and efficiency = step / (step + exposed), from THIS line's
        # step time scaled to one rank's share of it (N = 1: the step itself; N > 1: the measured step minus its predicted exposed part)
        try:
            from glass_amd import dist as gdist
            fb = getattr(bucket, "flat", None)
            if fb is not None:
                es = fb.element_size()
                big0 = int(getattr(bucket, "big_start", fb.numel()) or fb.numel())
                big = (fb.numel() - big0) * es if (getattr(bucket, "shard_optimizer", False) or world == 1) else 0
                payload = {"small_allreduce": (big0 if big else fb.numel()) * es, "big_reduce_scatter": big, "big_all_gather": big}
                ms1 = dt / args.steps * 1e3
                overlaps = bool(big)
                if world > 1 and collective is not None and collective.get("predicted_exposed_us") is not None:
                    ms1 = max(ms1 - collective["predicted_exposed_us"] * 1e-3, 1e-6)
                fc = gdist.predict_scaling(payload, ms1, overlaps_small=overlaps)
                out["collective_forecast"] = fc
                out["predicted_efficiency"] = {str(r["world"]): round(r["ring_efficiency"], 4) for r in fc["per_world"]}
        except Exception as e:  # noqa: BLE001 — a forecast must never fail a bench line
            out["collective_forecast"] = {"error": repr(e)}
        print(json.dumps(out), flush=True)
    if world > 1:
        import torch.distributed as td
        td.destroy_process_group()


if __name__ == "__main__":
    main()
