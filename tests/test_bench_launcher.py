"""bench.py --gpus N must start the N ranks itself (missing 1): run on the CPU box with --dry-run, which
keeps the whole launch protocol (child torch.distributed.run job, process group, barrier, max-over-ranks, rank 0 prints
one JSON line) and skips the GPU work."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*flags):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *flags], env=env, capture_output=True, text=True,
                       timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout  # exactly one JSON line, from rank 0
    return json.loads(lines[0])


def test_gpus_flag_spawns_that_many_ranks():
    out = _run("--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-run")
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["dry_run"] is True and out["scaling"] == "weak"


def test_single_rank_dry_run():
    out = _run("--steps", "2", "--warmup", "0", "--dry-run")
    assert out["n_gpus"] == 1


def test_child_failure_propagates():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    # no GPU on this box and no --dry-run: every rank exits non-zero, and so must the launcher
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    import torch
    if not torch.cuda.is_available():
        assert p.returncode != 0


def test_more_ranks_than_gpus_is_refused_before_any_rank_starts():
    """`--gpus N` beyond the node's GPU count under RCCL: a clear message and a non-zero exit from the launcher itself —
    no rank is started that would wait in init_process_group for peers that cannot come up."""
    import torch
    n = torch.cuda.device_count()
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "GLASS_BENCH_BACKEND")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n + 2), "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=120)
    assert p.returncode == 2 and "RCCL needs one GPU per rank" in p.stderr
    assert "torch.distributed" not in p.stderr  # refused by the launcher, not by a failing child job


def test_eight_ranks_dry_run_end_to_end():
    """the N = 8 launch the driver will run, end to end on the CPU box — launcher -> 8 ranks -> rank 0
    generates the workload and the other seven load it from the published file BEFORE any collective -> gloo group ->
    barrier / max-over-ranks -> one JSON line whose `collective` block answers "how many ranks did the backend see"."""
    out = _run("--gpus", "8", "--steps", "2", "--warmup", "0", "--dry-run")
    assert out["n_gpus"] == 8 and out["dry_run"] is True
    c = out["collective"]
    assert c["world_size"] == 8 and c["backend"] == "gloo"
    assert {"in_graph", "capture_error", "exposed_us", "payload_bytes", "rccl_version"} <= set(c)
    assert out["config"]["workload_identical_on_all_ranks"] is True and out["config"]["batches_per_rank"] == 2
    import glob
    assert not glob.glob("/tmp/glass_bench_*_ppi_bp_8.npz*"), "rank 0 must remove the published workload"


def test_profiler_environment_is_detected_and_stripped():
    """bench.py's child `rocprofv3 --pmc` passes are skipped when bench.py itself runs under a profiler
    (LD_PRELOAD of the rocprofiler tool library / ROCP_* / ROCPROF* variables), and a child environment never inherits those."""
    sys.path.insert(0, ROOT)
    import bench
    assert not bench.under_profiler({"PATH": "/usr/bin"})
    assert bench.under_profiler({"LD_PRELOAD": "/opt/rocm/lib/librocprofiler-sdk-tool.so"})
    assert bench.under_profiler({"ROCP_TOOL_LIBRARIES": "x"}) and bench.under_profiler({"ROCPROF_OUTPUT_PATH": "/tmp"})
    env = bench.child_env_without_profiler({"LD_PRELOAD": "/opt/rocm/lib/librocprofiler-sdk-tool.so", "ROCP_TOOL_LIBRARIES": "x",
                                            "ROCPROFILER_X": "1", "HSA_TOOLS_LIB": "y", "PATH": "/usr/bin", "HOME": "/root"})
    assert env == {"PATH": "/usr/bin", "HOME": "/root"}
    assert bench.child_env_without_profiler({"LD_PRELOAD": "/lib/libfoo.so"}) == {"LD_PRELOAD": "/lib/libfoo.so"}
    import unittest.mock as mock
    with mock.patch.dict(os.environ, {"ROCP_TOOL_LIBRARIES": "x"}):
        assert bench.k1_pmc_traffic("ppi_bp", 64) == (None, mock.ANY) or bench.k1_pmc_traffic("ppi_bp", 64)[0] is None


def test_trailing_period_of_a_kernel_trace():
    """bench.step_floor finds one replayed step in a kernel trace as the shortest trailing period seen three times."""
    import bench
    step = ["labels", "pack", "fwd", "k1", "k1", "bwd", "adam"]
    assert bench.trailing_period(["setup", "csr", "warm"] + step * 5) == len(step)
    assert bench.trailing_period(["a", "b"] * 3) == 2
    assert bench.trailing_period(["a", "b", "c", "a", "b", "c", "a", "b"]) is None      # fewer than three full periods
    assert bench.trailing_period(["x"] * 2 + step * 2) is None


def test_f1_distribution_comparison_rule():
    """tools/f1_table.compare: |mean_gpu - mean_ref| <= 2 sqrt(se_gpu^2 + se_ref^2) with the driver's own standard error
    (np.std / sqrt(n), GLASSTest.py:266-268)."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import f1_table
    import numpy as np
    a, b = [0.90, 0.92, 0.94, 0.96], [0.91, 0.93, 0.95, 0.97]
    c = f1_table.compare(a, b)
    se = np.std(a) / 2.0
    assert abs(c["gpu_se"] - se) < 1e-12 and abs(c["gap"] - 0.01) < 1e-12 and abs(c["bound_2sigma"] - 2 * np.sqrt(2) * se) < 1e-12 and c["ok"]
    assert not f1_table.compare([0.5] * 4, [0.9, 0.91, 0.9, 0.91])["ok"]
    ref = f1_table.reference_table("density", "use_one")
    assert ref is not None and len(ref["tst"]) == 10 and "--use_one" in ref["command"] and "--repeat 10" in ref["command"]
