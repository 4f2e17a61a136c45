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
