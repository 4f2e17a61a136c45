"""Task-level parity over repeats: run THIS repo's driver (GLASSTest.py, the reference's flags) for
`--repeat R` repeats per shipped synthetic set and put its per-repeat test micro-F1 beside what the reference's own driver
printed for the same command (tests/golden/g12_f1_<dataset>_<feature>.npz, written by tests/golden/make_f1_table.py from
/root/reference/GLASSTest.py:178-269 on CPU).  The two runs do not share a trajectory (different shuffles and dropout bits,
a rounding-defined `use_one` embedding: SURVEY.md App. B.1): what must agree are the DISTRIBUTIONS —
|mean_gpu - mean_ref| <= 2 sqrt(se_gpu^2 + se_ref^2) per set.

    python tools/f1_table.py [--datasets density,cut_ratio,coreness,component] [--features use_one,use_deg] [--repeat 10]
-> gpurun_out/f1_table_r05.json (copied to profiles/r05_f1_table.json) and a printed table."""
import argparse
import contextlib
import io
import json
import math
import os
import re
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np


def run_driver(dataset, feature, repeat):
    """One invocation of the repo driver; -> per-repeat (epochs, val, tst) parsed from its own log lines + wall seconds."""
    import GLASSTest
    buf = io.StringIO()
    t0 = time.time()
    cwd = os.getcwd()
    os.chdir(ROOT)  # the driver reads config/<dataset>.yml and dataset_/ relative to the repo root, as the reference does
    try:
        with contextlib.redirect_stdout(buf):
            results = GLASSTest.main([f"--{feature}", "--use_seed", "--use_maxzeroone", "--repeat", str(repeat), "--device", "0",
                                      "--dataset", dataset])
    finally:
        os.chdir(cwd)
    wall = time.time() - t0
    end = re.compile(r"^end: epoch (\d+), train time ([0-9.]+) s, val ([0-9.]+), tst ([0-9.]+)", re.M)
    rows = end.findall(buf.getvalue())
    assert len(rows) == repeat == len(results), (len(rows), repeat)
    return {"epochs": [int(r[0]) for r in rows], "train_seconds": [float(r[1]) for r in rows], "val": [float(r[2]) for r in rows],
            "tst": [float(t) for t in results], "wall_seconds": wall}


def compare(gpu_tst, ref_tst):
    g, r = np.asarray(gpu_tst, dtype=np.float64), np.asarray(ref_tst, dtype=np.float64)
    se = lambda a: float(a.std() / math.sqrt(len(a)))  # noqa: E731  (the driver's own "error" column, GLASSTest.py:266-268)
    gap, bound = abs(float(g.mean()) - float(r.mean())), 2.0 * math.sqrt(se(g) ** 2 + se(r) ** 2)
    return {"gpu_mean": float(g.mean()), "gpu_se": se(g), "ref_mean": float(r.mean()), "ref_se": se(r), "gap": gap,
            "bound_2sigma": bound, "ok": bool(gap <= bound)}


def reference_table(dataset, feature):
    path = os.path.join(ROOT, "tests", "golden", f"g12_f1_{dataset}_{feature}.npz")
    if not os.path.exists(path):
        return None
    z = np.load(path, allow_pickle=False)
    return {"tst": [float(v) for v in z["tst"]], "val": [float(v) for v in z["val"]], "epochs": [int(v) for v in z["epochs"]],
            "command": str(z["command"]), "train_seconds": [float(v) for v in z["train_seconds"]]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--datasets", default="density,cut_ratio,coreness,component")
    ap.add_argument("--features", default="use_one,use_deg")
    ap.add_argument("--repeat", type=int, default=10)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "f1_table_r05.json"))
    a = ap.parse_args()
    table = {}
    for feature in a.features.split(","):
        for ds in a.datasets.split(","):
            ref = reference_table(ds, feature)
            if ref is None:
                print(f"{ds} --{feature}: no reference fixture, skipped", flush=True)
                continue
            gpu = run_driver(ds, feature, a.repeat)
            cmp_ = compare(gpu["tst"], ref["tst"][:a.repeat] if a.repeat < len(ref["tst"]) else ref["tst"])
            table[f"{ds}/{feature}"] = {"gpu": gpu, "reference": ref, "compare": cmp_}
            print(f"{ds:10s} --{feature}: gpu {cmp_['gpu_mean']:.3f} +- {cmp_['gpu_se']:.3f}   reference {cmp_['ref_mean']:.3f} +- "
                  f"{cmp_['ref_se']:.3f}   |gap| {cmp_['gap']:.3f} <= {cmp_['bound_2sigma']:.3f}: {cmp_['ok']}   "
                  f"(gpu {sum(gpu['train_seconds']):.1f} s of training for {sum(gpu['epochs'])} epochs; reference "
                  f"{sum(ref['train_seconds']):.0f} s for {sum(ref['epochs'])})", flush=True)
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    with open(a.out, "w") as f:
        json.dump(table, f, indent=1)
    bad = [k for k, v in table.items() if not v["compare"]["ok"]]
    print("ALL WITHIN 2 SIGMA" if not bad else f"OUTSIDE 2 SIGMA: {bad}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
