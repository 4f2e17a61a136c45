"""Summarise tools/k1_pmc.sh's rocprofv3 CSVs into <round>_k1_traffic.json: per shape, HBM-side bytes per K1 launch
(sum over the launch's kernels: sweep + long-row + reduce) = 2 x FETCH_SIZE + WRITE_SIZE (KiB counters; FETCH_SIZE halves
wide loads on gfx950, MI355X_MICROARCH.md §HBM), the average launch duration from the kernel trace, the L2 hit rate where
collected.  usage: python3 tools/k1_pmc_summary.py <dir> <round>"""
import csv
import glob
import json
import os
import sys

d, R = sys.argv[1], sys.argv[2]
SHAPES = {"ppi_bp": ("ppi_bp", 64, 17080, 633902), "powerlaw": ("powerlaw", 256, 1000000, 20000000),
          "calib_4000000": ("calib_4000000", 64, 4000000, 4000000), "2000000_3000000": ("uniform_deg3", 64, 2000000, 6000000),
          "2000000_6000000": ("uniform_deg6", 64, 2000000, 12000000)}


def per_launch(path, counter):
    """Sum the counter over the spmm kernels, divided by the number of launches (= dispatches of the first kernel
    of a launch: sweep when present, else the long-row kernel)."""
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter and "spmm_" in r["Kernel_Name"]]
    if not rows:
        return None
    names = sorted({r["Kernel_Name"].split("(")[0] for r in rows})
    lead = [n for n in names if "sweep" in n] or [n for n in names if "long" in n]
    launches = sum(1 for r in rows if r["Kernel_Name"].split("(")[0] == lead[0])
    return sum(float(r["Counter_Value"]) for r in rows) / launches


out = {"_method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE and, in a separate pass, --pmc WRITE_SIZE on tools/bin/spmm_bench "
                  "<shape> <H> 10 (raw per-dispatch CSVs: %s_k1_pmc_*).  Counter unit = KiB.  gfx950 correction: FETCH_SIZE "
                  "reports half the bytes of wide loads -> doubled (checked on the calib shape: a permutation matrix, every X "
                  "row read exactly once).  Counters are L2 memory-side requests: Infinity-Cache hits are included, so for "
                  "cache-resident shapes this is L2-miss traffic, an upper bound on HBM bytes." % R}
for tag, (name, H, n, nnz) in SHAPES.items():
    f = os.path.join(d, f"{R}_k1_pmc_{tag}_FETCH_SIZE.csv")
    w = os.path.join(d, f"{R}_k1_pmc_{tag}_WRITE_SIZE.csv")
    if not (os.path.exists(f) and os.path.exists(w)):
        continue
    fetch, write = per_launch(f, "FETCH_SIZE"), per_launch(w, "WRITE_SIZE")
    alg = nnz * (4 * H + 8) + n * (4 * H + 4)
    e = {"H": H, "alg_bytes_per_launch": alg, "fetch_corrected": int(2 * fetch * 1024), "write": int(write * 1024),
         "hbm_bytes_per_launch": int((2 * fetch + write) * 1024)}
    e["traffic_over_alg"] = round(e["hbm_bytes_per_launch"] / alg, 3)
    st = glob.glob(os.path.join(d, f"{R}_k1_{tag}_h{H}_kernel_stats.csv"))
    if st:
        rows = [r for r in csv.DictReader(open(st[0])) if "spmm_" in r["Name"]]
        lead = [r for r in rows if "sweep" in r["Name"]] or rows
        e["avg_launch_us_rocprof"] = round(sum(float(r["TotalDurationNs"]) for r in rows) / int(lead[0]["Calls"]) / 1e3, 2)
        e["hbm_frac_of_8TBps_alg"] = round(alg / (e["avg_launch_us_rocprof"] * 1e-6) / 8e12, 3)
    l2 = os.path.join(d, f"{R}_k1_pmc_{tag}_L2.csv")
    if os.path.exists(l2):
        hit, miss = per_launch(l2, "TCC_HIT_sum"), per_launch(l2, "TCC_MISS_sum")
        e["l2_hit_rate"] = round(hit / (hit + miss), 3)
    out[name] = e
print(json.dumps(out, indent=1))
