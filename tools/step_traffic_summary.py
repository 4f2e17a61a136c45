"""Per-kernel memory-side traffic of a step from two rocprofv3 counter_collection CSVs (FETCH_SIZE pass, WRITE_SIZE pass):
mean per dispatch, bytes = 2 x FETCH_SIZE KiB (gfx950: wide loads are counted at half) / WRITE_SIZE KiB; the duration from the
same passes' dispatch records (End - Start).  usage: step_traffic_summary.py fetch.csv write.csv"""
import csv
import sys
from collections import defaultdict


def load(path, ctr):
    val, dur, n = defaultdict(float), defaultdict(float), defaultdict(int)
    for r in csv.DictReader(open(path)):
        if r.get("Counter_Name") != ctr:
            continue
        k = r["Kernel_Name"]
        val[k] += float(r["Counter_Value"])
        n[k] += 1
        if r.get("Start_Timestamp") and r.get("End_Timestamp"):
            dur[k] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    return val, dur, n


fv, fd, fn = load(sys.argv[1], "FETCH_SIZE")
wv, wd, wn = load(sys.argv[2], "WRITE_SIZE")
print("kernel,dispatches,mean_us_under_counters,fetch_MB,write_MB,total_MB,TB_per_s")
rows = []
for k in fv:
    if fn[k] < 5 or k not in wv:
        continue
    f = 2.0 * fv[k] / fn[k] * 1024 / 1e6
    w = wv[k] / wn[k] * 1024 / 1e6
    us = (fd[k] / fn[k]) / 1e3 if fd[k] else float("nan")
    rows.append((f + w, k, fn[k], us, f, w))
for tot, k, n, us, f, w in sorted(rows, reverse=True)[:24]:
    short = k.split("(")[0][:70]
    print(f'"{short}",{n},{us:.1f},{f:.1f},{w:.1f},{tot:.1f},{tot / us if us == us and us > 0 else float("nan"):.2f}')  # MB / us = TB/s
