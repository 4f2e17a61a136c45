"""Per-kernel summary of tools/dense_pmc.sh's per-dispatch counter CSVs (fused dense kernels + the library GEMMs run
beside them).  Columns: dispatches, mean duration (counter runs serialise kernels: slightly above the timing runs),
MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x 128) — the share of SIMD cycles a matrix-core instruction
was executing: the busy counter is summed over the 1 024 SIMDs, GRBM_GUI_ACTIVE over the 8 XCDs (checked on the library
GEMM: 0.87 at 137 TF/s = 0.87 of the fp32 peak) —, LDS bank-conflict cycles / LDS active cycles, LDS issue stalls / wave cycles.
usage: python3 tools/dense_pmc_summary.py <mfma_raw.csv> <lds_raw.csv>"""
import csv
import re
import sys
from collections import defaultdict

SIMDS_PER_XCD = 32 * 4  # GRBM_GUI_ACTIVE is summed over the 8 XCDs


def short(name):
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([\w:]+(<[^(]*>)?)", name)
    return (m.group(1) if m else name)[:90]


def load(path):
    acc = defaultdict(lambda: defaultdict(float))
    n = defaultdict(set)
    dur = defaultdict(float)
    for r in csv.DictReader(open(path)):
        k = short(r["Kernel_Name"])
        if not ("glass::" in k or "Cijk" in k):
            continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in n[k]:
            n[k].add(r["Dispatch_Id"])
            dur[k] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    return acc, {k: len(v) for k, v in n.items()}, dur


mf, n1, d1 = load(sys.argv[1])
ld, n2, d2 = load(sys.argv[2])
print("kernel,dispatches,mean_us,mfma_busy_frac,lds_conflict_over_active,lds_issue_stall_over_wave_cycles")
for k in sorted(mf, key=lambda k: -d1[k]):
    c = mf[k]
    gui = c.get("GRBM_GUI_ACTIVE", 0.0)
    busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (gui * SIMDS_PER_XCD) if gui else float("nan")
    l = ld.get(k, {})
    conf = l.get("SQ_LDS_BANK_CONFLICT", 0.0) / l["SQ_LDS_IDX_ACTIVE"] if l.get("SQ_LDS_IDX_ACTIVE") else float("nan")
    stall = l.get("SQ_WAIT_INST_LDS", 0.0) / l["SQ_WAVE_CYCLES"] if l.get("SQ_WAVE_CYCLES") else float("nan")
    print(f"\"{k}\",{n1[k]},{d1[k] / n1[k] / 1e3:.1f},{busy:.3f},{conf:.3f},{stall:.3f}")
