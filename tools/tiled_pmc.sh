#!/bin/bash
# where the wave cycles of the tiled dense kernels go (laboratory): issue-side counters per kernel, relative to SQ_WAVE_CYCLES.
# usage: tiled_pmc.sh workload   -> gpurun_out/tiled_pmc/summary.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/tiled_pmc
rm -rf $out; mkdir -p $out
W=${1:-powerlaw}
PATTERN=${PATTERN:-tiled}
ARGS="bench.py --workload $W --steps ${STEPS:-2} --warmup 1 --min-blocks 1 --graph 0 --no-cpu-baseline --no-roofline-hbm --no-pmc"
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_MISC SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  timeout 900 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/p$i -- python3 $ARGS > $out/p$i.log 2>&1
  cp $(ls $out/p$i/*/*counter_collection.csv | head -1) $out/raw$i.csv
  rm -rf $out/p$i
done
python3 - <<PY > $out/summary.txt
import csv, glob, re
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(float)); n = defaultdict(lambda: defaultdict(set)); dur = defaultdict(float); nd = defaultdict(set)
for f in sorted(glob.glob("$out/raw*.csv")):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"^void ", "", r["Kernel_Name"]); k = re.match(r"([\w:]+(<[^(]*>)?)", k).group(1)
        if not any(p in k for p in "${PATTERN:-tiled}".split(",")): continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]].add((f, r["Dispatch_Id"]))
        if (f, r["Dispatch_Id"]) not in nd[k]:
            nd[k].add((f, r["Dispatch_Id"])); dur[k] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
for k in sorted(acc, key=lambda k: -dur[k]):
    c = {m: acc[k][m] / len(n[k][m]) for m in acc[k]}
    wc = c.get("SQ_WAVE_CYCLES", float("nan"))
    print(k, "mean_us %.0f" % (dur[k] / len(nd[k]) / 1e3))
    for m in sorted(c):
        print("   %-28s %14.0f  /wave_cycles %.3f" % (m, c[m], c[m] / wc))
PY
cat $out/summary.txt
rm -f $out/raw*.csv
