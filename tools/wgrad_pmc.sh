#!/bin/bash
# memory- and issue-side counters of the tiled weight-gradient launch alone (tools/wgrad_probe.py, hidden 256, N = 1 M;
# laboratory).  usage: wgrad_pmc.sh  -> gpurun_out/wgrad_pmc/summary.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/wgrad_pmc
rm -rf $out; mkdir -p $out
i=0
for set in "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_MFMA" "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/p$i -- python3 tools/wgrad_probe.py 256 1000000 ${PROBE_ITERS:-4} > $out/p$i.log 2>&1
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
        if "wgrad" not in k: continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]].add((f, r["Dispatch_Id"]))
        if (f, r["Dispatch_Id"]) not in nd[k]:
            nd[k].add((f, r["Dispatch_Id"])); dur[k] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
for k in sorted(acc, key=lambda k: -dur[k]):
    c = {m: acc[k][m] / len(n[k][m]) for m in acc[k]}
    print(k, "mean_us %.0f" % (dur[k] / len(nd[k]) / 1e3), "dispatches", len(nd[k]))
    for m in sorted(c):
        print("   %-30s %16.0f" % (m, c[m]))
PY
cat $out/summary.txt
rm -f $out/raw*.csv
