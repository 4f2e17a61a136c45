#!/bin/bash
# Memory-side traffic of every kernel of a training step (eager launches of bench.py, counters per dispatch): rocprofv3 --pmc
# FETCH_SIZE and, in a separate pass, WRITE_SIZE (KiB units; FETCH_SIZE doubled per the gfx950 rule, MI355X_MICROARCH.md),
# plus a kernel trace for the durations.  Output: gpurun_out/step_traffic/<round>_step_traffic_<workload>.csv
# usage: tools/step_traffic.sh [workload]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/step_traffic
rm -rf $out; mkdir -p $out
R=${ROUND:-r06}
W=${1:-em_user}
ARGS="bench.py --workload $W --steps ${STEPS:-30} --warmup ${WARM:-5} --min-blocks 1 --graph 0 --no-cpu-baseline --no-roofline-hbm --no-pmc"
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $out/$ctr -- python3 $ARGS > $out/$ctr.log 2>&1
  cp $(ls $out/$ctr/*/*counter_collection.csv | head -1) $out/${ctr}_raw.csv
done
python3 tools/step_traffic_summary.py $out/FETCH_SIZE_raw.csv $out/WRITE_SIZE_raw.csv > $out/${R}_step_traffic_${W}.csv
cat $out/${R}_step_traffic_${W}.csv | cut -c1-200
rm -rf $out/FETCH_SIZE $out/WRITE_SIZE $out/*_raw.csv
