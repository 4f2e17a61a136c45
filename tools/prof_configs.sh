#!/bin/bash
# rocprofv3 kernel tables of the other BASELINE configs (C1 shipped density graph, C3 hpo_neuro-shape, C4 em_user-shape):
# the bench command itself under --kernel-trace --stats.  Output: gpurun_out/prof3/ ; copy the rNN_*.csv to profiles/.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof3
mkdir -p $out
for w in density hpo_neuro em_user; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/$w -- python3 bench.py --workload $w --steps 200 --warmup 20 --no-cpu-baseline --no-roofline-hbm --no-pmc > $out/${ROUND:-r03}_bench_${w}_bench_line.json 2> $out/$w.err
  cp $(ls $out/$w/*/*kernel_stats.csv | head -1) $out/${ROUND:-r03}_bench_${w}_kernel_stats.csv
  echo "== $w"; python3 tools/prof_summary.py $out/$w 30
done
