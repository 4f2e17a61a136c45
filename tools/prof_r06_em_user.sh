#!/bin/bash
# Round-6 refresh of config 4's profile files alone (after a hidden-128 kernel change): kernel table + bench line, default bench
# line, per-kernel counter summary.  Output: gpurun_out/final4/ ; copy the r06_* files to profiles/.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
R=${ROUND:-r06}
out=gpurun_out/final4
rm -rf $out; mkdir -p $out
w=em_user
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/$w -- python3 bench.py --workload $w --steps 200 --warmup 20 --no-cpu-baseline --no-roofline-hbm --no-pmc --no-floor > $out/${R}_bench_${w}_bench_line.json 2> $out/$w.err
cp $(ls $out/$w/*/*kernel_stats.csv | head -1) $out/${R}_bench_${w}_kernel_stats.csv
echo "== $w"; python3 tools/prof_summary.py $out/$w 8 | cut -c1-70,88-140
rm -rf $out/$w
timeout 900 python3 bench.py --workload $w > $out/${R}_bench_${w}_default_bench_line.json 2> $out/default_$w.err; echo "$w default rc=$?"
python3 -c "
import json; d=json.load(open('$out/${R}_bench_${w}_default_bench_line.json')); r=d['roofline']; c=d['cpu_baseline']
print('$w ms %.4f value %.3e frac %s (%s) traffic %s floor %s cpu %s' % (d['ms_per_step'], d['value'], r['frac'], r['bound'], r['traffic'], d['step_floor'].get('us'), c and round(c['ms_per_step'],1)))"
ROUND=$R bash tools/step_pmc.sh $w > $out/pmc_$w.log 2>&1; cp gpurun_out/step_pmc/${R}_step_pmc_${w}_summary.csv $out/
head -9 $out/${R}_step_pmc_${w}_summary.csv | cut -c1-160
