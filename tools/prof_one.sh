#!/bin/bash
# refresh the profiles/ files of ONE workload: kernel table + bench line under rocprofv3, default bench line, counter summary
# usage: prof_one.sh <workload> [round]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
w=${1:-em_user}; R=${2:-r05}
out=gpurun_out/final; mkdir -p $out
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/$w -- python3 bench.py --workload $w --steps 200 --warmup 20 --no-cpu-baseline --no-roofline-hbm --no-pmc --no-floor > $out/${R}_bench_${w}_bench_line.json 2> $out/$w.err
cp $(ls $out/$w/*/*kernel_stats.csv | head -1) $out/${R}_bench_${w}_kernel_stats.csv
python3 tools/prof_summary.py $out/$w 10 | cut -c1-75,88-140
rm -rf $out/$w
timeout 900 python3 bench.py --workload $w > $out/${R}_bench_${w}_default_bench_line.json 2> $out/default_$w.err; echo "$w default rc=$?"
python3 -c "
import json; d=json.load(open('$out/${R}_bench_${w}_default_bench_line.json')); r=d['roofline']; print('$w ms %.4f frac %s floor %.1f dense %s' % (d['ms_per_step'], r['frac'], d['step_floor']['us'], d['step_breakdown'].get('dense_mfma',{}).get('us_per_step')))"
ROUND=$R bash tools/step_pmc.sh $w > $out/pmc_$w.log 2>&1; cp gpurun_out/step_pmc/${R}_step_pmc_${w}_summary.csv $out/; head -8 $out/${R}_step_pmc_${w}_summary.csv | cut -c1-140
