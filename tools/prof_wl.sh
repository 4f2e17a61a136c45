#!/bin/bash
# rocprofv3 kernel table of one bench workload: the bench command itself under --kernel-trace --stats.
# usage: prof_wl.sh workload steps [warmup]   -> gpurun_out/prof_wl/${ROUND}_bench_<workload>_{bench_line.json,kernel_stats.csv}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_wl
mkdir -p $out
w=$1; steps=${2:-200}; warm=${3:-10}; R=${ROUND:-r04}
rm -rf $out/$w
rocprofv3 --kernel-trace --stats --output-format csv -d $out/$w -- python3 bench.py --workload $w --steps $steps --warmup $warm --no-cpu-baseline --no-roofline-hbm --no-pmc > $out/${R}_bench_${w}_bench_line.json 2> $out/$w.err
cp $(ls $out/$w/*/*kernel_stats.csv | head -1) $out/${R}_bench_${w}_kernel_stats.csv
echo "== $w"; python3 tools/prof_summary.py $out/$w ${4:-30}
rm -rf $out/$w
