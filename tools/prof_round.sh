#!/bin/bash
# rocprofv3 summaries for profiles/: the bench command itself under --kernel-trace --stats (C2 default and C5), then the
# K1 stand-alone runs + PMC passes (tools/k1_pmc.sh).  Output: gpurun_out/prof/ ; copy the *.csv / *.json to profiles/.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
R=${ROUND:-r03}
out=gpurun_out/prof
mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/c2 -- python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-roofline-hbm --no-pmc > $out/${R}_bench_ppi_bp_bench_line.json 2> $out/c2.err
cp $(ls $out/c2/*/*kernel_stats.csv | head -1) $out/${R}_bench_ppi_bp_kernel_stats.csv
python3 tools/prof_summary.py $out/c2 25
rocprofv3 --kernel-trace --stats --output-format csv -d $out/c5 -- python3 bench.py --workload powerlaw --steps 20 --warmup 3 --no-cpu-baseline --no-roofline-hbm --no-pmc > $out/${R}_bench_powerlaw_bench_line.json 2> $out/c5.err
cp $(ls $out/c5/*/*kernel_stats.csv | head -1) $out/${R}_bench_powerlaw_kernel_stats.csv
python3 tools/prof_summary.py $out/c5 25
ROUND=$R bash tools/k1_pmc.sh
cp gpurun_out/k1_pmc/${R}_*.csv gpurun_out/k1_pmc/${R}_k1_traffic.json $out/ 2>/dev/null
ls $out
