#!/bin/bash
# rocprofv3 kernel table of the default bench command (C2): per-step launches and time per kernel.
# usage (GPU box): bash tools/prof_c2.sh <tag> [extra bench args]   -> gpurun_out/<tag>/
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=${1:-prof_c2}; shift
out=gpurun_out/$tag
mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-roofline-hbm --no-pmc "$@" > $out/bench_line.json 2> $out/bench.err
cp $(ls $out/trace/*/*kernel_stats.csv | head -1) $out/kernel_stats.csv
python3 tools/prof_summary.py $out/trace 45
python3 -c "
import json;d=json.load(open('$out/bench_line.json'));print('ms_per_step', d['ms_per_step'])"
rm -rf $out/trace
