#!/bin/bash
# rocprofv3 kernel table of the link-prediction pre-training step (tools/ssl_step.py) at ppi_bp-shape.
# usage (GPU box): bash tools/prof_ssl.sh [round] [mode: program|graph|eager]   -> gpurun_out/<round>_ssl_step_kernel_stats.csv + summary
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ROUND=${1:-r04}
MODE=${2:-program}
out=gpurun_out/ssl_trace
rm -rf $out
# (eager launches under the profiler: a replayed hipGraph shows up as one opaque launch)
EAGER=$MODE; [ "$MODE" = "program" ] && EAGER=program_eager; [ "$MODE" = "graph" ] && EAGER=eager
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 tools/ssl_step.py ppi_bp 50 2 0.5 131072 $EAGER > gpurun_out/${ROUND}_ssl_step.txt 2> gpurun_out/ssl_step.err
cp $(ls $out/*/*kernel_stats.csv | head -1) gpurun_out/${ROUND}_ssl_step_kernel_stats.csv
python3 tools/prof_summary.py $out 40 >> gpurun_out/${ROUND}_ssl_step.txt
python3 tools/ssl_step.py ppi_bp 200 2 0.5 131072 $MODE >> gpurun_out/${ROUND}_ssl_step.txt 2>> gpurun_out/ssl_step.err
cat gpurun_out/${ROUND}_ssl_step.txt | cut -c1-150
rm -rf $out
