#!/bin/bash
# rocprofv3 kernel table of the link-prediction pre-training step (tools/ssl_step.py) at ppi_bp-shape.
# usage (GPU box): bash tools/prof_ssl.sh [round]   -> gpurun_out/<round>_ssl_step_kernel_stats.csv + summary
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ROUND=${1:-r03}
out=gpurun_out/ssl_trace
rm -rf $out
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 tools/ssl_step.py ppi_bp 50 > gpurun_out/${ROUND}_ssl_step.txt 2> gpurun_out/ssl_step.err
cp $(ls $out/*/*kernel_stats.csv | head -1) gpurun_out/${ROUND}_ssl_step_kernel_stats.csv
python3 tools/prof_summary.py $out 30 >> gpurun_out/${ROUND}_ssl_step.txt
cat gpurun_out/${ROUND}_ssl_step.txt | cut -c1-150
rm -rf $out
