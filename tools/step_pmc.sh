#!/bin/bash
# Matrix-core and LDS counters of every kernel of the C2 training step (eager launches of bench.py, counters per dispatch):
# rocprofv3 --pmc in two passes (SQ slots), then tools/dense_pmc_summary.py.  Output: gpurun_out/step_pmc/ (copy the summary to profiles/).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/step_pmc
rm -rf $out; mkdir -p $out
R=${ROUND:-r04}
W=${1:-ppi_bp}
ARGS="bench.py --workload $W --steps ${STEPS:-30} --warmup ${WARM:-5} --min-blocks 1 --graph 0 --no-cpu-baseline --no-roofline-hbm --no-pmc"
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/mfma -- python3 $ARGS > $out/mfma.log 2>&1
cp $(ls $out/mfma/*/*counter_collection.csv | head -1) $out/mfma_raw.csv
timeout 600 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES --output-format csv -d $out/lds -- python3 $ARGS > $out/lds.log 2>&1
cp $(ls $out/lds/*/*counter_collection.csv | head -1) $out/lds_raw.csv
python3 tools/dense_pmc_summary.py $out/mfma_raw.csv $out/lds_raw.csv > $out/${R}_step_pmc_${W}_summary.csv
cat $out/${R}_step_pmc_${W}_summary.csv | cut -c1-220
rm -rf $out/mfma $out/lds $out/mfma_raw.csv $out/lds_raw.csv
