#!/bin/bash
# Matrix-core and LDS counters of the fused dense kernels (and of the library GEMM of the same shapes, for scale):
# rocprofv3 --pmc over `tools/bench_ops.py dual`, two passes (SQ slots), per-dispatch CSVs + a per-kernel summary.
# Run on the GPU box from the repo root: bash tools/dense_pmc.sh ; output in gpurun_out/dense_pmc/ (copy what is to
# be kept to profiles/).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/dense_pmc
mkdir -p $out
R=${ROUND:-r02}
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/mfma -- python3 tools/bench_ops.py dual > $out/mfma.log 2>&1
cp $(ls $out/mfma/*/*counter_collection.csv | head -1) $out/mfma_raw.csv
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES --output-format csv -d $out/lds -- python3 tools/bench_ops.py dual > $out/lds.log 2>&1
cp $(ls $out/lds/*/*counter_collection.csv | head -1) $out/lds_raw.csv
python3 tools/dense_pmc_summary.py $out/mfma_raw.csv $out/lds_raw.csv > $out/${R}_dense_pmc_summary.csv
cat $out/${R}_dense_pmc_summary.csv
