#!/bin/bash
# K1 HBM-side traffic with rocprofv3 PMC counters, one counter per pass (MI355X_MICROARCH.md: separate --pmc passes;
# FETCH_SIZE is reported in KiB and halves wide loads on gfx950 -> doubled in the summary).  Run on the GPU box from the
# repo root: bash tools/k1_pmc.sh ; the per-dispatch CSVs land in gpurun_out/k1_pmc/ (copy the ones to keep to profiles/).
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/k1_pmc
mkdir -p $out
for spec in "ppi_bp 64" "powerlaw 256" "calib:4000000 64"; do
  set -- $spec
  tag=$(echo $1 | tr ':' '_')
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $out/${tag}_$ctr -- ./tools/bin/spmm_bench $1 $2 10 > $out/${tag}_$ctr.log 2>&1
    cp $(ls $out/${tag}_$ctr/*/*counter_collection.csv | head -1) $out/r01_k1_pmc_${tag}_$ctr.csv
  done
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_stats -- ./tools/bin/spmm_bench $1 $2 30 > $out/${tag}_stats.log 2>&1
  cp $(ls $out/${tag}_stats/*/*kernel_stats.csv | head -1) $out/r01_k1_${tag}_kernel_stats.csv
done
ls -la $out/*.csv
