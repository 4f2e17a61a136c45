#!/bin/bash
# K1 per-launch durations (rocprofv3 --kernel-trace --stats) and HBM-side traffic (PMC counters, one counter per pass:
# MI355X_MICROARCH.md — separate --pmc passes; FETCH_SIZE is reported in KiB and halves wide loads on gfx950 -> doubled in
# the summary).  Run on the GPU box from the repo root: bash tools/k1_pmc.sh ; the per-dispatch CSVs land in
# gpurun_out/k1_pmc/ (copy the ones to keep to profiles/).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/k1_pmc
mkdir -p $out
R=${ROUND:-r02}
for spec in "ppi_bp 64" "powerlaw 256" "calib:4000000 64" "2000000:3000000 64" "2000000:6000000 64"; do
  set -- $spec
  tag=$(echo $1 | tr ':' '_')
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_stats -- ./tools/bin/spmm_bench $1 $2 30 > $out/${tag}_stats.log 2>&1
  cp $(ls $out/${tag}_stats/*/*kernel_stats.csv | head -1) $out/${R}_k1_${tag}_h$2_kernel_stats.csv
  tail -1 $out/${tag}_stats.log
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $out/${tag}_$ctr -- ./tools/bin/spmm_bench $1 $2 10 > $out/${tag}_$ctr.log 2>&1
    cp $(ls $out/${tag}_$ctr/*/*counter_collection.csv | head -1) $out/${R}_k1_pmc_${tag}_$ctr.csv
  done
done
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/ppi_L2 -- ./tools/bin/spmm_bench ppi_bp 64 10 > $out/ppi_L2.log 2>&1
cp $(ls $out/ppi_L2/*/*counter_collection.csv | head -1) $out/${R}_k1_pmc_ppi_bp_L2.csv
python3 tools/k1_pmc_summary.py $out $R > $out/${R}_k1_traffic.json
cat $out/${R}_k1_traffic.json
ls $out/*.csv | wc -l
