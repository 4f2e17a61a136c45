#!/bin/bash
# The four shipped synthetic sets through the drop-in driver with their own YAMLs (hidden 8 / 17 / 20), the density run
# under rocprofv3: the kernel table must hold no library GEMM (Cijk_*) row.  Output: gpurun_out/shipped/
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/shipped
mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 GLASSTest.py --use_one --use_seed --use_maxzeroone --repeat 1 --device 0 --dataset density > $out/density.log 2> $out/density.err
cp $(ls $out/trace/*/*kernel_stats.csv | head -1) $out/r03_driver_density_kernel_stats.csv
echo "library GEMM rows in the density run: $(grep -c Cijk $out/r03_driver_density_kernel_stats.csv)"
head -30 $out/r03_driver_density_kernel_stats.csv | cut -c1-150
tail -3 $out/density.log
rm -rf $out/trace
for d in cut_ratio coreness component; do
  python3 GLASSTest.py --use_one --use_seed --use_maxzeroone --repeat 1 --device 0 --dataset $d > $out/$d.log 2> $out/$d.err
  echo "== $d"; tail -2 $out/$d.log; tail -2 $out/$d.err
done
