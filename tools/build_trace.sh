#!/bin/bash
# Laboratory build of the library -> tools/bin/libglass_trace.so: the phase stamps of tools/dense_trace.py
# (-DGLASS_DENSE_TRACE) and the laboratory variants / knobs (-DGLASS_LAB=1: tools/lab/*.inc kernels, environment knobs through
# tools/lab/lab_knobs.cpp — GLASS_FWD_WG, GLASS_TRANS_DGRAD3, GLASS_COMB_FWD_PF, GLASS_COMB_FWD3, GLASS_TILED_H128_ROWS128,
# GLASS_TILED_LDS_PAD).  The product library has none of this.  Extra flags: tools/build_trace.sh -DSOMETHING
cd "$(dirname "$0")/../glass_amd/csrc" && mkdir -p ../../tools/bin/trace
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=on -DGLASS_DENSE_TRACE -DGLASS_LAB=1"
for f in dense dense_tiled linear wgrad_tiled; do
  /opt/rocm/bin/hipcc $FLAGS "$@" -c $f.hip -o ../../tools/bin/trace/$f.o || exit 1
done
/opt/rocm/bin/hipcc $FLAGS -c ../../tools/lab/lab_knobs.cpp -o ../../tools/bin/trace/lab_knobs.o || exit 1
T=../../tools/bin/trace
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC spmm.o graphnorm.o elementwise.o labels.o embnorm.o pool.o $T/linear.o $T/wgrad_tiled.o $T/dense.o $T/dense_tiled.o dense_narrow.o head.o readout.o pairhead.o peer.o wgrad128.o api.o $T/lab_knobs.o -o ../../tools/bin/libglass_trace.so && echo "trace lib built"
