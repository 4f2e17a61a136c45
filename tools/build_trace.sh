#!/bin/bash
# laboratory build of the library with the phase stamps of tools/dense_trace.py (-DGLASS_DENSE_TRACE) -> tools/bin/libglass_trace.so
# extra flags: tools/build_trace.sh -DSOMETHING
cd "$(dirname "$0")/../glass_amd/csrc" && mkdir -p ../../tools/bin/trace
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=on -DGLASS_DENSE_TRACE "$@" -c dense.hip -o ../../tools/bin/trace/dense.o &&
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=on -DGLASS_DENSE_TRACE "$@" -c dense_tiled.hip -o ../../tools/bin/trace/dense_tiled.o &&
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC spmm.o graphnorm.o elementwise.o labels.o embnorm.o pool.o linear.o wgrad_tiled.o ../../tools/bin/trace/dense.o ../../tools/bin/trace/dense_tiled.o dense_narrow.o head.o readout.o pairhead.o api.o -o ../../tools/bin/libglass_trace.so && echo "trace lib built"
