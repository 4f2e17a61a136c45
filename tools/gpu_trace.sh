#!/bin/bash
# phase timeline of the hidden-64 dense kernels (trace build) + the default bench line.  gpurun_out/$1/
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/${1:-trace}
mkdir -p $out
GLASS_HIP_LIB=$PWD/tools/bin/libglass_trace.so python tools/dense_trace.py ppi_bp > $out/dense_trace.txt 2>&1
cat $out/dense_trace.txt
python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-roofline-hbm --no-pmc > $out/bench_c2.json 2> $out/bench_c2.err; echo "bench c2 rc=$?"
python - <<PY
import json
d=json.load(open("$out/bench_c2.json"))
print("ms_per_step", d["ms_per_step"])
for k,v in d["step_breakdown"]["calls"].items(): print(k, v)
PY
