#!/bin/bash
# A/B of an environment switch on the default bench line + the phase trace.  usage: gpu_ab.sh <tag> VAR v1 v2 [pytest -k expr]
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/$1; var=$2; v1=$3; v2=$4; kexpr=$5
mkdir -p $out
for v in $v1 $v2 $v1 $v2; do
  env $var=$v python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-roofline-hbm --no-pmc > $out/bench_$v.json 2> $out/bench_$v.err
  python - <<PY
import json
d=json.load(open("$out/bench_$v.json"))
c=d["step_breakdown"]["calls"]
print("$var=$v ms_per_step %.4f" % d["ms_per_step"], {k: c[k]["us"] for k in c if "linear" in k or "comb" in k})
PY
done
for v in $v1 $v2; do
  echo "== trace $var=$v"; env $var=$v GLASS_HIP_LIB=$PWD/tools/bin/libglass_trace.so python tools/dense_trace.py ppi_bp 2>&1 | grep -A12 "comb fwd\|trans fwd" | grep -v "^--" | head -30
done
if [ -n "$kexpr" ]; then
  env $var=$v2 python -m pytest tests -m gpu -q -x --timeout 1500 -k "$kexpr" 2>&1 | tail -5
fi
