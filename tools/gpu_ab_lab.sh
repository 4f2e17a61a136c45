#!/bin/bash
# A/B of a LABORATORY knob (tools/lab/lab_knobs.cpp: read from the environment by tools/bin/libglass_trace.so only) on a bench
# workload, alternating on one box, + optional parity tests with the second value.
# usage: gpu_ab_lab.sh VAR v1 v2 workload steps [pytest -k expr]
cd "$GRAFT_REPO_ROOT"
var=$1; v1=$2; v2=$3; wl=${4:-ppi_bp}; steps=${5:-200}; kexpr=$6
export GLASS_HIP_LIB=$PWD/tools/bin/libglass_trace.so
for rep in 1 2 3; do for v in $v1 $v2; do
  env $var=$v python bench.py --workload $wl --steps $steps --warmup 10 --no-cpu-baseline --no-roofline-hbm --no-pmc 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); c=d['step_breakdown']['calls']
print('$wl $var=$v ms_per_step %.4f' % d['ms_per_step'], {k[6:]: round(c[k]['us'],1) for k in c if 'linear' in k or 'comb' in k})"
done; done
if [ -n "$kexpr" ]; then env $var=$v2 python -m pytest tests -m gpu -q -x --timeout 900 -k "$kexpr" 2>&1 | tail -4; fi
