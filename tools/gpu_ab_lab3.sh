#!/bin/bash
# A/B/C of a laboratory knob over several values on a bench workload, alternating on one box.
# usage: gpu_ab_lab3.sh VAR "v1 v2 v3" workload steps
cd "$GRAFT_REPO_ROOT"
var=$1; vals=$2; wl=${3:-ppi_bp}; steps=${4:-200}
export GLASS_HIP_LIB=$PWD/tools/bin/libglass_trace.so
for rep in 1 2; do for v in $vals; do
  env $var=$v python bench.py --workload $wl --steps $steps --warmup 10 --no-cpu-baseline --no-roofline-hbm --no-pmc 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); c=d['step_breakdown']['calls']
print('$wl $var=$v ms_per_step %.4f' % d['ms_per_step'], {k[6:]: round(c[k]['us'],1) for k in c if 'linear' in k or 'comb' in k})"
done; done
