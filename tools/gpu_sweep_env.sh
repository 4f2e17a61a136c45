#!/bin/bash
# one bench line per value of an environment switch.  usage: gpu_sweep_env.sh VAR workload steps v1 v2 ...
cd "$GRAFT_REPO_ROOT"
var=$1; wl=$2; steps=$3; shift 3
for v in "$@"; do
  env $var=$v python bench.py --workload $wl --steps $steps --warmup 3 --no-cpu-baseline --no-roofline-hbm --no-pmc 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); c=d['step_breakdown']['calls']
print('$wl $var=$v ms_per_step %.4f' % d['ms_per_step'], {k[6:]: round(c[k]['us'],1) for k in c if 'linear' in k or 'comb' in k})"
done
