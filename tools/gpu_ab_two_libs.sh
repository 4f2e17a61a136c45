#!/bin/bash
# A/B of two laboratory builds (paths relative to the repo root) on a bench workload, alternating on one box.
# usage: gpu_ab_two_libs.sh <libA> <libB> [workload] [steps]
cd "$GRAFT_REPO_ROOT"
a=$PWD/$1; b=$PWD/$2; wl=${3:-ppi_bp}; steps=${4:-200}
for rep in 1 2 3; do for v in a b; do
  if [ $v = a ]; then export GLASS_HIP_LIB=$a; else export GLASS_HIP_LIB=$b; fi
  python bench.py --workload $wl --steps $steps --warmup 10 --no-cpu-baseline --no-roofline-hbm --no-pmc 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); c=d['step_breakdown']['calls']
print('$wl lib=$v ms_per_step %.4f' % d['ms_per_step'], {k[6:]: round(c[k]['us'],1) for k in c if 'linear' in k or 'comb' in k})"
done; done
