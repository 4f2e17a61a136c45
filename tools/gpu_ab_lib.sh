#!/bin/bash
# A/B of two builds of the library on the same box, alternating.  usage: gpu_ab_lib.sh <tag> <altlib relative path> <workload> <steps>
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/$1; alt=$PWD/$2; wl=${3:-ppi_bp}; steps=${4:-200}
mkdir -p $out
for rep in 1 2 3; do
  for v in default alt; do
    if [ $v = alt ]; then export GLASS_HIP_LIB=$alt; else unset GLASS_HIP_LIB; fi
    python bench.py --workload $wl --steps $steps --warmup 10 --no-cpu-baseline --no-roofline-hbm --no-pmc 2>/dev/null > $out/b_$v.json
    python -c "
import json; d=json.load(open('$out/b_$v.json')); c=d['step_breakdown']['calls']
print('$wl $v ms_per_step %.4f' % d['ms_per_step'], {k[6:]: round(c[k]['us'],1) for k in c if 'linear' in k or 'comb' in k})"
  done
done
