#!/bin/bash
# the reference caller on the measured path: its tests, then the default bench line beside the --caller reference one
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/refcaller; mkdir -p $out
python -m pytest tests/test_gpu_reference_caller.py -m gpu -q -x --timeout 900 2>&1 | tail -25
for c in step reference; do
  python bench.py --caller $c --no-cpu-baseline --no-roofline-hbm --no-pmc > $out/bench_$c.json 2> $out/bench_$c.err || tail -5 $out/bench_$c.err
  python -c "
import json,sys; d=json.load(open('$out/bench_$c.json')); print('$c', 'ms_per_step %.4f (min %.4f max %.4f) blocks %d' % (d['ms_per_step'], d['ms_per_step_min'], d['ms_per_step_max'], d['blocks']), 'hip_graph', d['config']['hip_graph'])"
done
