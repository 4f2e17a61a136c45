#!/bin/bash
# one iteration of the dense-kernel work: targeted parity tests, the phase trace, the default bench line twice.
cd "$GRAFT_REPO_ROOT"
python -m pytest tests -m gpu -q -x --timeout 900 -k "${1:-dual_linear or comb or stack_program or step_program_full or bitwise or dropout_step or pair_program}" 2>&1 | tail -4
GLASS_HIP_LIB=$PWD/tools/bin/libglass_trace.so python tools/dense_trace.py ppi_bp 2>&1 | grep -v "^/opt"
for i in 1 2; do python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-roofline-hbm --no-pmc 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); c=d['step_breakdown']['calls']
print('ms_per_step %.4f' % d['ms_per_step'], {k[6:]: round(c[k]['us'],1) for k in c})"; done
