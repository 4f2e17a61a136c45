#!/bin/bash
# config 4 iteration: the parity / race tests that cover hidden 128 at N = 50 000, then the bench line's dense breakdown twice
cd "$GRAFT_REPO_ROOT"
python -m pytest tests -m gpu -q -x --timeout 900 -k "${1:-repeat_bitwise or full_size_vs_oracle or dual_linear_mix_fused or both_product_forms or dropout_step}" 2>&1 | tail -4
for i in 1 2; do python bench.py --workload em_user --steps 200 --warmup 20 --no-cpu-baseline --no-roofline-hbm --no-pmc --no-floor 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); c=d['step_breakdown']['calls']
print('em_user ms_per_step %.4f' % d['ms_per_step'], {k[6:]: round(c[k]['us'],1) for k in c})"; done
