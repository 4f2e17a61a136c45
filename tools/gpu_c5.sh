#!/bin/bash
# config 5 iteration: the parity / race tests that cover hidden 256 / 512, then the dense breakdown of the bench line
cd "$GRAFT_REPO_ROOT"
python -m pytest tests -m gpu -q -x --timeout 1200 -k "${1:-repeat_bitwise or dual_linear_mix_fused or both_product_forms or c5_family or tiled_product_forms or two_threads}" 2>&1 | tail -4
python bench.py --workload powerlaw --steps 10 --warmup 2 --min-blocks 5 --no-cpu-baseline --no-roofline-hbm --no-pmc --no-floor 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); c=d['step_breakdown']['calls']
print('powerlaw ms_per_step %.3f' % d['ms_per_step'], {k[6:]: round(c[k]['us'],0) for k in c if c[k]['us'] > 100})"
