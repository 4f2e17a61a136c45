#!/bin/bash
# sweep of tools/exact_sweep.py (GPU box): exact accumulators on / off per shape
for cfg in "50000 500000 64 2" "120000 1200000 64 2" "250000 2500000 64 2" "30000 300000 64 2" "12000 120000 128 1" "20000 200000 128 1" "50000 500000 128 1"; do
  for e in 1 0; do GLASS_GN_EXACT=$e timeout 300 python3 tools/exact_sweep.py $cfg 2>&1 | tail -1; done
done
