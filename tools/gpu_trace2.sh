#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for e in "GLASS_GN_EXACT_FWD=1" "GLASS_GN_EXACT_FWD=0"; do
echo "=== $e"; env $e GLASS_HIP_LIB=$PWD/tools/bin/libglass_trace.so python tools/dense_trace.py ppi_bp 2>&1 | grep -A14 "comb fwd\|trans fwd" | grep -v "^--\|wgrad waves"
done
