#!/bin/bash
# usage: ab.sh lib1 lib2 ... : per lib, kernel table of a profiled run + two plain timings
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for lib in "$@"; do
  export GLASS_HIP_LIB=$PWD/$lib
  echo "== $lib"
  timeout 300 bash tools/prof_c2.sh ab_$(basename $lib .so) 2>&1 | head -18 | cut -c1-60,88-140
  for i in 1 2; do timeout 120 python bench.py --steps 400 --warmup 40 --no-cpu-baseline --no-roofline-hbm --no-pmc | python -c "import json,sys;d=json.loads(sys.stdin.read());print('ms_per_step',d['ms_per_step'])"; done
done
