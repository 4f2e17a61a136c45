#!/bin/bash
# K1 laboratory run (GPU box): timing sweep over the lab builds tools/bin/k1_u_<name> (compile-time variants of
# spmm.hip built by `make -C tools lab LABFLAGS=...`) on the HBM-bound and the cache-resident shapes, every row checked
# in fp64 (--full).  Output: gpurun_out/k1_lab/perf2.jsonl
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/k1_lab
mkdir -p $out
: > $out/perf2.jsonl
for b in tools/bin/k1_u*; do
  for spec in "calib:4000000 64" "2000000:6000000 64" "ppi_bp 64" "hpo_neuro 64" "hpo_neuro 128" "em_user 128" "powerlaw 64" "powerlaw 256" "density-like 64"; do
    set -- $spec
    echo -n "{\"build\": \"$(basename $b)\", \"r\": " >> $out/perf2.jsonl
    timeout 300 $b $1 $2 30 --full >> $out/perf2.jsonl 2>&1 || echo "\"FAIL $spec\"" >> $out/perf2.jsonl
    echo "}" >> $out/perf2.jsonl
  done
done
cat $out/perf2.jsonl | tr -d '\n' | sed 's/}}/}}\n/g' | python3 -c "
import sys, json
for line in sys.stdin:
    line=line.strip()
    if not line: continue
    try:
        d=json.loads(line); r=d['r']
        print(d['build'], r['shape'], r['H'], '%.1f us' % r['us_per_pass'], 'frac %.3f' % r['frac_of_8TBps'], 'err %.1e' % r['spot_rel_err'])
    except Exception as e:
        print('BAD', line[:200])
"
