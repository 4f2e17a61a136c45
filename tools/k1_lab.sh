#!/bin/bash
# K1 laboratory run (GPU box): full-row correctness on varied degree shapes, then a timing sweep over
# shapes x cache policy (nt0/nt1 builds) x flat-mode factor.  Output: gpurun_out/k1_lab/{check,perf}.jsonl
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/k1_lab
mkdir -p $out
: > $out/check.jsonl; : > $out/perf.jsonl
B=./tools/bin
for H in 64 32 16 128 17 256 8; do
  for shape in "30000:90000" "calib:100000" "20000:400000:0.8" "ppi_bp" "density-like" "50000:50000" "3000:400000"; do
    for rp in 0 2 100; do
      timeout 120 $B/k1_nt1 $shape $H 3 --rp $rp --full >> $out/check.jsonl 2>&1 || echo "{\"FAIL\": \"$shape H=$H rp=$rp rc=$?\"}" >> $out/check.jsonl
    done
  done
done
grep -c FAIL $out/check.jsonl
for nt in 0 1; do
  for spec in "calib:4000000 64 0" "calib:4000000 64 2" "2000000:3000000 64 0" "2000000:3000000 64 2" "2000000:3000000 64 4" \
              "2000000:6000000 64 0" "2000000:6000000 64 2" "2000000:6000000 64 4" "2000000:12000000 64 0" "2000000:12000000 64 2" \
              "2000000:12000000 64 4" "2000000:12000000 64 8" "ppi_bp 64 2" "ppi_bp 64 0" "hpo_neuro 64 2" "em_user 128 2" "em_user 128 0" \
              "powerlaw 256 2" "powerlaw 64 2" "powerlaw 64 0" "density-like 64 2" "density-like 64 0" "calib:4000000 128 2" "calib:4000000 32 2"; do
    set -- $spec
    echo -n "{\"nt\": $nt, \"r\": " >> $out/perf.jsonl
    timeout 300 $B/k1_nt$nt $1 $2 30 --rp $3 >> $out/perf.jsonl 2>&1 || echo "\"FAIL $spec\"" >> $out/perf.jsonl
    echo "}" >> $out/perf.jsonl
  done
done
cat $out/perf.jsonl | tr -d '\n' | sed 's/}}/}}\n/g' | python3 -c "
import sys, json
for line in sys.stdin:
    line=line.strip()
    if not line: continue
    try:
        d=json.loads(line); r=d['r']
        print(d['nt'], r['shape'], r['H'], 'rp', r['rp'], '%.1f us' % r['us_per_pass'], 'frac %.3f' % r['frac_of_8TBps'], 'err %.1e' % r['spot_rel_err'])
    except Exception as e:
        print('BAD', line[:200])
"
