cd "$GRAFT_REPO_ROOT"; out=gpurun_out/r02i; mkdir -p $out
for spec in "ppi_bp 64" "hpo_neuro 64" "em_user 128" "density-like 64" "powerlaw 64" "powerlaw 256" "calib:4000000 64" "2000000:6000000 64" "30000:90000 64"; do set -- $spec; ./tools/bin/spmm_bench $1 $2 100 --full 2>&1 | python3 -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l); print(d['shape'],d['H'],'rp',d['rp'],'%.2f us'%d['us_per_pass'],'frac %.3f'%d['frac_of_8TBps'],'err %.1e'%d['spot_rel_err'])
    except Exception: print(l.strip()[:200])
"; done
python -m pytest tests -m gpu -q --timeout 1500 > $out/pytest.log 2>&1; tail -4 $out/pytest.log
python bench.py --no-cpu-baseline > $out/bench_c2.json 2> $out/c2.err; python3 -c "
import json; d=json.load(open('$out/bench_c2.json')); print('C2', d['ms_per_step'], d['value'], d['roofline']['avg_launch_us'], d['roofline']['back_to_back_us'], d['roofline']['frac'], [round(e['frac'],3) for e in d['roofline_hbm']])"
