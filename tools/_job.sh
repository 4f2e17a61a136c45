cd "$GRAFT_REPO_ROOT"; out=gpurun_out/r02j; mkdir -p $out
python -m pytest tests -m gpu -q --timeout 1500 > $out/pytest.log 2>&1; tail -4 $out/pytest.log
python bench.py > $out/bench_default.json 2> $out/default.err; python3 -c "
import json; d=json.load(open('$out/bench_default.json')); print('default', d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['avg_launch_us'], d['roofline']['traffic'], [round(e['frac'],3) for e in d['roofline_hbm']], d['cpu_baseline']['value'])"
for wl in hpo_neuro em_user powerlaw; do st=200; [ $wl = powerlaw ] && st=20; python bench.py --workload $wl --steps $st --warmup 5 --no-cpu-baseline --no-roofline-hbm > $out/bench_$wl.json 2> $out/$wl.err; python3 -c "
import json; d=json.load(open('$out/bench_$wl.json')); print('$wl', d['ms_per_step'], d['value'], d['roofline']['bound'], round(d['roofline']['frac'],3), d['step_breakdown'].get('dense_mfma',{}).get('frac'))"; done
ROUND=r02 bash tools/prof_round.sh > $out/prof_round.log 2>&1; tail -5 $out/prof_round.log | cut -c1-200
