cd "$GRAFT_REPO_ROOT"; out=gpurun_out/r02e; mkdir -p $out
python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "wgrad or dual_linear" > $out/pytest_k.log 2>&1; tail -3 $out/pytest_k.log
python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-roofline-hbm > $out/bench_c2.json 2> $out/c2.err; python3 -c "
import json; d=json.load(open('$out/bench_c2.json')); print('C2', d['ms_per_step'], {k:v['us'] for k,v in list(d['step_breakdown']['calls'].items())[:6]})"
for wl in hpo_neuro em_user; do python bench.py --workload $wl --steps 200 --warmup 20 --no-cpu-baseline --no-roofline-hbm > $out/bench_$wl.json 2> $out/$wl.err; python3 -c "
import json; d=json.load(open('$out/bench_$wl.json')); print('$wl', d['ms_per_step'], d['value'], d['roofline']['frac'], {k:v['us'] for k,v in list(d['step_breakdown']['calls'].items())[:6]})"; done
bash tools/k1_lab.sh > $out/k1_lab2.log 2>&1; tail -55 $out/k1_lab2.log
