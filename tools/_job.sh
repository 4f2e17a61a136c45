cd "$GRAFT_REPO_ROOT"; out=gpurun_out/r02g; mkdir -p $out
python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "dual_linear or dense_pack" > $out/pytest_k.log 2>&1; tail -4 $out/pytest_k.log
python -m pytest tests/test_gpu_model.py -m gpu -q -k "stack_program or step_program or fused_readout or randomised" --timeout 1500 > $out/pytest_m.log 2>&1; tail -4 $out/pytest_m.log
python tools/bench_ops.py dual > $out/bench_dual.log 2>&1; tail -7 $out/bench_dual.log
for wl in em_user; do python bench.py --workload $wl --steps 200 --warmup 20 --no-cpu-baseline --no-roofline-hbm > $out/bench_$wl.json 2> $out/$wl.err; python3 -c "
import json; d=json.load(open('$out/bench_$wl.json')); print('$wl', d['ms_per_step'], d['value'], {k:v['us'] for k,v in list(d['step_breakdown']['calls'].items())[:6]})"; done
