#!/bin/bash
# dense-kernel pass on the GPU box: targeted tests, per-kernel timings, the C5 bench line.  gpurun_out/$1/
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/${1:-dense}
mkdir -p $out
python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "dual_linear or dense_pack or spmm" > $out/pytest_k.log 2>&1; echo "pytest kernels rc=$?"; tail -15 $out/pytest_k.log
python -m pytest tests/test_gpu_model.py -m gpu -q -k "stack_program_vs_oracle or per_op_path_with_dropout or c5 or detached or refused" --timeout 1500 > $out/pytest_m.log 2>&1; echo "pytest model rc=$?"; tail -15 $out/pytest_m.log
python tools/bench_ops.py dual > $out/bench_dual.log 2>&1; cat $out/bench_dual.log | tail -12
python bench.py --workload powerlaw --steps 20 --warmup 3 --no-cpu-baseline --no-roofline-hbm > $out/bench_c5.json 2> $out/bench_c5.err; echo "bench c5 rc=$?"
tail -3 $out/bench_c5.err; cat $out/bench_c5.json
