#!/bin/bash
# One GPU-box pass: the GPU test-suite, per-kernel dense timings, the default bench line, the C5 bench line.
# Output under gpurun_out/$1/
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/${1:-round}
mkdir -p $out
python -m pytest tests -m gpu -q --timeout 1500 > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
tail -25 $out/pytest.log
python tools/bench_ops.py dual > $out/bench_dual.log 2>&1; tail -8 $out/bench_dual.log
python bench.py --steps 200 --warmup 20 > $out/bench_c2.json 2> $out/bench_c2.err; echo "bench c2 rc=$?"
tail -3 $out/bench_c2.err; cat $out/bench_c2.json
python bench.py --workload powerlaw --steps 20 --warmup 3 --no-cpu-baseline --no-roofline-hbm > $out/bench_c5.json 2> $out/bench_c5.err; echo "bench c5 rc=$?"
tail -3 $out/bench_c5.err; cat $out/bench_c5.json
GLASS_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 50 --warmup 5 --workload em_user --features nodeid --no-cpu-baseline --no-roofline-hbm > $out/bench_gloo2_nodeid.json 2> $out/bench_gloo2_nodeid.err; echo "bench gloo2 nodeid rc=$?"
tail -3 $out/bench_gloo2_nodeid.err; cat $out/bench_gloo2_nodeid.json | cut -c1-1200
