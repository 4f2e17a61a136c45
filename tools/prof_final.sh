#!/bin/bash
# End-of-round refresh of profiles/: kernel tables + bench lines of C1-C4 (C5: tools/prof_round.sh), the pre-training step,
# the default bench lines (counters + CPU baseline), smoke().  Output: gpurun_out/final/ ; copy the rNN_* files to profiles/.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
R=${ROUND:-r04}
out=gpurun_out/final
rm -rf $out; mkdir -p $out
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/c2 -- python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-roofline-hbm --no-pmc > $out/${R}_bench_ppi_bp_bench_line.json 2> $out/c2.err
cp $(ls $out/c2/*/*kernel_stats.csv | head -1) $out/${R}_bench_ppi_bp_kernel_stats.csv
python3 tools/prof_summary.py $out/c2 24 | cut -c1-70,88-140
rm -rf $out/c2
for w in density hpo_neuro em_user; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/$w -- python3 bench.py --workload $w --steps 200 --warmup 20 --no-cpu-baseline --no-roofline-hbm --no-pmc > $out/${R}_bench_${w}_bench_line.json 2> $out/$w.err
  cp $(ls $out/$w/*/*kernel_stats.csv | head -1) $out/${R}_bench_${w}_kernel_stats.csv
  echo "== $w"; python3 tools/prof_summary.py $out/$w 3 | cut -c1-70,88-140
  rm -rf $out/$w
done
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/c5 -- python3 bench.py --workload powerlaw --steps 20 --warmup 3 --no-cpu-baseline --no-roofline-hbm --no-pmc > $out/${R}_bench_powerlaw_bench_line.json 2> $out/c5.err
cp $(ls $out/c5/*/*kernel_stats.csv | head -1) $out/${R}_bench_powerlaw_kernel_stats.csv
echo "== powerlaw"; python3 tools/prof_summary.py $out/c5 3 | cut -c1-70,88-140
rm -rf $out/c5
GLASS_BENCH_BACKEND=gloo timeout 600 python3 bench.py --gpus 2 --steps 50 --warmup 5 --no-cpu-baseline --no-roofline-hbm --no-pmc > $out/${R}_bench_gloo2_smoke.json 2> $out/gloo2.err; echo "gloo2 rc=$?"; cut -c1-300 $out/${R}_bench_gloo2_smoke.json
bash tools/prof_ssl.sh $R > $out/ssl.log 2>&1; head -3 gpurun_out/${R}_ssl_step.txt
mv gpurun_out/${R}_ssl_step* $out/
timeout 300 python3 tools/ssl_step.py ppi_bp 100 2 0.5 131072 program >> $out/${R}_ssl_step.txt; tail -1 $out/${R}_ssl_step.txt
timeout 900 python3 bench.py > $out/${R}_bench_ppi_bp_default_bench_line.json 2> $out/default.err; echo "default rc=$?"; cut -c1-400 $out/${R}_bench_ppi_bp_default_bench_line.json
timeout 900 python3 bench.py --workload em_user > $out/${R}_bench_em_user_default_bench_line.json 2> $out/default_em.err; echo "em_user default rc=$?"; cut -c1-300 $out/${R}_bench_em_user_default_bench_line.json
timeout 1200 python3 bench.py --workload powerlaw --steps 10 --warmup 2 > $out/${R}_bench_powerlaw_default_bench_line.json 2> $out/default_pl.err; echo "powerlaw default rc=$?"; cut -c1-300 $out/${R}_bench_powerlaw_default_bench_line.json
timeout 600 python3 bench.py --mode eval --no-cpu-baseline --no-roofline-hbm --no-pmc > $out/${R}_bench_ppi_bp_eval_bench_line.json 2> $out/eval.err; echo "eval rc=$?"; cut -c1-300 $out/${R}_bench_ppi_bp_eval_bench_line.json
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
du -sh $out
