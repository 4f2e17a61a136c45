#!/bin/bash
# Round-6 refresh of profiles/: kernel tables + bench lines of C1-C5, default bench lines (counters, CPU baseline, launch-chain
# floor) for EVERY config, the reference-caller line, the evaluation loop, the pre-training step, the per-kernel counter
# summaries (hidden 64, hidden 128, hidden 256), smoke().  Output: gpurun_out/final/ ; copy the r05_* files to profiles/.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
R=${ROUND:-r06}
out=gpurun_out/final
rm -rf $out; mkdir -p $out
for w in ppi_bp density hpo_neuro em_user; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/$w -- python3 bench.py --workload $w --steps 200 --warmup 20 --no-cpu-baseline --no-roofline-hbm --no-pmc --no-floor > $out/${R}_bench_${w}_bench_line.json 2> $out/$w.err
  cp $(ls $out/$w/*/*kernel_stats.csv | head -1) $out/${R}_bench_${w}_kernel_stats.csv
  echo "== $w"; python3 tools/prof_summary.py $out/$w 4 | cut -c1-70,88-140
  rm -rf $out/$w
done
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/c5 -- python3 bench.py --workload powerlaw --steps 20 --warmup 3 --no-cpu-baseline --no-roofline-hbm --no-pmc --no-floor > $out/${R}_bench_powerlaw_bench_line.json 2> $out/c5.err
cp $(ls $out/c5/*/*kernel_stats.csv | head -1) $out/${R}_bench_powerlaw_kernel_stats.csv
echo "== powerlaw"; python3 tools/prof_summary.py $out/c5 4 | cut -c1-70,88-140
rm -rf $out/c5
# default lines: every BASELINE config with cpu_baseline, counter traffic and the launch-chain floor
for w in ppi_bp density hpo_neuro em_user; do
  timeout 900 python3 bench.py --workload $w > $out/${R}_bench_${w}_default_bench_line.json 2> $out/default_$w.err; echo "$w default rc=$?"
  python3 -c "
import json; d=json.load(open('$out/${R}_bench_${w}_default_bench_line.json')); r=d['roofline']; c=d['cpu_baseline']
print('$w ms %.4f value %.3e frac %s (%s) alg_hbm %.2f gather %s traffic %s floor %s cpu %s' % (d['ms_per_step'], d['value'], r['frac'], r['bound'], r['frac_algorithmic_hbm'], r['frac_gather_path'], r['traffic'], d['step_floor'].get('us'), c and round(c['ms_per_step'],1)))"
done
timeout 1500 python3 bench.py --workload powerlaw --steps 10 --warmup 2 > $out/${R}_bench_powerlaw_default_bench_line.json 2> $out/default_pl.err; echo "powerlaw default rc=$?"; cut -c1-300 $out/${R}_bench_powerlaw_default_bench_line.json
timeout 600 python3 bench.py --caller reference --no-cpu-baseline --no-roofline-hbm > $out/${R}_bench_ppi_bp_reference_caller_bench_line.json 2> $out/refcaller.err; echo "reference caller rc=$?"; cut -c1-200 $out/${R}_bench_ppi_bp_reference_caller_bench_line.json
timeout 600 python3 bench.py --mode eval --no-cpu-baseline --no-roofline-hbm --no-pmc > $out/${R}_bench_ppi_bp_eval_bench_line.json 2> $out/eval.err; echo "eval rc=$?"; cut -c1-300 $out/${R}_bench_ppi_bp_eval_bench_line.json
GLASS_BENCH_BACKEND=gloo timeout 600 python3 bench.py --gpus 2 --steps 50 --warmup 5 --no-cpu-baseline --no-roofline-hbm --no-pmc > $out/${R}_bench_gloo2_smoke.json 2> $out/gloo2.err; echo "gloo2 rc=$?"; cut -c1-200 $out/${R}_bench_gloo2_smoke.json
ROUND=$R bash tools/prof_ssl.sh $R > $out/ssl.log 2>&1; head -3 gpurun_out/${R}_ssl_step.txt
mv gpurun_out/${R}_ssl_step* $out/ 2>/dev/null
# counters per kernel of the step (hidden 64 / 128 / 256)
for w in ppi_bp em_user; do ROUND=$R bash tools/step_pmc.sh $w > $out/pmc_$w.log 2>&1; cp gpurun_out/step_pmc/${R}_step_pmc_${w}_summary.csv $out/; done
ROUND=$R STEPS=3 WARM=1 bash tools/step_pmc.sh powerlaw > $out/pmc_powerlaw.log 2>&1; cp gpurun_out/step_pmc/${R}_step_pmc_powerlaw_summary.csv $out/
head -12 $out/${R}_step_pmc_powerlaw_summary.csv | cut -c1-200
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
du -sh $out
