#!/bin/bash
# round-2 evidence of the final kernel: bench lines (driver's command x3, default long run, configs 3 / 4 / 5), rocprofv3 kernel stats of the
# same commands, PMC passes (configs 1 and 5), secondary-kernel stats
set -u
O=gpurun_out/r2z; mkdir -p $O
export TMPDIR=/tmp
for i in 1 2 3; do timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $O/bench_steps20_$i.json 2> $O/bench_steps20.err || exit 1; done
timeout -k 10 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err || exit 1
for c in 3 4 5; do timeout -k 10 400 python bench.py --config $c --steps 800 --no-cpu-baseline > $O/bench_config$c.json 2> $O/bench_config$c.err || exit 1; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r2z/bench_*.json')):
    d=json.load(open(f)); r=d['roofline']; print(f, d['value'], d['ms_per_step'], 'lone', r['lone_frame_ms'], 'frac', r['frac'], 'hbm', r['hbm_frac_traffic'], r.get('hbm_frac_packet_alg'), 'cpu', (d.get('cpu_baseline') or {}).get('value'))
PY
bash tools/pmc_run.sh r2c1 > $O/pmc_c1.log 2>&1; tail -24 $O/pmc_c1.log
bash tools/pmc_run.sh r2c5 --config 5 > $O/pmc_c5.log 2>&1; tail -3 $O/pmc_c5.log
bash tools/pmc_run.sh r2c4 --config 4 > $O/pmc_c4.log 2>&1; tail -3 $O/pmc_c4.log
cp gpurun_out/pmc_r2c1/summary.txt $O/r2_final_pmc_summary.txt; cp gpurun_out/pmc_r2c5/summary.txt $O/r2_final_stress_pmc_summary.txt; cp gpurun_out/pmc_r2c4/summary.txt $O/r2_final_4k_pmc_summary.txt
prof() { # name, command...
  local name=$1; shift
  ( cd /tmp && timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -- "$@" > $GRAFT_REPO_ROOT/$O/prof_$name.log 2>&1 )
  find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/r2_final_${name}_kernel_stats.csv; head -5 $O/r2_final_${name}_kernel_stats.csv; rm -rf $O/prof
}
prof default python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline
prof steps20 python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 20 --warmup 5
prof stress python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --config 5 --steps 400
prof whitted_refl python3 $GRAFT_REPO_ROOT/tools/time_whitted.py atrium 1 refl
prof whitted_1light python3 $GRAFT_REPO_ROOT/tools/time_whitted.py atrium 1
timeout -k 10 300 python tools/time_whitted.py atrium 1 refl 2>&1 | grep "frames in flight\|rays traced" > $O/time_whitted_refl.txt; cat $O/time_whitted_refl.txt
timeout -k 10 300 python tools/time_whitted.py atrium 1 2>&1 | grep "frames in flight\|rays traced" > $O/time_whitted_1.txt; tail -2 $O/time_whitted_1.txt
