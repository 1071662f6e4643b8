#!/bin/bash
# round-3 evidence of the current kernels, in ONE run on the GPU box:
#   1. PMC passes (configs 1, 4, 5 through bench.py; config 3 and the mirrored bounce through tools/whitted_once.py, every kernel)
#   2. profiles/traffic.json regenerated from them on the box (so that every bench line below is priced with THIS kernel's counters:
#      `counters_stale` false); the summaries and the json come back under gpurun_out/r3z/ and are copied into profiles/ by hand
#   3. bench lines: the driver's command x3 with and without the settle period, default long run, moving camera, configs 3 / 4 / 5, the two-rank
#      rehearsal of the plain `--gpus 2` command
#   4. rocprofv3 kernel stats of the same commands; the mirrored-bounce frame times
set -u
O=gpurun_out/r3z; mkdir -p $O
export TMPDIR=/tmp
bash tools/pmc_run.sh r3c1 > $O/pmc_c1.log 2>&1; tail -24 $O/pmc_c1.log
bash tools/pmc_run.sh r3c5 --config 5 > $O/pmc_c5.log 2>&1; tail -3 $O/pmc_c5.log
bash tools/pmc_run.sh r3c4 --config 4 > $O/pmc_c4.log 2>&1; tail -3 $O/pmc_c4.log
bash tools/pmc_cmd.sh r3c3 python3 $PWD/tools/whitted_once.py atrium 1 > $O/pmc_c3.log 2>&1; tail -3 $O/pmc_c3.log
bash tools/pmc_cmd.sh r3refl python3 $PWD/tools/whitted_once.py atrium 1 refl > $O/pmc_refl.log 2>&1; tail -3 $O/pmc_refl.log
cp gpurun_out/pmc_r3c1/summary.txt $O/r3_final_pmc_summary.txt; cp gpurun_out/pmc_r3c5/summary.txt $O/r3_final_stress_pmc_summary.txt; cp gpurun_out/pmc_r3c4/summary.txt $O/r3_final_4k_pmc_summary.txt
grep "kernel=dev::" gpurun_out/pmc_r3c3/summary.txt > $O/r3_final_config3_pmc_summary.txt; grep "kernel=dev::" gpurun_out/pmc_r3refl/summary.txt > $O/r3_final_whitted_refl_pmc_summary.txt
cp $O/r3_final_pmc_summary.txt $O/r3_final_stress_pmc_summary.txt $O/r3_final_4k_pmc_summary.txt $O/r3_final_config3_pmc_summary.txt $O/r3_final_whitted_refl_pmc_summary.txt profiles/
python tools/make_traffic.py atrium_1920x1080_n1_c1=profiles/r3_final_pmc_summary.txt stress_1920x1080_n1_c5=profiles/r3_final_stress_pmc_summary.txt \
  atrium_3840x2160_n1_c4=profiles/r3_final_4k_pmc_summary.txt atrium_1920x1080_n1_c3=profiles/r3_final_config3_pmc_summary.txt:frames=6 \
  atrium_1920x1080_whitted_refl=profiles/r3_final_whitted_refl_pmc_summary.txt:frames=6 || exit 1
cp profiles/traffic.json $O/traffic.json
for i in 1 2 3; do timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $O/bench_steps20_$i.json 2> $O/bench_steps20.err || exit 1; done
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --settle-ms 0 --no-cpu-baseline > $O/bench_steps20_nosettle.json 2>> $O/bench_steps20.err || exit 1
timeout -k 10 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err || exit 1
timeout -k 10 400 python bench.py --camera-path orbit --no-cpu-baseline > $O/bench_orbit.json 2>> $O/bench_default.err || exit 1
for c in 3 4 5; do timeout -k 10 400 python bench.py --config $c --steps 800 --no-cpu-baseline > $O/bench_config$c.json 2> $O/bench_config$c.err || exit 1; done
timeout -k 10 400 python bench.py --gpus 2 --backend gloo --steps 200 --warmup 20 > $O/bench_gloo2.json 2> $O/bench_gloo2.err || exit 1
timeout -k 10 400 python bench.py --gpus 2 --backend gloo --steps 200 --warmup 20 --frames-per-launch 1 > $O/bench_gloo2_fpl1.json 2>> $O/bench_gloo2.err || exit 1
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3z/bench_*.json')):
    d=json.loads([l for l in open(f) if l.startswith('{')][-1]); r=d['roofline']
    print(f.split('/')[-1], d['value'], d['ms_per_step'], 'lone', r.get('lone_frame_ms'), d['config'].get('lone_launch_ms'), 'frac', r['frac'], r.get('frac_of_measured_issue_rate'), 'stale', r.get('counters_stale'), 'hbm', r.get('hbm_frac_traffic'), 'cpu', (d.get('cpu_baseline') or {}).get('value'), (d.get('cpu_baseline') or {}).get('scalar_port_value'), 'fpl1', (r.get('one_frame_per_launch') or {}).get('value'))
PY
prof() { # name, command...
  local name=$1; shift
  ( cd /tmp && timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -- "$@" > $GRAFT_REPO_ROOT/$O/prof_$name.log 2>&1 )
  find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/r3_final_${name}_kernel_stats.csv; head -5 $O/r3_final_${name}_kernel_stats.csv; rm -rf $O/prof
}
prof default python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline
prof steps20 python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 20 --warmup 5
prof orbit python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --camera-path orbit
prof stress python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --config 5 --steps 400
prof config3 python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --config 3 --steps 400
prof whitted_refl python3 $GRAFT_REPO_ROOT/tools/time_whitted.py atrium 1 refl
timeout -k 10 300 python tools/time_whitted.py atrium 1 refl 2>&1 | grep "frames in flight\|rays traced" > $O/r3_final_time_whitted_refl.txt; cat $O/r3_final_time_whitted_refl.txt
