#!/bin/bash
# round-5 evidence of the current kernels (as tools/history/r4_final.sh), in TWO calls on the GPU box because of gpurun's 20-minute limit:
#   bash tools/r5_final.sh pmc_ieee | pmc_host_sse | pmc   PMC passes (one arithmetic per call fits gpurun's limit) of the five workloads in BOTH arithmetics (configs 1, 4, 5 through bench.py; config 3 and the mirrored
#                                  bounce through tools/whitted_once.py, every kernel); summaries -> gpurun_out/r5z/; then, in the build container:
#                                  cp gpurun_out/r5z/r5_final_*pmc_summary.txt profiles/ && bash tools/r5_final.sh traffic
#   bash tools/r5_final.sh prof    rocprofv3 --kernel-trace --stats of the bench commands (kernel-stats CSVs)
#   bash tools/r5_final.sh bench   bench lines (the driver's command x3, default long run, --arith ieee, orbit and dolly cameras, configs 3 / 3 + bounce /
#                                  4 / 5, the two-rank rehearsal of the plain `--gpus 2` command), rocprofv3 kernel stats of the same commands, soak runs
set -u
O=gpurun_out/r5z; mkdir -p $O
export TMPDIR=/tmp
case "${1:-}" in
pmc|pmc_ieee|pmc_host_sse)
  ARITHS="ieee host_sse"; [ "$1" = pmc_ieee ] && ARITHS=ieee; [ "$1" = pmc_host_sse ] && ARITHS=host_sse
  for a in $ARITHS; do
    s=""; [ $a = host_sse ] && s="_host_sse"
    bash tools/pmc_run.sh r5c1$s --arith $a > $O/pmc_c1$s.log 2>&1; tail -3 $O/pmc_c1$s.log
    bash tools/pmc_run.sh r5c5$s --config 5 --arith $a > $O/pmc_c5$s.log 2>&1; tail -3 $O/pmc_c5$s.log
    bash tools/pmc_run.sh r5c4$s --config 4 --arith $a > $O/pmc_c4$s.log 2>&1; tail -3 $O/pmc_c4$s.log
    bash tools/pmc_cmd.sh r5c3$s python3 $PWD/tools/whitted_once.py atrium 1 norefl $a > $O/pmc_c3$s.log 2>&1; tail -3 $O/pmc_c3$s.log
    bash tools/pmc_cmd.sh r5refl$s python3 $PWD/tools/whitted_once.py atrium 1 refl $a > $O/pmc_refl$s.log 2>&1; tail -3 $O/pmc_refl$s.log
    cp gpurun_out/pmc_r5c1$s/summary.txt $O/r5_final${s}_pmc_summary.txt; cp gpurun_out/pmc_r5c5$s/summary.txt $O/r5_final${s}_stress_pmc_summary.txt
    cp gpurun_out/pmc_r5c4$s/summary.txt $O/r5_final${s}_4k_pmc_summary.txt
    grep "kernel=dev" gpurun_out/pmc_r5c3$s/summary.txt > $O/r5_final${s}_config3_pmc_summary.txt; grep "kernel=dev" gpurun_out/pmc_r5refl$s/summary.txt > $O/r5_final${s}_whitted_refl_pmc_summary.txt
  done
  ;;
traffic)
  python tools/make_traffic.py atrium_1920x1080_n1_c1=profiles/r5_final_pmc_summary.txt atrium_1920x1080_n1_c1_sse=profiles/r5_final_host_sse_pmc_summary.txt \
    stress_1920x1080_n1_c5=profiles/r5_final_stress_pmc_summary.txt stress_1920x1080_n1_c5_sse=profiles/r5_final_host_sse_stress_pmc_summary.txt \
    atrium_3840x2160_n1_c4=profiles/r5_final_4k_pmc_summary.txt atrium_3840x2160_n1_c4_sse=profiles/r5_final_host_sse_4k_pmc_summary.txt \
    atrium_1920x1080_n1_c3=profiles/r5_final_config3_pmc_summary.txt:frames=6 atrium_1920x1080_n1_c3_sse=profiles/r5_final_host_sse_config3_pmc_summary.txt:frames=6 \
    atrium_1920x1080_n1_c3r=profiles/r5_final_whitted_refl_pmc_summary.txt:frames=6 atrium_1920x1080_n1_c3r_sse=profiles/r5_final_host_sse_whitted_refl_pmc_summary.txt:frames=6
  ;;
bench)
  b() { local name=$1; shift; timeout -k 10 400 python bench.py "$@" > $O/bench_$name.json 2> $O/bench_$name.err || { echo "bench $name FAILED"; tail -5 $O/bench_$name.err; }; }
  for i in 1 2 3; do b steps20_$i --steps 20 --warmup 5; done
  b default
  b ieee --no-cpu-baseline --arith ieee
  b orbit --no-cpu-baseline --camera-path orbit
  b dolly --no-cpu-baseline --camera-path dolly
  b config3 --no-cpu-baseline --config 3 --steps 800
  b config3_refl --no-cpu-baseline --config 3 --reflections --steps 300
  b config3_refl_orbit --no-cpu-baseline --config 3 --reflections --steps 300 --camera-path orbit
  b config4 --no-cpu-baseline --config 4 --steps 800
  b config5 --no-cpu-baseline --config 5 --steps 800
  b gloo2 --gpus 2 --backend gloo --steps 200 --warmup 20
  b gloo2_fpl1 --gpus 2 --backend gloo --steps 200 --warmup 20 --frames-per-launch 1
  python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r5z/bench_*.json')):
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][-1]); r=d['roofline']
        print(f.split('/')[-1], d['config']['arith'], d['value'], d['ms_per_step'], 'lone', r.get('lone_frame_ms'), d['config'].get('lone_launch_ms'), 'frac', r['frac'], r.get('frac_of_measured_issue_rate'), 'stale', r.get('counters_stale'), 'hbm', r.get('hbm_frac_traffic'), 'verified', d.get('verified'), 'other', (r.get('other_arith') or {}).get('value'), 'cpu', (d.get('cpu_baseline') or {}).get('value'), 'fpl1', (r.get('one_frame_per_launch') or {}).get('value'))
    except Exception as e: print(f, 'ERR', e)
PY
  ;;
prof)
  prof() { # name, command...
    local name=$1; shift
    ( cd /tmp && timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -- "$@" > $GRAFT_REPO_ROOT/$O/prof_$name.log 2>&1 )
    find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/r5_final_${name}_kernel_stats.csv; head -6 $O/r5_final_${name}_kernel_stats.csv; rm -rf $O/prof
  }
  prof default python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline
  prof steps20 python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 20 --warmup 5
  prof ieee python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --arith ieee
  prof orbit python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --camera-path orbit
  prof dolly python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --camera-path dolly
  prof stress python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --config 5 --steps 400
  prof config3 python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --config 3 --steps 400
  GPU_MAX_HW_QUEUES=16 prof config3_refl python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --config 3 --reflections --steps 200   # (bench.py's own choice for this workload, made before the profiler's library starts HIP)
  ;;
*) echo "usage: bash tools/r5_final.sh pmc_ieee|pmc_host_sse|pmc|traffic|bench|prof"; exit 1;;
esac
