#!/bin/bash
# round-2 GPU session 5: bench with diagnostics in front of the timed region; evidence for the secondary kernels; per-ray cull A/B
set -u
O=gpurun_out/r2e; mkdir -p $O
export TMPDIR=/tmp
for i in 1 2 3; do timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/s$i.json 2> $O/s$i.err || exit 1; done
timeout -k 10 400 python bench.py --no-cpu-baseline > $O/long.json 2> $O/long.err || exit 1
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r2e/*.json')):
    d=json.load(open(f)); print(f, d['value'], d['ms_per_step'], 'lone', d['roofline']['lone_frame_ms'])
PY
echo "== whitted refl: cull vs nocull"
for v in base nocull; do
  if [ $v = base ]; then unset SNAIL_LIB_PATH; else export SNAIL_LIB_PATH=$PWD/snail_amd/exp/lib_$v.so; fi
  timeout -k 10 300 python tools/time_whitted.py atrium 1 refl 2>&1 | grep "frames in flight 4\|rays traced" | tail -3
done
unset SNAIL_LIB_PATH
echo "== kernel-trace: whitted refl"
cd /tmp && timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_refl -- python3 $GRAFT_REPO_ROOT/tools/time_whitted.py atrium 1 refl > $GRAFT_REPO_ROOT/$O/prof_refl.log 2>&1; cd $GRAFT_REPO_ROOT
find $O/prof_refl -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/r2_whitted_refl_kernel_stats.csv; head -12 $O/r2_whitted_refl_kernel_stats.csv
echo "== kernel-trace: whitted 1 light"
cd /tmp && timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_w1 -- python3 $GRAFT_REPO_ROOT/tools/time_whitted.py atrium 1 > $GRAFT_REPO_ROOT/$O/prof_w1.log 2>&1; cd $GRAFT_REPO_ROOT
find $O/prof_w1 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/r2_whitted_1light_kernel_stats.csv; head -8 $O/r2_whitted_1light_kernel_stats.csv
echo "== kernel-trace: stress (config 5)"
cd /tmp && timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_c5 -- python3 $GRAFT_REPO_ROOT/bench.py --config 5 --steps 400 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/prof_c5.log 2>&1; cd $GRAFT_REPO_ROOT
find $O/prof_c5 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/r2_stress_kernel_stats.csv; head -6 $O/r2_stress_kernel_stats.csv
rm -rf $O/prof_refl $O/prof_w1 $O/prof_c5
