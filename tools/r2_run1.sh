#!/bin/bash
# round-2 GPU session 1: tests, the driver's short bench command, variants (waves per block, issue priority), evidence for the secondary kernels
set -u
O=gpurun_out/r2a; mkdir -p $O
export TMPDIR=/tmp
echo "== occupancy" && timeout -k 10 120 python tools/occupancy.py 2>&1 | tail -2 || exit 1
for v in bw2 bw4; do SNAIL_LIB_PATH=$PWD/snail_amd/exp/lib_$v.so timeout -k 10 120 python tools/occupancy.py 2>&1 | tail -1 || exit 1; done
echo "== short bench x3 (driver's command)"
for i in 1 2 3; do timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/base_s$i.json 2> $O/base_s$i.err || exit 1; done
echo "== long bench"
timeout -k 10 400 python bench.py --no-cpu-baseline > $O/base_long.json 2> $O/base_long.err || exit 1
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r2a/base_*.json')):
    d=json.load(open(f)); r=d['roofline']; print(f, d['value'], d['ms_per_step'], 'lone', r['lone_frame_ms'], 'frac', r['frac'], 'hbm', r['hbm_frac_traffic'], r.get('hbm_frac_packet_alg'))
PY
echo "== timeline"
timeout -k 10 200 python tools/timeline.py 20 5 4 > $O/timeline.txt 2>&1 || exit 1
tail -24 $O/timeline.txt
for v in prio priorank; do SNAIL_LIB_PATH=$PWD/snail_amd/exp/lib_$v.so timeout -k 10 200 python tools/timeline.py 20 5 4 2>&1 | grep "^rep" ; done
echo "== variants"
for v in bw2 bw4 prio priorank; do
  export SNAIL_LIB_PATH=$PWD/snail_amd/exp/lib_$v.so
  timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/${v}_s1.json 2> $O/${v}_s1.err || exit 1
  timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/${v}_s2.json 2> $O/${v}_s2.err || exit 1
  timeout -k 10 400 python bench.py --no-cpu-baseline > $O/${v}_long.json 2> $O/${v}_long.err || exit 1
  timeout -k 10 400 python bench.py --no-cpu-baseline --config 5 --steps 800 > $O/${v}_c5.json 2> $O/${v}_c5.err || exit 1
  python - $v <<'PY'
import json,sys
v=sys.argv[1]
for k in ('s1','s2','long','c5'):
    d=json.load(open('gpurun_out/r2a/%s_%s.json'%(v,k))); print(v,k, d['value'], d['ms_per_step'], 'lone', d['roofline']['lone_frame_ms'])
PY
done
unset SNAIL_LIB_PATH
timeout -k 10 400 python bench.py --no-cpu-baseline --config 5 --steps 800 > $O/base_c5.json 2> $O/base_c5.err || exit 1
python -c "
import json; d=json.load(open('gpurun_out/r2a/base_c5.json')); print('base c5', d['value'], d['ms_per_step'], 'lone', d['roofline']['lone_frame_ms'])"
echo "== gpu tests"
timeout -k 10 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -5 $O/pytest_gpu.log
