#!/bin/bash
# round-2 GPU session 2: the record-prefetching node loop against the plain one (same box)
set -u
O=gpurun_out/r2b; mkdir -p $O
export TMPDIR=/tmp
echo "== gpu tests (product = prefetch loop)"
timeout -k 10 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; rc=$?; tail -6 $O/pytest_gpu.log
timeout -k 10 600 python tools/sse_counts.py --write > $O/sse_counts.txt 2>&1; tail -3 $O/sse_counts.txt
for v in base nopf pfrank; do
  if [ $v = base ]; then unset SNAIL_LIB_PATH; else export SNAIL_LIB_PATH=$PWD/snail_amd/exp/lib_$v.so; fi
  echo "== $v"
  timeout -k 10 200 python tools/heavy_alone.py 2>&1 | grep "heaviest packets alone\|lightest" | head -12
  timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/${v}_s1.json 2> $O/${v}_s1.err || exit 1
  timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/${v}_s2.json 2> $O/${v}_s2.err || exit 1
  timeout -k 10 400 python bench.py --no-cpu-baseline > $O/${v}_long.json 2> $O/${v}_long.err || exit 1
  timeout -k 10 400 python bench.py --no-cpu-baseline --config 5 --steps 800 > $O/${v}_c5.json 2> $O/${v}_c5.err || exit 1
  timeout -k 10 400 python bench.py --no-cpu-baseline --config 3 --steps 800 > $O/${v}_c3.json 2> $O/${v}_c3.err || exit 1
  python - $v <<'PY'
import json,sys
v=sys.argv[1]
for k in ('s1','s2','long','c5','c3'):
    d=json.load(open('gpurun_out/r2b/%s_%s.json'%(v,k))); print(v,k, d['value'], d['ms_per_step'], 'lone', d['roofline']['lone_frame_ms'])
PY
done
