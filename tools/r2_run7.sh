#!/bin/bash
# round-2 GPU session 7: A/B-unrolled prefetch loop vs no prefetch, same box; full gpu tests
set -u
O=gpurun_out/r2g; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -4 $O/pytest_gpu.log
for round in 1 2; do
for v in base nopf; do
  if [ $v = base ]; then unset SNAIL_LIB_PATH; else export SNAIL_LIB_PATH=$PWD/snail_amd/exp/lib_$v.so; fi
  timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/${v}_s$round.json 2> $O/${v}_s.err || exit 1
  timeout -k 10 400 python bench.py --no-cpu-baseline > $O/${v}_long$round.json 2> $O/${v}_long.err || exit 1
  timeout -k 10 400 python bench.py --no-cpu-baseline --config 5 --steps 800 > $O/${v}_c5$round.json 2> $O/${v}_c5.err || exit 1
  timeout -k 10 400 python bench.py --no-cpu-baseline --config 3 --steps 800 > $O/${v}_c3$round.json 2> $O/${v}_c3.err || exit 1
  python - $v $round <<'PY'
import json,sys
v,r=sys.argv[1],sys.argv[2]
for k in ('s','long','c5','c3'):
    d=json.load(open('gpurun_out/r2g/%s_%s%s.json'%(v,k,r))); print(v,k, d['value'], d['ms_per_step'], 'lone', d['roofline']['lone_frame_ms'])
PY
done
done
unset SNAIL_LIB_PATH
timeout -k 10 200 python tools/heavy_alone.py 2>&1 | grep "heaviest packets alone" | head -3
