#!/bin/bash
# round-2 GPU session 4: heavy head of each frame on a high-priority stream
set -u
O=gpurun_out/r2d; mkdir -p $O
export TMPDIR=/tmp
for H in 0 256 512 1024 2048; do
  echo "== split-heavy $H"
  timeout -k 10 200 python tools/timeline.py 20 5 4 0 $H 2>&1 | grep "^rep"
  for i in 1 2; do timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --split-heavy $H > $O/h${H}_s$i.json 2> $O/h${H}_s$i.err || exit 1; done
  timeout -k 10 400 python bench.py --no-cpu-baseline --split-heavy $H > $O/h${H}_long.json 2> $O/h${H}_long.err || exit 1
  timeout -k 10 400 python bench.py --no-cpu-baseline --split-heavy $H --config 5 --steps 800 > $O/h${H}_c5.json 2> $O/h${H}_c5.err || exit 1
  python - $H <<'PY'
import json,sys
H=sys.argv[1]
for k in ('s1','s2','long','c5'):
    d=json.load(open('gpurun_out/r2d/h%s_%s.json'%(H,k))); print(H,k, d['value'], d['ms_per_step'], 'lone', d['roofline']['lone_frame_ms'])
PY
done
timeout -k 10 200 python tools/timeline.py 20 5 4 0 1024 > $O/timeline_h1024.txt 2>&1; tail -21 $O/timeline_h1024.txt
