#!/bin/bash
set -u
O=gpurun_out/r2k; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -5 $O/pytest.log
for B in 1 2 4 5 8; do
  for i in 1 2; do timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --frames-per-launch $B > $O/b${B}_s$i.json 2> $O/b.err || exit 1; done
  timeout -k 10 400 python bench.py --no-cpu-baseline --frames-per-launch $B > $O/b${B}_long.json 2> $O/b.err || exit 1
  timeout -k 10 400 python bench.py --no-cpu-baseline --frames-per-launch $B --config 5 --steps 800 > $O/b${B}_c5.json 2> $O/b.err || exit 1
  python - $B <<'PY'
import json,sys
B=sys.argv[1]
for k in ('s1','s2','long','c5'):
    d=json.load(open('gpurun_out/r2k/b%s_%s.json'%(B,k))); print(B,k, d['value'], d['ms_per_step'], 'lone', d['roofline']['lone_frame_ms'])
PY
done
