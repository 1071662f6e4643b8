#!/bin/bash
# host-SSE arithmetic: parity tests, then its rate beside the default's (two runs each)
set -u
O=gpurun_out/r4sse; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_host_sse.py -x -q -m gpu 2>&1 | tail -3
for i in 1 2; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --lone-frames 0 --arith host_sse > $O/sse_$i.json 2>$O/err.txt || echo FAIL
  timeout -k 10 300 python bench.py --no-cpu-baseline --lone-frames 0 > $O/ieee_$i.json 2>$O/err.txt || echo FAIL
  timeout -k 10 300 python bench.py --no-cpu-baseline --lone-frames 0 --arith host_sse --config 5 --steps 800 > $O/sse_c5_$i.json 2>$O/err.txt || echo FAIL
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4sse/*.json')):
    d=json.loads([l for l in open(f) if l.startswith('{')][-1]); print(f.split('/')[-1], d['value'], d['ms_per_step'], d['verified'])
PY
