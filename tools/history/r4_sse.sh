#!/bin/bash
# host-SSE arithmetic: its parity tests, then its rate beside the default's for configs 1 / 5 / 3 / 3 + bounce (two runs each) -- the A/B harness of the
# look-up batching (profiles/README.md "Round-4 measurements that did not become code")
set -u
O=gpurun_out/r4sse; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_host_sse.py -x -q -m gpu 2>&1 | tail -3
for i in 1 2; do
  for a in host_sse ieee; do
    timeout -k 10 300 python bench.py --no-cpu-baseline --lone-frames 0 --arith $a > $O/${a}_c1_$i.json 2>$O/err.txt || echo FAIL
    timeout -k 10 300 python bench.py --no-cpu-baseline --lone-frames 0 --arith $a --config 5 --steps 800 > $O/${a}_c5_$i.json 2>$O/err.txt || echo FAIL
    timeout -k 10 300 python bench.py --no-cpu-baseline --lone-frames 0 --arith $a --config 3 --steps 800 > $O/${a}_c3_$i.json 2>$O/err.txt || echo FAIL
    timeout -k 10 300 python bench.py --no-cpu-baseline --lone-frames 0 --arith $a --config 3 --reflections --steps 300 > $O/${a}_c3r_$i.json 2>$O/err.txt || echo FAIL
  done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4sse/*.json')):
    d=json.loads([l for l in open(f) if l.startswith('{')][-1]); print(f.split('/')[-1], d['value'], d['ms_per_step'], d['verified'])
PY
