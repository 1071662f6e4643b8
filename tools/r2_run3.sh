#!/bin/bash
# round-2 GPU session 3: de-phased start (stagger), any-order launches, host issue rate of the multi-GPU route
set -u
O=gpurun_out/r2c; mkdir -p $O
export TMPDIR=/tmp
echo "== anyorder" && timeout -k 10 200 python tools/anyorder.py 2>&1 | grep "^flags" | tail -8
echo "== timeline stagger 1 / 0"
timeout -k 10 200 python tools/timeline.py 20 5 4 1 > $O/timeline_stag1.txt 2>&1 || exit 1; grep "^rep" $O/timeline_stag1.txt; tail -20 $O/timeline_stag1.txt
timeout -k 10 200 python tools/timeline.py 20 5 4 0 2>&1 | grep "^rep"
for st in 1 0; do
  for i in 1 2 3; do timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --stagger $st > $O/s${st}_$i.json 2> $O/s${st}_$i.err || exit 1; done
  timeout -k 10 400 python bench.py --no-cpu-baseline --stagger $st > $O/s${st}_long.json 2> $O/s${st}_long.err || exit 1
  timeout -k 10 400 python bench.py --no-cpu-baseline --stagger $st --steps 100 --warmup 20 > $O/s${st}_100.json 2> $O/s${st}_100.err || exit 1
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r2c/s*.json')):
    d=json.load(open(f)); print(f, d['value'], d['ms_per_step'], 'lone', d['roofline']['lone_frame_ms'])
PY
echo "== host rate of one share of an N-rank plan"
for cfg in "8 4" "8 6" "8 8" "8 12" "4 4" "4 8" "2 4" "1 4"; do timeout -k 10 200 python tools/host_rate.py $cfg 2>&1 | grep "^plan"; done
echo "== gpu tests"
timeout -k 10 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -4 $O/pytest_gpu.log
