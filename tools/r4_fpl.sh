#!/bin/bash
# the driver's command (--steps 20 --warmup 5) at different frames per launch / launches in flight
set -u
O=gpurun_out/r4fpl; mkdir -p $O
for fpl in 2 4 5 8; do for s in 4 3 2; do
  for i in 1 2; do timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --lone-frames 0 --frames-per-launch $fpl --streams $s > $O/b_${fpl}_${s}_$i.json 2>$O/err.txt || echo FAIL $fpl $s; done
done; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4fpl/b_*.json')):
    d=json.loads([l for l in open(f) if l.startswith('{')][-1]); print(f.split('/')[-1], d['value'], d['ms_per_step'], d['verified'], d['config']['frames_in_flight'])
PY
