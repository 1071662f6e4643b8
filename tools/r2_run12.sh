#!/bin/bash
set -u
O=gpurun_out/r2n; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1 || { tail -20 $O/pytest_gpu.log; exit 1; }
tail -1 $O/pytest_gpu.log
for r in 1 2; do for b in 2 3 4 5 8; do
timeout -k 10 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --frames-per-launch $b --lone-frames 0 > $O/x.json 2> $O/b.err || exit 1
python -c "
import json; d=json.load(open('$O/x.json')); print('fpl $b', d['value'], d['ms_per_step'])"
done; done
for b in 2 4 8; do
timeout -k 10 300 python bench.py --no-cpu-baseline --frames-per-launch $b --lone-frames 0 > $O/x.json 2> $O/b.err || exit 1
python -c "
import json; d=json.load(open('$O/x.json')); print('long fpl $b', d['value'], d['ms_per_step'])"
done
