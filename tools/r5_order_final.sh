#!/bin/bash
# Round 5, after the change: the shipped feedback (no priorities by rank; sorted order only for heavy-tailed costs) against no feedback at all, static and turning cameras.
set -u
for r in 1 2; do
for c in "--camera-path static" "--camera-path orbit" "--camera-path dolly" "--config 3 --steps 800" "--config 3 --steps 800 --camera-path orbit" "--config 4 --steps 800" "--config 5 --steps 800" "--config 5 --steps 800 --camera-path orbit" "--config 3 --reflections --steps 300" "--config 3 --reflections --steps 300 --camera-path orbit"; do
  for v in product none; do
    F=""; [ $v = none ] && F="--feedback-order 0"
    timeout -k 10 200 python bench.py $c $F --no-cpu-baseline --lone-frames 0 --no-live-check 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$r $v $c', d['value'], d['ms_per_step'], d['verified'])"
  done
done
done
