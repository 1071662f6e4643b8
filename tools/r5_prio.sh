#!/bin/bash
# Round 5: what the issue priority by dispatch rank (SNAIL_PRIO_RANK: s_setprio 3 / 2 / 1 for the first 1024 / 2048 / 4096 blocks of an ordered launch) is worth when the
# order is exact (static camera) and when it is a stale prediction (turning camera).  Variants: tools/variant.sh prio0 -DSNAIL_PRIO_RANK=0; prio0nat = + -DSNAIL_ORDER_HEAVY_SHIFT=13
set -u
for r in 1 2; do
for c in "--camera-path static" "--camera-path orbit" "--config 3 --steps 800" "--config 3 --steps 800 --camera-path orbit" "--config 5 --steps 800" "--config 5 --steps 800 --camera-path orbit" "--config 3 --reflections --steps 300" "--config 3 --reflections --steps 300 --camera-path orbit"; do
  for v in product prio0 prio0nat none; do
    L="X=1"; F=""
    [ $v = prio0 ] && L="SNAIL_LIB_PATH=$PWD/snail_amd/exp/lib_prio0.so"
    [ $v = prio0nat ] && L="SNAIL_LIB_PATH=$PWD/snail_amd/exp/lib_prio0nat.so"
    [ $v = none ] && F="--feedback-order 0"
    env $L timeout -k 10 200 python bench.py $c $F --no-cpu-baseline --lone-frames 0 --no-live-check 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$r $v $c', d['value'], d['ms_per_step'], d['verified'])"
  done
done
done
