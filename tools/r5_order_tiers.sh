#!/bin/bash
# Round 5: the first tier of a derived dispatch order (SNAIL_ORDER_HEAVY_SHIFT: the heaviest n >> shift slots first, the rest in the built-in order; 0 = the
# fully sorted order of rounds 2-4).  Build host: for v in 0 2 3 4; do tools/variant.sh heavy$v -DSNAIL_ORDER_HEAVY_SHIFT=$v; done
set -u
for r in 1 2; do
for v in heavy0 heavy2 heavy3 heavy4; do
  for c in "--camera-path static" "--camera-path orbit" "--config 5 --steps 800" "--config 5 --steps 800 --camera-path orbit" "--config 3 --reflections --steps 300" "--config 3 --steps 800 --camera-path orbit"; do
    SNAIL_LIB_PATH=$PWD/snail_amd/exp/lib_$v.so timeout -k 10 200 python bench.py $c --no-cpu-baseline --lone-frames 0 --no-live-check 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$r $v $c', d['value'], d['ms_per_step'], d['verified'])"
  done
done
done
