#!/bin/bash
set -u
O=gpurun_out/r2o; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1 || { tail -20 $O/pytest_gpu.log; exit 1; }
tail -1 $O/pytest_gpu.log
one() { # label, args
  local label=$1; shift
  timeout -k 10 400 python bench.py --no-cpu-baseline --lone-frames 0 "$@" > $O/x.json 2> $O/b.err || exit 1
  python -c "
import json; d=json.load(open('$O/x.json')); print('$label', d['value'], d['ms_per_step'])"
}
for r in 1 2; do for v in base norecip; do
  if [ $v = base ]; then unset SNAIL_LIB_PATH; else export SNAIL_LIB_PATH=$PWD/snail_amd/exp/lib_$v.so; fi
  one "$v short" --steps 20 --warmup 5
  one "$v long" 
  one "$v c5" --config 5 --steps 800
  one "$v c3" --config 3 --steps 800
done; done
unset SNAIL_LIB_PATH
for v in base norecip; do
  if [ $v = base ]; then unset SNAIL_LIB_PATH; else export SNAIL_LIB_PATH=$PWD/snail_amd/exp/lib_$v.so; fi
  timeout -k 10 300 python tools/time_whitted.py atrium 1 refl 2>&1 | grep "frames in flight 4" | sed "s/^/$v refl /"
done
