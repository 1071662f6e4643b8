for r in 1 2; do
for c in "--camera-path static" "--camera-path orbit" "--camera-path dolly" "--config 3 --steps 800" "--config 3 --steps 800 --camera-path orbit" "--config 4 --steps 800"; do
  for v in sorted natural none; do
    if [ $v = natural ]; then L="SNAIL_LIB_PATH=$PWD/snail_amd/exp/lib_heavy13.so"; else L="X=1"; fi
    if [ $v = none ]; then F="--feedback-order 0"; else F=""; fi
    env $L timeout -k 10 200 python bench.py $c $F --no-cpu-baseline --lone-frames 0 --no-live-check 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$r $v $c', d['value'], d['ms_per_step'], d['verified'])"
  done
done
done
