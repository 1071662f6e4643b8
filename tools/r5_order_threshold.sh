for r in 1 2; do
for c in "--config 3 --reflections --steps 300" "--config 3 --reflections --steps 300 --camera-path orbit" "--config 3 --steps 800" "--config 3 --steps 800 --camera-path orbit" "--camera-path orbit" "--camera-path static" "--config 4 --steps 800"; do
  for v in product adapt3 adapt2; do
    L="X=1"; [ $v != product ] && L="SNAIL_LIB_PATH=$PWD/snail_amd/exp/lib_$v.so"
    env $L timeout -k 10 200 python bench.py $c --no-cpu-baseline --lone-frames 0 --no-live-check 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$r $v $c', d['value'], d['ms_per_step'], d['verified'])"
  done
done
done
