# the exact-order hint: the driver's command and the long runs, product vs always-sorted, two stream / frames-per-launch shapes
for r in 1 2 3; do
for c in "--steps 20 --warmup 5" "--steps 20 --warmup 5 --streams 3 --frames-per-launch 4" "" "--streams 3 --frames-per-launch 4" "--camera-path orbit" "--camera-path orbit --streams 3 --frames-per-launch 4" "--config 3 --steps 800" "--config 3 --steps 800 --camera-path orbit"; do
  for v in product adapt0; do
    L="X=1"; [ $v != product ] && L="SNAIL_LIB_PATH=$PWD/snail_amd/exp/lib_$v.so"
    env $L timeout -k 10 200 python bench.py $c --no-cpu-baseline --lone-frames 0 --no-live-check 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$r $v [$c]', d['value'], d['ms_per_step'], d['verified'])"
  done
done
done
