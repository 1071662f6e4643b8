# the driver's command (--steps 20 --warmup 5) on one box: product (adaptive order) vs always-sorted, frames per launch and streams
for r in 1 2 3; do
for c in "" "--frames-per-launch 4" "--frames-per-launch 8" "--frames-per-launch 4 --streams 5" "--frames-per-launch 1 --streams 8" "--streams 6" "--streams 3 --frames-per-launch 4"; do
  for v in product adapt0; do
    L="X=1"; [ $v != product ] && L="SNAIL_LIB_PATH=$PWD/snail_amd/exp/lib_$v.so"
    env $L timeout -k 10 200 python bench.py --steps 20 --warmup 5 $c --no-cpu-baseline --lone-frames 0 --no-live-check 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$r $v [$c]', d['value'], d['ms_per_step'], d['verified'])"
  done
done
done
