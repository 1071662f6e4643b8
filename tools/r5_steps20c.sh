# the driver's command: frames per launch that divide its 20 steps evenly over the streams (one joint tail) against the default
for r in 1 2 3; do
for c in "" "--frames-per-launch 5" "--frames-per-launch 5 --streams 4" "--frames-per-launch 7 --streams 3" "--frames-per-launch 8 --streams 3" "--frames-per-launch 5 --streams 5" "--frames-per-launch 6 --streams 4" "--frames-per-launch 3 --streams 4" ; do
    timeout -k 10 200 python bench.py --steps 20 --warmup 5 $c --no-cpu-baseline --lone-frames 0 --no-live-check 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$r [$c]', d['value'], d['ms_per_step'], d['verified'])"
done
done
