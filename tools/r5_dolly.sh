for r in 1 2; do
for c in "--camera-path dolly" "--camera-path dolly --frames-per-launch 2" "--camera-path dolly --frames-per-launch 8" "--camera-path orbit" "" ; do
    timeout -k 10 200 python bench.py $c --no-cpu-baseline --lone-frames 0 --no-live-check 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$r [$c]', d['value'], d['ms_per_step'], d['verified'])"
done
done
