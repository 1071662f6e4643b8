# frames in flight: streams x frames per launch, the driver's command and the long run (one box)
for r in 1 2; do
for sh in "4 2" "3 4" "2 8" "3 8" "2 4" "3 3" "3 6" "4 4" "2 6"; do
  set -- $sh
  for c in "--steps 20 --warmup 5" "" "--config 5 --steps 800" "--config 4 --steps 800"; do
    timeout -k 10 200 python bench.py $c --streams $1 --frames-per-launch $2 --no-cpu-baseline --lone-frames 0 --no-live-check 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$r streams $1 fpl $2 [$c]', d['value'], d['ms_per_step'], d['verified'])"
  done
done
for s in 2 3 4 5 6; do
  for c in "--config 3 --steps 800" "--config 3 --reflections --steps 300"; do
    timeout -k 10 200 python bench.py $c --streams $s --no-cpu-baseline --lone-frames 0 --no-live-check 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$r streams $s [$c]', d['value'], d['ms_per_step'], d['verified'])"
  done
done
done
