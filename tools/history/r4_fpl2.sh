set -u
O=gpurun_out/r4fpl2; mkdir -p $O
for c in 5 1 4; do for fpl in 2 4 8; do
  timeout -k 10 300 python bench.py --config $c --steps 1600 --no-cpu-baseline --lone-frames 0 --frames-per-launch $fpl > $O/b_${c}_${fpl}.json 2>$O/err.txt || echo FAIL $c $fpl
done; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4fpl2/b_*.json')):
    d=json.loads([l for l in open(f) if l.startswith('{')][-1]); print(f.split('/')[-1], d['value'], d['ms_per_step'], d['verified'], d['config']['frames_in_flight'])
PY
