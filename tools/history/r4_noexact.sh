set -u
O=gpurun_out/r4noexact; mkdir -p $O
export SNAIL_LIB_PATH=$PWD/snail_amd/libsnailhip_debug.so
for i in 1 2; do
  for v in 0 1; do
    SNAIL_DEBUG_NO_EXACT_PASS=$v timeout -k 10 300 python bench.py --no-cpu-baseline --lone-frames 0 > $O/long_${v}_$i.json 2>$O/err.txt || echo FAIL
    SNAIL_DEBUG_NO_EXACT_PASS=$v timeout -k 10 300 python bench.py --no-cpu-baseline --lone-frames 0 --steps 20 --warmup 5 > $O/short_${v}_$i.json 2>$O/err.txt || echo FAIL
    SNAIL_DEBUG_NO_EXACT_PASS=$v timeout -k 10 300 python bench.py --no-cpu-baseline --lone-frames 0 --config 5 --steps 800 > $O/c5_${v}_$i.json 2>$O/err.txt || echo FAIL
  done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4noexact/*.json')):
    d=json.loads([l for l in open(f) if l.startswith('{')][-1]); print(f.split('/')[-1], d['value'], d['ms_per_step'], d['verified'])
PY
