#!/bin/bash
set -u
O=gpurun_out/r2l; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -4 $O/pytest.log
for cfg in "8 4 8 1" "8 4 8 2" "8 4 8 4" "8 4 8 8" "4 4 8 4" "2 4 8 4" "2 4 8 2" "1 4 8 2"; do timeout -k 10 200 python tools/host_rate.py $cfg 2>&1 | grep "^plan"; done
for i in 1 2 3; do timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/s$i.json 2> $O/b.err || exit 1; done
timeout -k 10 400 python bench.py --no-cpu-baseline > $O/long.json 2> $O/b.err || exit 1
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r2l/*.json')):
    d=json.load(open(f)); print(f, d['value'], d['ms_per_step'], d['config']['frames_per_launch'], 'lone', d['roofline']['lone_frame_ms'])
PY
