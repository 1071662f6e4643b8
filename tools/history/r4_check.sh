#!/bin/bash
# round-4 checkpoint on the GPU box: the whole -m gpu suite, then the bench lines the round's changes move.
# usage: bash tools/history/r4_check.sh <tag> [notests]
set -u
T=${1:-a}; O=gpurun_out/r4$T; mkdir -p $O
export TMPDIR=/tmp
if [ "${2:-}" != "notests" ]; then
  timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -5 $O/gpu_tests.log
fi
python tests/golden/full_size.py host_sse $O/host_sse_digests.json > $O/host_sse_digests.log 2>&1; tail -1 $O/host_sse_digests.log | cut -c1-80
b() { local name=$1; shift; timeout -k 10 400 python bench.py "$@" > $O/bench_$name.json 2> $O/bench_$name.err || { echo "bench $name FAILED"; tail -5 $O/bench_$name.err; }; }
b steps20 --steps 20 --warmup 5
b default --no-cpu-baseline
b sse --no-cpu-baseline --arith host_sse
b orbit --no-cpu-baseline --camera-path orbit
b dolly --no-cpu-baseline --camera-path dolly
b c3 --no-cpu-baseline --config 3 --steps 800
b c3_nofb --no-cpu-baseline --config 3 --steps 800 --feedback-order 0
b c3r --no-cpu-baseline --config 3 --reflections --steps 300
b c3r_nofb --no-cpu-baseline --config 3 --reflections --steps 300 --feedback-order 0
b c4 --no-cpu-baseline --config 4 --steps 800
b c5 --no-cpu-baseline --config 5 --steps 800
python - <<PY
import json,glob
for f in sorted(glob.glob('$O/bench_*.json')):
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][-1]); r=d['roofline']
        print(f.split('/')[-1], d['value'], d['ms_per_step'], 'lone', r.get('lone_frame_ms'), 'verified', d.get('verified'), 'other', (r.get('other_arith') or {}).get('value'), 'settle', d['config'].get('settle_frames'), d['config'].get('settle_measured_ms'), 'fpl1', (r.get('one_frame_per_launch') or {}).get('value'))
    except Exception as e: print(f, 'ERR', e)
PY
