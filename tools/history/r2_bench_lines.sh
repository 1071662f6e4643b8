#!/bin/bash
# the bench lines of tools/history/r2_final.sh alone (after profiles/traffic.json has been refreshed from that run's PMC passes)
set -u
O=gpurun_out/r2z; mkdir -p $O
export TMPDIR=/tmp
for i in 1 2 3; do timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $O/bench_steps20_$i.json 2> $O/bench_steps20.err || exit 1; done
timeout -k 10 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err || exit 1
for c in 3 4 5; do timeout -k 10 400 python bench.py --config $c --steps 800 --no-cpu-baseline > $O/bench_config$c.json 2> $O/bench_config$c.err || exit 1; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r2z/bench_*.json')):
    d=json.load(open(f)); r=d['roofline']; print(f, d['value'], d['ms_per_step'], 'lone', r['lone_frame_ms'], 'frac', r['frac'], 'hbm', r['hbm_frac_traffic'], r.get('hbm_frac_packet_alg'), 'cpu', (d.get('cpu_baseline') or {}).get('value'))
PY
