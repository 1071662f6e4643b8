#!/bin/bash
# the bench lines of tools/r3_final.sh again, AFTER profiles/traffic.json was regenerated from that run's PMC passes (so that every fraction is
# priced with the final kernel's own counters: `counters_stale` false)
set -u
O=gpurun_out/r3y; mkdir -p $O
for i in 1 2 3; do timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $O/bench_steps20_$i.json 2> $O/err.log || exit 1; done
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --settle-ms 0 --no-cpu-baseline > $O/bench_steps20_nosettle.json 2>> $O/err.log || exit 1
timeout -k 10 400 python bench.py > $O/bench_default.json 2>> $O/err.log || exit 1
timeout -k 10 400 python bench.py --camera-path orbit --no-cpu-baseline > $O/bench_orbit.json 2>> $O/err.log || exit 1
for c in 3 4 5; do timeout -k 10 400 python bench.py --config $c --steps 800 --no-cpu-baseline > $O/bench_config$c.json 2>> $O/err.log || exit 1; done
timeout -k 10 400 python bench.py --gpus 2 --backend gloo --steps 200 --warmup 20 > $O/bench_gloo2.json 2>> $O/err.log || exit 1
timeout -k 10 400 python bench.py --gpus 2 --backend gloo --steps 200 --warmup 20 --frames-per-launch 1 > $O/bench_gloo2_fpl1.json 2>> $O/err.log || exit 1
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3y/bench_*.json')):
    d=json.loads([l for l in open(f) if l.startswith('{')][-1]); r=d['roofline']
    print(f.split('/')[-1], d['value'], d['ms_per_step'], 'lone', r.get('lone_frame_ms'), d['config'].get('lone_launch_ms'), 'frac', r['frac'], 'stale', r.get('counters_stale'), 'hbm', r.get('hbm_frac_traffic'), 'cpu', (d.get('cpu_baseline') or {}).get('value'), (d.get('cpu_baseline') or {}).get('scalar_port_value'), 'fpl1', (r.get('one_frame_per_launch') or {}).get('value'))
PY
