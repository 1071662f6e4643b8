O=gpurun_out/r4z; mkdir -p $O
for i in 1 2 3; do timeout -k 10 400 python bench.py --steps 20 --warmup 5 > $O/bench_steps20_$i.json 2> $O/bench_steps20_$i.err || echo FAIL; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4z/bench_steps20_*.json')):
    d=json.loads([l for l in open(f) if l.startswith('{')][-1]); r=d['roofline']
    print(f.split('/')[-1], d['config']['arith'], d['value'], d['ms_per_step'], 'frac', r['frac'], r.get('frac_of_measured_issue_rate'), 'stale', r.get('counters_stale'), 'verified', d.get('verified'), 'other', (r.get('other_arith') or {}).get('value'), 'cpu', (d.get('cpu_baseline') or {}).get('value'), (d.get('cpu_baseline') or {}).get('scalar_port_value'))
PY
