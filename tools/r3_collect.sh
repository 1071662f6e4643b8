#!/bin/bash
# copies what tools/r3_final.sh left under gpurun_out/r3z/ into profiles/ (run on the build host after the GPU call)
set -eu
O=gpurun_out/r3z
cp $O/traffic.json profiles/traffic.json
for f in $O/r3_final_*; do cp $f profiles/; done
for f in $O/bench_*.json; do cp $f profiles/r3_final_$(basename $f); done
python - <<'PY'
import json, sys
sys.path.insert(0, '.')
import bench
d = json.load(open('profiles/traffic.json'))
print('counter hash', d.get('_kernel_sha16'), 'sources', bench.kernel_source_sha16(), 'OK' if d.get('_kernel_sha16') == bench.kernel_source_sha16() else 'STALE')
PY
