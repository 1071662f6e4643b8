#!/bin/bash
set -u
O=gpurun_out/r2h; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 200 python tools/ramp.py 0.5 2>&1 | grep -v amdgpu.ids
echo "== active wait"
for w in 0 2000; do
  for i in 1 2 3; do ROC_ACTIVE_WAIT_TIMEOUT=$w timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/w${w}_$i.json 2> $O/w.err || exit 1; done
  python - $w <<'PY'
import json,sys
w=sys.argv[1]
print(w, [json.load(open('gpurun_out/r2h/w%s_%d.json'%(w,i)))['value'] for i in (1,2,3)])
PY
done
timeout -k 10 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -4 $O/pytest_gpu.log
