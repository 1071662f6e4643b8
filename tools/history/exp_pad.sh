#!/bin/bash
# experiment: throughput sensitivity to extra VALU / SALU / s_nop per node visit.  Build the variants first, on the build host:
# tools/history/build_pad_variants.sh (-> snail_amd/exp/lib_*.so, which travel with the snapshot); results in profiles/README.md.  Runs on the
# GPU box.  Variants are loaded through SNAIL_LIB_PATH (snail_amd/_lib.py): the product library is never overwritten.
set -u
for v in base mulchain mul2 max3 subs mov salu20 nop20 vcmp rfl; do
  if [ $v = base ]; then unset SNAIL_LIB_PATH; else export SNAIL_LIB_PATH=$PWD/snail_amd/exp/lib_$v.so; fi
  timeout -k 10 200 python bench.py --no-cpu-baseline > gpurun_out/exp_$v.json 2> gpurun_out/exp_$v.err || exit 1
  python - $v <<'PY'
import json,sys
d=json.load(open('gpurun_out/exp_%s.json'%sys.argv[1])); print(sys.argv[1], d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])
PY
done
