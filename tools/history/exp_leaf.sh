#!/bin/bash
# experiment: broadcast of a leaf's surviving triangle by ds_bpermute (product) vs v_readlane (-DSNAIL_EXP_LEAF_READLANE).
# Build first, on the build host: tools/history/exp_leaf.sh build (-> snail_amd/exp/lib_readlane.so, travels with the snapshot); then on the GPU box:
# tools/history/exp_leaf.sh.  Variants are loaded through SNAIL_LIB_PATH (snail_amd/_lib.py): the product library is never overwritten.
set -u
if [ "${1:-}" = build ]; then
  cd "$(dirname "$0")/../snail_amd/csrc" && mkdir -p ../exp
  FLAGS=$(grep '^FLAGS' Makefile | sed 's/^FLAGS *?= *//')
  /opt/rocm/bin/hipcc --offload-arch=gfx950 $FLAGS -DSNAIL_EXP_LEAF_READLANE -shared snail_hip.hip bvh_build.cpp -o ../exp/lib_readlane.so && echo built
  exit
fi
for round in 1 2; do
for v in base ${VARIANTS:-readlane}; do
  if [ $v = base ]; then unset SNAIL_LIB_PATH; else export SNAIL_LIB_PATH=$PWD/snail_amd/exp/lib_$v.so; fi
  for sc in atrium stress; do
    timeout -k 10 300 python bench.py --no-cpu-baseline --scene $sc > gpurun_out/expleaf_${v}_$sc.json 2> gpurun_out/expleaf.err || exit 1
    python - $v $sc <<'PY'
import json,sys
d=json.load(open('gpurun_out/expleaf_%s_%s.json'%(sys.argv[1],sys.argv[2]))); print(sys.argv[1], sys.argv[2], d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])
PY
  done
done
done
