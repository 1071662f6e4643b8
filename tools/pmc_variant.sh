#!/bin/bash
# instruction counters of dev::k_primary for library variants: tools/pmc_variant.sh <variant|base> ...
set -u
export TMPDIR=/tmp
for V in "$@"; do
  if [ $V = base ]; then unset SNAIL_LIB_PATH; else export SNAIL_LIB_PATH=$GRAFT_REPO_ROOT/snail_amd/exp/lib_$V.so; fi
  OUT=$GRAFT_REPO_ROOT/gpurun_out/pmcv_$V; rm -rf $OUT; mkdir -p $OUT
  ( cd /tmp && timeout -k 10 240 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $OUT/sq1 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --lone-frames 0 --frames-per-launch 1 > $OUT/sq1.log 2>&1 )
  ( cd /tmp && timeout -k 10 240 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU --output-format csv -d $OUT/sq2 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --lone-frames 0 --frames-per-launch 1 > $OUT/sq2.log 2>&1 )
  echo "== $V"; python $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT
  find $OUT -name "*.csv" -size +200k -delete 2>/dev/null
done
