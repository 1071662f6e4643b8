#!/bin/bash
# PMC passes (separate runs per counter group; --kernel-trace only, no other trace domains) of an arbitrary command, every kernel summarised.
# usage: tools/pmc_cmd.sh <tag> <program> [args...]   (the program itself, e.g. python3 tools/whitted_once.py atrium 1 refl)
#   -> gpurun_out/pmc_<tag>/<group>/...csv ; gpurun_out/pmc_<tag>/summary.txt = per-kernel means per dispatch
set -u
TAG=$1; shift
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc_$TAG
mkdir -p $OUT
run() { # name, counters...
  local name=$1; shift
  ( cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -- "${CMD[@]}" > $OUT/$name.log 2>&1 )
  echo "$name rc=$?"
}
CMD=("$@")
run fetch FETCH_SIZE
run write WRITE_SIZE
run sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES
run sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SMEM SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU
python tools/pmc_summary.py $OUT all | tee $OUT/summary.txt
find $OUT -name "*.csv" -size +200k -delete 2>/dev/null
