#!/bin/bash
# PMC passes for the primary kernel (separate runs per counter group; --kernel-trace only, no other trace domains).
# usage: tools/pmc_run.sh <tag> [bench.py arguments, e.g. --config 5]   -> gpurun_out/pmc_<tag>/<group>/...csv ; prints per-counter means for dev::k_primary
set -u
TAG=${1:-x}; shift || true
export TMPDIR=/tmp
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
run() { # name, counters...
  local name=$1; shift
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --lone-frames 0 --frames-per-launch 1 --settle-ms 0 $EXTRA > $OUT/$name.log 2>&1
  echo "$name rc=$?"
}
EXTRA="$*"
run fetch FETCH_SIZE
run write WRITE_SIZE
run tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum
run sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES
run sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SMEM SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU
run grbm GRBM_GUI_ACTIVE GRBM_COUNT
run lds SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS
python tools/pmc_summary.py $OUT | tee $OUT/summary.txt
find $OUT -name "*.csv" -size +200k -delete 2>/dev/null   # keep the merge-back small: the summary is what is kept
