#!/bin/bash
# PMC passes over ONE saturated launch of dev::k_primary (tools/steady.py: the 1080p packet list x8 in one dispatch).
# Separate runs per counter group; --kernel-trace only.  usage: tools/pmc_steady.sh <tag>
set -u
TAG=${1:-x}
export TMPDIR=/tmp
OUT=gpurun_out/pmcs_$TAG
mkdir -p $OUT
run() { local name=$1; shift
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -- python tools/steady.py 8 3 > $OUT/$name.log 2>&1
  echo "$name rc=$?"; }
run g1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH
run g2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VALU2 SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM
run g3 SQ_IFETCH SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_TC_STALL
run g4 InstrFetchLatency OccupancyPercent SALUBusy VALUBusy
run g5 GRBM_GUI_ACTIVE SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM_RD SQ_CYCLES
python tools/pmc_summary.py $OUT | tee $OUT/summary.txt
