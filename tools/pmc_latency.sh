#!/bin/bash
# memory latencies seen by dev::k_primary in a saturated launch (tools/steady.py): derived VmemLatency / SmemLatency (cycles) and the
# vector L1 TLB counters.  usage: tools/pmc_latency.sh <tag>
set -u
TAG=${1:-x}
export TMPDIR=/tmp
OUT=gpurun_out/pmcl_$TAG
mkdir -p $OUT
run() { local name=$1; shift
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -- python tools/steady.py 8 3 > $OUT/$name.log 2>&1
  echo "$name rc=$?"; }
run l1 VmemLatency
run l2 SmemLatency
run l3 TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum
run l4 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_ANY SQ_WAVE_CYCLES
python tools/pmc_summary.py $OUT | tee $OUT/summary.txt
