#!/bin/bash
# lone-wave (1 wave/SIMD via unused dynamic LDS) counter breakdown of the primary kernel
export TMPDIR=/tmp
export SNAIL_DEBUG_DYNLDS=${1:-36000}
OUT=gpurun_out/pmc_lone_$SNAIL_DEBUG_DYNLDS
mkdir -p $OUT
run() { local name=$1; shift
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline > $OUT/$name.log 2>&1; echo "$name rc=$?"; }
run a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA
run b SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM
run c SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_WAVE32_LDS
python tools/pmc_summary.py $OUT
