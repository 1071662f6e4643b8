#!/bin/bash
# scalar-cache and wait counters of dev::k_primary (two --pmc passes over bench.py --steps 10 --frames-per-launch 1); run on the GPU box
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_sqc; rm -rf $OUT; mkdir -p $OUT
( cd /tmp && timeout -k 10 240 rocprofv3 --kernel-trace --pmc SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_ICACHE_REQ SQC_ICACHE_MISSES --output-format csv -d $OUT/a -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --lone-frames 0 --frames-per-launch 1 > $OUT/a.log 2>&1 )
( cd /tmp && timeout -k 10 240 rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_INST_CYCLES_SMEM SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INST_LEVEL_SMEM --output-format csv -d $OUT/b -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --lone-frames 0 --frames-per-launch 1 > $OUT/b.log 2>&1 )
python $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT

find $OUT -name "*.csv" -size +200k -delete 2>/dev/null
