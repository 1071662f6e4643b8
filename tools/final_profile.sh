#!/bin/bash
# the round's evidence: bench line, rocprofv3 kernel stats of the same command, PMC passes (lone-frame and saturated launch)
set -u
TAG=${1:-final}
export TMPDIR=/tmp
timeout -k 10 400 python bench.py > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err || exit 1
cut -c1-220 gpurun_out/bench_$TAG.json
( cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG.log 2>&1 ) || exit 1
cat gpurun_out/prof_$TAG/*/*kernel_stats.csv | cut -c1-160 | head -5
bash tools/pmc_run.sh $TAG > gpurun_out/pmc_$TAG.log 2>&1 || exit 1
tail -25 gpurun_out/pmc_$TAG.log
bash tools/pmc_steady.sh $TAG > gpurun_out/pmcs_$TAG.log 2>&1 || exit 1
tail -40 gpurun_out/pmcs_$TAG.log
