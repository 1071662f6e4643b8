#!/bin/bash
# round-2 GPU session 6: transparency stage tests, PMC passes of the final kernel (configs 1 and 5), kernel-trace stats of bench.py
set -u
O=gpurun_out/r2f; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -q -k "transparency or adapter or whitted or invalid" > $O/pytest_sel.log 2>&1; echo "pytest rc $?"; tail -4 $O/pytest_sel.log
bash tools/pmc_run.sh r2c1 > $O/pmc_c1.log 2>&1; tail -24 $O/pmc_c1.log
bash tools/pmc_run.sh r2c5 --config 5 > $O/pmc_c5.log 2>&1; tail -8 $O/pmc_c5.log
cp gpurun_out/pmc_r2c1/summary.txt $O/r2_pmc_summary.txt; cp gpurun_out/pmc_r2c5/summary.txt $O/r2_stress_pmc_summary.txt
( cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/prof.log 2>&1 )
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/r2_kernel_stats.csv; head -6 $O/r2_kernel_stats.csv; rm -rf $O/prof
( cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 20 --warmup 5 > $GRAFT_REPO_ROOT/$O/prof20.log 2>&1 )
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/r2_kernel_stats_steps20.csv; head -4 $O/r2_kernel_stats_steps20.csv; rm -rf $O/prof
