set -e
O=gpurun_out/r3b; mkdir -p $O
run() { tag=$1; shift; python bench.py --no-cpu-baseline --lone-frames 0 "$@" > $O/$tag.json 2> $O/$tag.err; python -c "
import json,sys
d=json.loads([l for l in open('$O/$tag.json') if l.startswith('{')][-1]); print('$tag', d['value'], d['ms_per_step'])"; }
for i in 1 2; do
run s20_fpl2_$i --steps 20 --warmup 5
run s20_fpl2_settle30_$i --steps 20 --warmup 5 --settle-ms 30
run s20_fpl2_settle100_$i --steps 20 --warmup 5 --settle-ms 100
run s20_fpl4_settle100_$i --steps 20 --warmup 5 --settle-ms 100 --frames-per-launch 4
run s20_fpl5_settle100_$i --steps 20 --warmup 5 --settle-ms 100 --frames-per-launch 5
run s20_fpl5_$i --steps 20 --warmup 5 --frames-per-launch 5
run s20_fpl1_settle100_$i --steps 20 --warmup 5 --settle-ms 100 --frames-per-launch 1
done
run orbit_default --camera-path orbit
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --camera-path orbit --lone-frames 0 > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
cd $GRAFT_REPO_ROOT; find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/orbit_kernel_stats.csv; head -6 $O/orbit_kernel_stats.csv; rm -rf $O/prof
