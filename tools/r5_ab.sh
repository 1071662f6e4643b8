#!/bin/bash
# Round 5's same-box A/B runs of bench.py, one experiment per call:  bash tools/r5_ab.sh <experiment> > gpurun_out/<file>.txt
# A variant is `product` (snail_amd/libsnailhip.so), `none` (product with --feedback-order 0) or the name of a library built beforehand in the build
# container with tools/variant.sh NAME -DFLAG... (snail_amd/exp/lib_NAME.so, loaded through SNAIL_LIB_PATH).  Every line: round, variant, bench arguments,
# Mrays/s, ms per step, verified.  What each experiment showed: profiles/README.md "Round 5".
#   prio        rank priorities (product of that day = s_setprio 3/2/1 by rank) vs prio0 (-DSNAIL_PRIO_RANK=0) vs prio0nat (+ -DSNAIL_ORDER_HEAVY_SHIFT=13) vs none   -> profiles/r5_prio_and_order.txt
#   natural     sorted vs heavy13 (-DSNAIL_ORDER_HEAVY_SHIFT=13: the built-in order through the feedback path) vs none                                                -> profiles/r5_order_natural.txt
#   adaptive    the adaptive rule vs no feedback, ten workloads                                                                                                        -> profiles/r5_prio_and_order.txt (second part)
#   threshold   -DSNAIL_ORDER_ADAPTIVE=4 (product of that day) / 3 / 2                                                                                                 -> profiles/r5_order_threshold.txt
#   steps20     the driver's command, adaptive rule vs adapt0 (-DSNAIL_ORDER_ADAPTIVE=0), launch shapes                                                                -> profiles/r5_steps20.txt
#   steps20b    the same with the exact-costs hint in the renderer, + long runs                                                                                        -> profiles/r5_steps20b.txt
#   steps20c    frames per launch that divide the 20 steps evenly                                                                                                      -> profiles/r5_steps20c.txt
#   shapes      streams x frames per launch (nine shapes, four workloads); streams for config 3                                                                         -> profiles/r5_shapes.txt
#   dolly       the dolly camera at 2 / 4 / 8 frames per launch                                                                                                        -> profiles/r5_dolly.txt
#   reltri      reltri (-DSNAIL_REL_TRI_PREFETCH=1): a leaf's triangle line requested ahead in the walks over origin-relative arrays; parity tests of the variant first -> profiles/r5_tri_prefetch.txt
#   notri       notri (-DSNAIL_TRI_PREFETCH=0): no such request in the walks over the loop's own copy (mirrored packets)                                               -> profiles/r5_tri_prefetch.txt
set -u
ROUNDS=2; VARIANTS="product"; PRE=""; CASES=(); TAILFLAGS="--lone-frames 0 --no-live-check"; PARITY=""
run() { # variant, bench arguments...
  local v=$1; shift
  local L="SNAIL_AB=1" F=""
  case $v in product) ;; none) F="--feedback-order 0" ;; *) L="SNAIL_LIB_PATH=$PWD/snail_amd/exp/lib_$v.so" ;; esac
  env $L timeout -k 10 200 python bench.py $PRE "$@" $F --no-cpu-baseline $TAILFLAGS 2>/dev/null |
    python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$R $v [$PRE $*]', d['value'], d['ms_per_step'], 'lone', d['roofline'].get('lone_frame_ms'), d['verified'])"
}
C3="--config 3 --steps 800"; C3R="--config 3 --reflections --steps 300"; C4="--config 4 --steps 800"; C5="--config 5 --steps 800"
case "${1:-}" in
prio)      VARIANTS="product prio0 prio0nat none"; CASES=("--camera-path static" "--camera-path orbit" "$C3" "$C3 --camera-path orbit" "$C5" "$C5 --camera-path orbit" "$C3R" "$C3R --camera-path orbit") ;;
natural)   VARIANTS="product heavy13 none"; CASES=("--camera-path static" "--camera-path orbit" "--camera-path dolly" "$C3" "$C3 --camera-path orbit" "$C4") ;;
adaptive)  VARIANTS="product none"; CASES=("--camera-path static" "--camera-path orbit" "--camera-path dolly" "$C3" "$C3 --camera-path orbit" "$C4" "$C5" "$C5 --camera-path orbit" "$C3R" "$C3R --camera-path orbit") ;;
threshold) VARIANTS="product adapt3 adapt2"; CASES=("$C3R" "$C3R --camera-path orbit" "$C3" "$C3 --camera-path orbit" "--camera-path orbit" "--camera-path static" "$C4") ;;
steps20)   ROUNDS=3; VARIANTS="product adapt0"; PRE="--steps 20 --warmup 5"
           CASES=("" "--frames-per-launch 4" "--frames-per-launch 8" "--frames-per-launch 4 --streams 5" "--frames-per-launch 1 --streams 8" "--streams 6" "--streams 3 --frames-per-launch 4") ;;
steps20b)  ROUNDS=3; VARIANTS="product adapt0"
           CASES=("--steps 20 --warmup 5" "--steps 20 --warmup 5 --streams 3 --frames-per-launch 4" "" "--streams 3 --frames-per-launch 4" "--camera-path orbit" "--camera-path orbit --streams 3 --frames-per-launch 4" "$C3" "$C3 --camera-path orbit") ;;
steps20c)  ROUNDS=3; PRE="--steps 20 --warmup 5"
           CASES=("" "--frames-per-launch 5" "--frames-per-launch 5 --streams 4" "--frames-per-launch 7 --streams 3" "--frames-per-launch 8 --streams 3" "--frames-per-launch 5 --streams 5" "--frames-per-launch 6 --streams 4" "--frames-per-launch 3 --streams 4") ;;
shapes)    for sh in "4 2" "3 4" "2 8" "3 8" "2 4" "3 3" "3 6" "4 4" "2 6"; do set -- $sh; for c in "--steps 20 --warmup 5" "" "$C5" "$C4"; do CASES+=("$c --streams $1 --frames-per-launch $2"); done; done
           for s in 2 3 4 5 6; do CASES+=("$C3 --streams $s" "$C3R --streams $s"); done ;;
dolly)     CASES=("--camera-path dolly --frames-per-launch 4" "--camera-path dolly --frames-per-launch 2" "--camera-path dolly --frames-per-launch 8" "--camera-path orbit" "") ;;
reltri)    ROUNDS=3; VARIANTS="product reltri"; TAILFLAGS="--lone-frames 12"; PARITY='tests/test_gpu_parity.py tests/test_gpu_ref_meshes.py -k "primary or whitted or config3 or reference_mesh or full_size"'
           CASES=("" "--steps 20 --warmup 5" "$C5" "$C3" "$C4" "$C3R" "--camera-path orbit") ;;
notri)     ROUNDS=3; VARIANTS="product notri"; TAILFLAGS="--lone-frames 12"; PARITY='tests/test_gpu_parity.py -k "rays or shadow or whitted or refl or transp"'
           CASES=("$C3R" "$C3R --scene stress") ;;
*) echo "usage: bash tools/r5_ab.sh prio|natural|adaptive|threshold|steps20|steps20b|steps20c|shapes|dolly|reltri|notri"; exit 2 ;;
esac
if [ -n "$PARITY" ]; then   # the GPU parity tests under the (last) variant before anything is timed
  eval "SNAIL_LIB_PATH=$PWD/snail_amd/exp/lib_${VARIANTS##* }.so timeout -k 10 600 python -m pytest -x -q $PARITY" 2>&1 | tail -2
fi
for R in $(seq 1 $ROUNDS); do
  for c in "${CASES[@]}"; do
    for v in $VARIANTS; do run $v $c; done
  done
done
