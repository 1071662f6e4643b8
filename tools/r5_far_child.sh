#!/bin/bash
# Round 5, VERDICT item 3: a reject-only test of the FAR child at push time (SNAIL_FAR_PRETEST, the C++ walk only) against the same walk without it.
# Build host:  tools/variant.sh far0 -DSNAIL_DEBUG_API -DSNAIL_FAR_PRETEST=0;  tools/variant.sh far1 -DSNAIL_DEBUG_API -DSNAIL_FAR_PRETEST=1
# GPU box:     bash tools/r5_far_child.sh  (SNAIL_DEBUG_FORCE_DEEP=1 sends every packet through the C++ walk -- the only one the experiment exists in)
set -u
O=gpurun_out/r5_far; mkdir -p $O
for v in far0 far1; do
  for c in "--config 1" "--config 5" "--config 3 --reflections"; do
    tag=$(echo $c | tr -d " -")
    SNAIL_DEBUG_FORCE_DEEP=1 SNAIL_LIB_PATH=$PWD/snail_amd/exp/lib_$v.so timeout -k 10 300 python bench.py $c --steps 400 --warmup 50 --no-cpu-baseline --lone-frames 0 --arith ieee > $O/${v}_$tag.json 2> $O/${v}_$tag.err || echo "FAILED $v $c"
  done
done
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/*.json")):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f.split("/")[-1], d["value"], "Mrays/s", d["ms_per_step"], "ms/step verified", d["verified"], "node visits/step", d["config"]["node_visits_per_step"])
    except Exception as e:
        print(f, "ERR", e)
PY
