# A/B of kernel variants on the secondary-ray paths: tools/r3_ab2.sh <label>=<lib or -> ...  (- = the product library)
# config 3 (primary + one light's shadow packets) twice, the mirrored-bounce frame (tools/time_whitted.py atrium 1 refl) twice
O=gpurun_out/r3ab2; mkdir -p $O
for spec in "$@"; do
  label=${spec%%=*}; lib=${spec#*=}
  if [ "$lib" = "-" ]; then unset SNAIL_LIB_PATH; else export SNAIL_LIB_PATH=$PWD/$lib; fi
  for rep in 1 2; do
    python bench.py --no-cpu-baseline --lone-frames 0 --config 3 --steps 800 > $O/${label}_c3_$rep.json 2>> $O/err.log
    python - <<PY
import json
d=json.loads([l for l in open("$O/${label}_c3_$rep.json") if l.startswith("{")][-1]); print("${label} c3", d["value"], d["ms_per_step"])
PY
    timeout -k 10 300 python tools/time_whitted.py atrium 1 refl 2>&1 | grep "frames in flight" | sed "s/^/${label} refl: /"
  done
done
