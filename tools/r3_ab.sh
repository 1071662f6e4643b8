# A/B of kernel variants: tools/r3_ab.sh <label>=<lib or -> ...   (- = the product library); atrium default run, steps-20 with settle, stress
O=gpurun_out/r3ab; mkdir -p $O
for spec in "$@"; do
  label=${spec%%=*}; lib=${spec#*=}
  if [ "$lib" = "-" ]; then unset SNAIL_LIB_PATH; else export SNAIL_LIB_PATH=$PWD/$lib; fi
  for rep in 1 2; do
    python bench.py --no-cpu-baseline --lone-frames 6 > $O/${label}_c1_$rep.json 2>> $O/err.log
    python bench.py --no-cpu-baseline --lone-frames 0 --config 5 --steps 800 > $O/${label}_c5_$rep.json 2>> $O/err.log
  done
  python bench.py --no-cpu-baseline --lone-frames 0 --config 3 --steps 800 > $O/${label}_c3_1.json 2>> $O/err.log
  python - <<PY
import json,glob
for f in sorted(glob.glob("$O/${label}_*.json")):
    d=json.loads([l for l in open(f) if l.startswith("{")][-1]); print(f.split("/")[-1], d["value"], d["ms_per_step"], d["roofline"].get("lone_frame_ms"))
PY
done
