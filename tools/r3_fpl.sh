for c in 5 1; do for f in 2 3 4 6; do python bench.py --no-cpu-baseline --lone-frames 0 --config $c --steps 1200 --frames-per-launch $f > gpurun_out/fpl_c${c}_$f.json 2>> gpurun_out/fpl.err; python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/fpl_c${c}_$f.json") if l.startswith("{")][-1]); print("c$c fpl $f", d["value"], d["ms_per_step"])
PY
done; done
