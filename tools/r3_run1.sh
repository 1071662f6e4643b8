set -e
mkdir -p gpurun_out/r3a
python -m pytest tests -m gpu -x -q > gpurun_out/r3a/pytest.log 2>&1 || { tail -40 gpurun_out/r3a/pytest.log; exit 1; }
tail -3 gpurun_out/r3a/pytest.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r3a/bench_s20.json 2> gpurun_out/r3a/bench_s20.err
python bench.py > gpurun_out/r3a/bench_default.json 2> gpurun_out/r3a/bench_default.err
python bench.py --camera-path orbit --no-cpu-baseline > gpurun_out/r3a/bench_orbit.json 2> gpurun_out/r3a/bench_orbit.err
python bench.py --gpus 2 --backend gloo --steps 200 --warmup 20 > gpurun_out/r3a/bench_g2.json 2> gpurun_out/r3a/bench_g2.err
python -c "
import json
for f in ('bench_s20','bench_default','bench_orbit','bench_g2'):
    d=json.loads([l for l in open('gpurun_out/r3a/%s.json'%f) if l.startswith('{')][-1])
    print(f, d['value'], d['ms_per_step'], d['roofline'].get('frac'), d['roofline'].get('one_frame_per_launch'), d['config'].get('lone_launch_ms'), d['roofline'].get('lone_frame_ms'))
"
