"""Timeline of a short pipelined run (what the driver's `bench.py --steps 20 --warmup 5` does): HIP events around EVERY frame, on the
frame's own stream; prints each frame's start / end offset from the first start.  Usage: python tools/timeline.py [steps] [warmup] [streams]"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from snail_amd import HostBVH, scenes, FPSCamera
from snail_amd.render import DistributedRenderer
from snail_amd.scene import Scene
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
warm = int(sys.argv[2]) if len(sys.argv) > 2 else 5
ns = int(sys.argv[3]) if len(sys.argv) > 3 else 4
stag = int(sys.argv[4]) if len(sys.argv) > 4 else 0
tv = scenes.scene_by_name("atrium"); h = HostBVH.build(tv)
cam = FPSCamera(*scenes.atrium_camera()).camera()
sc = Scene(h, 0)
rnd = DistributedRenderer(sc, 1920, 1080, 0, 1, slots=ns)
for rep in range(3):
    for _ in range(warm): rnd.render(cam)
    rnd.flush(); torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    t0 = time.perf_counter()
    host = []
    for e in ev:
        rnd.render(cam, events=e); host.append(time.perf_counter() - t0)
    rnd.flush(); torch.cuda.synchronize()
    el = time.perf_counter() - t0
    base = ev[0][0]
    rows = [(base.elapsed_time(e0), base.elapsed_time(e1)) for e0, e1 in ev]
    print("rep %d: %d steps on %d streams: wall %.3f ms = %.4f ms/step; last end %.3f ms; host enqueue done at %.3f ms" % (rep, steps, ns, el * 1e3, el * 1e3 / steps, max(r[1] for r in rows), host[-1] * 1e3))
    if rep == 2:
        for i, (a, b) in enumerate(rows):
            print("  frame %2d stream %d: start %7.3f end %7.3f  dur %6.3f   host enqueued at %6.3f" % (i, i % ns, a, b, b - a, host[i] * 1e3))
