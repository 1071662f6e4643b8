"""How the frame rate develops from an idle GPU: after `idle` seconds without work, 600 pipelined frames with a HIP event after every
10th; prints ms per frame for each chunk of 10 (on the event's stream: coarse, but the trend is what matters) and the GPU clock that
rocm-smi reports before / after.  Usage: python tools/ramp.py [idle_seconds]"""
import os as _os
_os.environ.setdefault("SNAIL_LIB_PATH", _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "snail_amd", "libsnailhip_debug.so"))  # workbench build (snail_debug_*)

import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from snail_amd import HostBVH, scenes, FPSCamera
from snail_amd.render import DistributedRenderer
from snail_amd.scene import Scene
idle = float(sys.argv[1]) if len(sys.argv) > 1 else 0.5
tv = scenes.scene_by_name("atrium"); h = HostBVH.build(tv)
cam = FPSCamera(*scenes.atrium_camera()).camera()
sc = Scene(h, 0)
rnd = DistributedRenderer(sc, 1920, 1080, 0, 1)
from snail_amd import _lib
probe_stream = torch.cuda.Stream()
probes = torch.zeros((80, 2), dtype=torch.int64, device="cuda")
def probe(k):   # shader clock over the next 100 us, sampled by one sleeping wave on a stream of its own
    _lib.check(_lib.lib().snail_debug_clock_dev(100.0, _lib.ptr(probes[k]), probe_stream.cuda_stream), "clock")
for _ in range(40): rnd.render(cam)
rnd.flush(); torch.cuda.synchronize()
for rep in range(2):
    time.sleep(idle)
    N, step = 600, 10
    marks = []
    t0 = time.perf_counter()
    e0 = torch.cuda.Event(enable_timing=True); e0.record(rnd.streams[0])
    nprobe = 0
    for i in range(N):
        rnd.render(cam)
        if i % step == 0 and nprobe < 60:
            probe(nprobe); nprobe += 1
        if (i + 1) % step == 0:
            e = torch.cuda.Event(enable_timing=True); e.record(rnd.streams[i % rnd.nslots]); marks.append(e)
    rnd.flush(); torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    ts = [e0.elapsed_time(e) for e in marks]
    per = [(ts[k] - (ts[k - 1] if k else 0.0)) / step for k in range(len(ts))]
    print("rep %d (after %.1f s idle): wall %.3f ms/frame; ms/frame per chunk of %d frames:" % (rep, idle, wall * 1e3 / N, step))
    print("  " + " ".join("%.3f" % x for x in per))
    pr = probes[:nprobe].cpu().numpy().astype(np.float64)
    print("  shader clock (GHz) sampled at the start of each chunk: " + " ".join("%.2f" % (c / max(t, 1.0) * 0.1) for c, t in pr))
