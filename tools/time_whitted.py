"""Timing of BASELINE config 3 (primary + shadow packets, staged on the device; optional third argument `refl` = one mirrored
bounce) -- development aid."""
import sys, os
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")   # one hardware queue per stream (before HIP initialises)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from snail_amd import HostBVH, scenes, FPSCamera
from snail_amd.scene import Scene
name = sys.argv[1] if len(sys.argv) > 1 else "atrium"
nl = int(sys.argv[2]) if len(sys.argv) > 2 else 1
refl = len(sys.argv) > 3 and sys.argv[3] == "refl"
resx, resy = 1920, 1080
tv = scenes.scene_by_name(name); h = HostBVH.build(tv)
pos, ang, pitch = scenes.atrium_camera() if name.startswith("atrium") else scenes.stress_camera()
cam = FPSCamera(pos, ang, pitch).camera()
sc = Scene(h, 0)
bmin, bmax = h.bbox(); c, e = (bmin + bmax) * 0.5, (bmax - bmin)
lights = np.array([[c[0], c[1] + 0.35 * e[1], c[2], 1.0, 0.9, 0.8, 2.0 * float(e.max())],
                   [c[0] - 0.3 * e[0], c[1] + 0.1 * e[1], c[2] + 0.2 * e[2], 0.3, 0.5, 1.0, 0.6 * float(e.max())]], dtype=np.float32)[:nl]
st = sc.new_stats()
out = sc.render_whitted(cam, resx, resy, lights, stats=st, reflections=refl); torch.cuda.synchronize()
s = st.cpu().numpy()
print("stats {intersects, iters, rays, skips}:", s.tolist()); print("rays traced per frame:", int(s[2]), "(primary 2088960 + %s %d)" % ("mirrored+shadow" if refl else "shadow", int(s[2]) - 2088960), "skips", int(s[3]))
for i in range(24): sc.render_whitted(cam, resx, resy, lights, reflections=refl)   # every scratch slot of the scene handle allocated
torch.cuda.synchronize()
NS = int(os.environ.get("NS", "4"))
streams = [torch.cuda.Stream() for _ in range(NS)]
outs = [torch.zeros_like(out) for _ in range(NS)]
for ns in (1, 1, 1, NS, NS, NS):   # the first round after a change of concurrency is a transient (queues, clocks)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    K = 90
    for i in range(K):
        s_ = streams[i % ns]
        s_.wait_event(e0) if i < ns else None
        with torch.cuda.stream(s_):
            sc.render_whitted(cam, resx, resy, lights, out=outs[i % ns], stream=s_, reflections=refl)
    for s_ in streams[:ns]:
        torch.cuda.current_stream().wait_stream(s_)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / K
    print("frames in flight %d: %.3f ms/frame, %.1f Mrays/s (primary+shadow)" % (ns, ms, s[2] / ms / 1e3))
# host-side cost of one call (launches only; the queue is drained first and after)
import time
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(200): sc.render_whitted(cam, resx, resy, lights, out=outs[0], reflections=refl)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("host time per call %.1f us (queue drained after another %.1f us per call)" % ((t1 - t0) / 200 * 1e6, (t2 - t1) / 200 * 1e6))
