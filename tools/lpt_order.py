"""Experiment: does dispatching the heaviest packets first (costs fed back from the previous frame of the same camera) shorten the
frame?  Packet list in cost-descending order through snail_trace_packets_dev against the kernel's own region-interleaved order."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from snail_amd import HostBVH, scenes, FPSCamera, _lib
from snail_amd.scene import Scene
name = sys.argv[1] if len(sys.argv) > 1 else "atrium"
resx, resy = 1920, 1080
tv = scenes.scene_by_name(name); h = HostBVH.build(tv)
pos, ang, pitch = scenes.atrium_camera() if name.startswith("atrium") else scenes.stress_camera()
cam = FPSCamera(pos, ang, pitch).camera()
sc = Scene(h, 0)
pw, ph = (resx + 15) // 16, (resy + 15) // 16
out = np.zeros((ph * pw, 8), dtype=np.uint32)
cam13 = np.ascontiguousarray(cam.as_array13(), dtype=np.float32)
for rep in range(2):
    _lib.check(_lib.lib().snail_account_packets(sc._h, _lib.ptr(cam13), resx, resy, _lib.ptr(out)), "costs")
iters = out[:, 0].astype(np.int64)                     # node visits: a cost proxy that a product path could return for free
ys, xs = np.divmod(np.arange(pw * ph), pw)
xy_all = np.stack([xs * 16, ys * 16], axis=1).astype(np.int32)
orders = {"row-major": np.arange(pw * ph), "heaviest first (node visits of the previous frame)": np.argsort(-iters, kind="stable")}
streams = [torch.cuda.Stream() for _ in range(4)]
def timeit(fn, ns, K=400):
    for _ in range(40): fn(torch.cuda.current_stream())
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(K):
        s_ = streams[i % ns]
        with torch.cuda.stream(s_): fn(s_)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / K * 1e3
frame = sc.alloc_frame(resx, resy)
print("frame launch (region-interleaved order): lone %.4f ms, 4 in flight %.4f ms" % (timeit(lambda s_: sc.trace_primary(cam, resx, resy, out=frame, stream=s_), 1),
                                                                                   timeit(lambda s_: sc.trace_primary(cam, resx, resy, out=frame, stream=s_), 4)))
for label, o in orders.items():
    xy = torch.from_numpy(np.ascontiguousarray(xy_all[o])).cuda()
    bufs = sc.trace_packets(cam, resx, resy, xy)
    print("packet list, %-52s lone %.4f ms, 4 in flight %.4f ms" % (label + ":", timeit(lambda s_: sc.trace_packets(cam, resx, resy, xy, out=bufs, stream=s_), 1),
                                                                  timeit(lambda s_: sc.trace_packets(cam, resx, resy, xy, out=bufs, stream=s_), 4)))
