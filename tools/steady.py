"""Steady-state behaviour of dev::k_primary in ONE launch: the atrium 1080p packet list replicated R times (default 4), so
that the launch's tail is amortised and PMC counters (which serialise dispatches) describe the saturated machine.
Usage: python tools/steady.py [R] [reps]"""
import sys, os
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from snail_amd import HostBVH, scenes, FPSCamera
from snail_amd.scene import Scene
R = int(sys.argv[1]) if len(sys.argv) > 1 else 4
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
tv = scenes.scene_by_name("atrium"); h = HostBVH.build(tv)
pos, ang, pitch = scenes.atrium_camera(); cam = FPSCamera(pos, ang, pitch).camera()
sc = Scene(h, 0)
resx, resy = 1920, 1080
xs, ys = np.meshgrid(np.arange(0, resx, 16), np.arange(0, resy, 16))
# same block -> packet locality as the frame launch: 4x4-packet regions, 16 consecutive entries each
pw, ph = xs.shape[1], xs.shape[0]
order = []
for ry in range((ph + 3) // 4):
    for rx in range((pw + 3) // 4):
        for k in range(16):
            cx, cy = rx * 4 + (k & 3), ry * 4 + (k >> 2)
            if cx < pw and cy < ph: order.append((cx * 16, cy * 16))
xy = np.tile(np.array(order, dtype=np.int32), (R, 1))
pxy = torch.from_numpy(xy).cuda()
out = sc.trace_packets(cam, resx, resy, pxy); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(reps): sc.trace_packets(cam, resx, resy, pxy, out=out)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
print("R=%d packets=%d  %.4f ms/launch  %.4f ms per frame-equivalent  %.1f Mrays/s" % (R, len(xy), ms, ms / R, len(xy) * 256 / ms / 1e3))
