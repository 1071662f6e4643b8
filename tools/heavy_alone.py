"""Where a lone frame's time goes: the heaviest packets of the atrium 1080p frame traced ALONE (1, 16, 256, 1024, ... packets per
launch, heaviest first), against the whole frame.  A 1-packet launch is the frame's critical path at full speed: no scheduling can
finish the frame sooner.  Usage: python tools/heavy_alone.py [scene]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from snail_amd import HostBVH, scenes, FPSCamera
from snail_amd.scene import Scene
name = sys.argv[1] if len(sys.argv) > 1 else "atrium"
resx, resy = 1920, 1080
tv = scenes.scene_by_name(name); h = HostBVH.build(tv)
cam = FPSCamera(*(scenes.atrium_camera() if name.startswith("atrium") else scenes.stress_camera())).camera()
sc = Scene(h, 0)
pk = sc.packet_costs(cam, resx, resy)
pw = (resx + 15) // 16
order = np.argsort(-pk[:, 0].astype(np.int64), kind="stable")
print("heaviest packets: visits", pk[order[:5], 0].tolist(), "tri tests", pk[order[:5], 1].tolist(), "cycles in a lone frame", pk[order[:5], 2].tolist(),
      "fetched", pk[order[:5], 4].tolist(), "leaf bodies", pk[order[:5], 5].tolist())
print("mean packet: visits %.1f cycles %.0f" % (pk[:, 0].mean(), pk[:, 2].mean()))
xy_all = np.stack([(order % pw) * 16, (order // pw) * 16], axis=1).astype(np.int32)
def timed(n, reps=20):
    xy = torch.from_numpy(np.ascontiguousarray(xy_all[:n])).cuda()
    out = sc.trace_packets(cam, resx, resy, xy); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); sc.trace_packets(cam, resx, resy, xy, out=out); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort(); return ts[len(ts) // 2]
for n in (1, 4, 16, 64, 256, 1024, 2048, 4096, len(order)):
    print("%5d heaviest packets alone: %.4f ms" % (n, timed(n)), flush=True)
# the same counts of LIGHT packets, for the launch floor
xy_all = xy_all[::-1].copy()
for n in (1, 1024):
    print("%5d lightest packets alone: %.4f ms" % (n, timed(n)), flush=True)
