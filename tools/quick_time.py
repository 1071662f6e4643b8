"""Ad-hoc timing of the primary kernel (development aid; bench.py is the contract)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from snail_amd import HostBVH, scenes, FPSCamera
from snail_amd.scene import Scene
name = sys.argv[1] if len(sys.argv) > 1 else "atrium"
resx, resy = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1920, 1080)
t0 = time.time(); tv = scenes.scene_by_name(name); t1 = time.time(); h = HostBVH.build(tv); t2 = time.time()
print("scene %s: %d tris, %d nodes, depth %d (gen %.2fs, build %.2fs)" % (name, len(tv), h.n_nodes, h.depth, t1 - t0, t2 - t1))
pos, ang, pitch = scenes.atrium_camera() if name.startswith("atrium") else scenes.stress_camera()
cam = FPSCamera(pos, ang, pitch).camera()
sc = Scene(h, 0)
fr = sc.alloc_frame(resx, resy)
stats = sc.new_stats()
sc.trace_primary(cam, resx, resy, out=fr, stats=stats); torch.cuda.synchronize()
st = stats.cpu().numpy()
print("stats", st, "iters/packet %.1f isect/ray %.2f" % (st[1] / (st[2] / 256), st[0] * 4 / st[2]))
print("hit frac", torch.isfinite(fr.t).float().mean().item())
acc = sc.account_primary(cam, resx, resy)
balg = (32 * acc[1] + 64 * acc[2]) / acc[0] + 16
print("account", acc, "Vn/ray %.1f Vt/ray %.2f B_alg %.0f" % (acc[1] / acc[0], acc[2] / acc[0], balg))
for rep in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(20): sc.trace_primary(cam, resx, resy, out=fr)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    rays = resx * ((resy + 15) // 16 * 16)
    print("%.3f ms/frame  %.1f Mrays/s  roofline frac %.3f" % (ms, rays / ms / 1e3, rays / (ms * 1e-3) * balg / 8e12))
