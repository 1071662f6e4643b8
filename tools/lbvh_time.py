"""The GPU builder option (snail_scene_create_lbvh) against the parity tree (host SAH sweep): build time and primary-ray throughput
of the same kernels on either tree.  Usage: python tools/lbvh_time.py [scene] [max_leaf_tris]"""
import sys, os
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from snail_amd import HostBVH, scenes, FPSCamera
from snail_amd.scene import Scene
from snail_amd.render import DistributedRenderer
name = sys.argv[1] if len(sys.argv) > 1 else "atrium"
max_leaf = int(sys.argv[2]) if len(sys.argv) > 2 else 4
tv = scenes.scene_by_name(name)
pos, ang, pitch = scenes.atrium_camera() if name.startswith("atrium") else scenes.stress_camera()
cam = FPSCamera(pos, ang, pitch).camera()
resx, resy = 1920, 1080
t0 = time.perf_counter(); h = HostBVH.build(tv); t_sah = time.perf_counter() - t0
Scene.from_lbvh(tv, 0, max_leaf).close()           # warm-up (module load, rocPRIM kernels)
t0 = time.perf_counter(); lb = Scene.from_lbvh(tv, 0, max_leaf); t_lb = time.perf_counter() - t0
used = 0; stack = [0]
while stack:
    i = stack.pop(); used += 1
    nd = lb.bvh.nodes[i]
    if not (nd["sub"] & 0x80000000): stack += [int(nd["sub"]), int(nd["sub"]) + 1]
print("%s: %d triangles; SAH sweep (host) %.3f s -> %d nodes, depth %d; LBVH (device) %.2f ms of kernels, %.1f ms incl. upload/download -> %d nodes in use, depth %d"
      % (name, len(tv), t_sah, h.n_nodes, h.depth, lb.build_ms, t_lb * 1e3, used, lb.bvh.depth))
for label, sc in (("SAH ", Scene(h, 0)), ("LBVH", lb)):
    rnd = DistributedRenderer(sc, resx, resy, 0, 1, slots=4)
    st = sc.new_stats(); sc.trace_primary(cam, resx, resy, stats=st); torch.cuda.synchronize(); s = st.cpu().numpy()
    for rep in range(3):
        for _ in range(20): rnd.render(cam)
        rnd.flush(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(100): rnd.render(cam)
        rnd.flush(); torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 100 * 1e3
    print("%s tree: %.4f ms/frame = %.0f Mrays/s; node visits per packet %.1f, quad-triangle tests per packet %.0f" % (label, ms, 2088960 / ms / 1e3, s[1] / 8160, s[0] / 8160))
