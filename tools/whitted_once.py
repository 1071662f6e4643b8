"""A few frames of the staged light pipeline, one at a time (for PMC passes: tools/pmc_cmd.sh).
usage: python3 tools/whitted_once.py [scene] [lights] [refl|norefl] [ieee|host_sse] -- absolute paths only under rocprofv3 (cwd is /tmp)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from snail_amd import HostBVH, scenes, FPSCamera
from snail_amd.scene import Scene
name = sys.argv[1] if len(sys.argv) > 1 else "atrium"
nl = int(sys.argv[2]) if len(sys.argv) > 2 else 1
refl = len(sys.argv) > 3 and sys.argv[3] == "refl"
tv = scenes.scene_by_name(name); h = HostBVH.build(tv)
pos, ang, pitch = scenes.atrium_camera() if name.startswith("atrium") else scenes.stress_camera()
cam = FPSCamera(pos, ang, pitch).camera()
sc = Scene(h, 0)
sc.set_arith(sys.argv[4] if len(sys.argv) > 4 else "ieee")
bmin, bmax = h.bbox(); c, e = (bmin + bmax) * 0.5, (bmax - bmin)
lights = np.array([[c[0], c[1] + 0.35 * e[1], c[2], 1.0, 0.9, 0.8, 2.0 * float(e.max())]], dtype=np.float32)[:nl]   # bench.py --config 3's light
st = sc.new_stats()
for i in range(6):
    sc.render_whitted(cam, 1920, 1080, lights, stats=st if i == 0 else None, reflections=refl)
    torch.cuda.synchronize()
print("rays per frame", int(st.cpu().numpy()[2]))
