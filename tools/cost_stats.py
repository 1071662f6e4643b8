import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from snail_amd import FPSCamera, HostBVH, scenes
from snail_amd.scene import Scene
def light_of(hb):
    bmin, bmax = hb.bbox(); c, e = (bmin + bmax) * 0.5, (bmax - bmin)
    return np.array([[c[0], c[1] + 0.35 * e[1], c[2], 1.0, 0.9, 0.8, 2.0 * float(e.max())]], dtype=np.float32)
for name, res in (("atrium", (1920, 1080)), ("atrium", (3840, 2160)), ("stress", (1920, 1080))):
    tv = scenes.scene_by_name(name); hb = HostBVH.build(tv); sc = Scene(hb, 0)
    cam = FPSCamera(*(scenes.stress_camera() if name == "stress" else scenes.atrium_camera())).camera()
    n = sc.primary_slots(*res)
    cost = torch.zeros((4, n), dtype=torch.int32, device="cuda")
    sc.render_whitted(cam, res[0], res[1], light_of(hb), reflections=True, slot_cost=cost)
    torch.cuda.synchronize()
    c = cost.cpu().numpy().astype(np.int64)
    for k, lab in enumerate(("primary", "shadow(primary)", "mirrored", "shadow(mirrored)")):
        x = c[k]; nz = x[x > 0]
        top = np.sort(x)[::-1]
        print("%-7s %4dx%-4d %-17s slots %6d packets %6d max %5d p99.9 %5d p99 %5d mean %7.1f sum %9d  max*6144/sum %.2f  (2 frames: %.2f)" % (
            name, res[0], res[1], lab, n, len(nz), x.max(), top[len(x) // 1000], top[len(x) // 100], nz.mean(), x.sum(), x.max() * 6144.0 / x.sum(), x.max() * 6144.0 / (2 * x.sum())))
    sc.close()
