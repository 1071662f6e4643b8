import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from snail_amd import HostBVH, scenes, FPSCamera, _lib
from snail_amd.scene import Scene
for name in ("atrium", "stress"):
    tv = scenes.scene_by_name(name); h = HostBVH.build(tv)
    pos, ang, pitch = scenes.atrium_camera() if name.startswith("atrium") else scenes.stress_camera()
    cam = FPSCamera(pos, ang, pitch).camera(); sc = Scene(h, 0)
    resx, resy = 1920, 1080; pw, ph = (resx + 15) // 16, (resy + 15) // 16
    out = np.zeros((ph * pw, 8), dtype=np.uint32)
    cam13 = np.ascontiguousarray(cam.as_array13(), dtype=np.float32)
    for rep in range(3):
        _lib.check(_lib.lib().snail_account_packets(sc._h, _lib.ptr(cam13), resx, resy, _lib.ptr(out)), "costs")
    it, isc, cyc = (out[:, k].astype(np.float64) for k in range(3))
    A = np.stack([it, isc, np.ones_like(it)], 1)
    coef, *_ = np.linalg.lstsq(A, cyc, rcond=None)
    print(name, "corr(cyc,iters)=%.4f corr(cyc,isect)=%.4f fit cyc = %.1f*iters + %.2f*isect + %.0f, corr(fit)=%.4f" % (
        np.corrcoef(cyc, it)[0, 1], np.corrcoef(cyc, isc)[0, 1], coef[0], coef[1], coef[2], np.corrcoef(cyc, A @ coef)[0, 1]),
        "mean iters %.0f isect %.0f" % (it.mean(), isc.mean()))
