"""Do consecutive dispatches of ONE stream overlap when launched with hipExtAnyOrderLaunch (AQL barrier bit cleared)?  hip_ext.h says the
flag is not supported on GFX9xx; this measures it: 20 full-frame primary launches back to back on one stream, with flags 0 and 1."""
import os as _os
_os.environ.setdefault("SNAIL_LIB_PATH", _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "snail_amd", "libsnailhip_debug.so"))  # workbench build (snail_debug_*)

import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from snail_amd import HostBVH, scenes, FPSCamera, _lib
from snail_amd.scene import Scene
tv = scenes.scene_by_name("atrium"); h = HostBVH.build(tv)
cam = FPSCamera(*scenes.atrium_camera()).camera()
sc = Scene(h, 0)
sc.trace_primary(cam, 1920, 1080); torch.cuda.synchronize()
cam13 = np.ascontiguousarray(cam.as_array13(), dtype=np.float32)
for rep in range(3):
    for flags in (0, 1):
        for frames in (1, 20):
            ms = C.c_float(0.0)
            _lib.check(_lib.lib().snail_debug_anyorder(sc._h, _lib.ptr(cam13), 1920, 1080, frames, flags, C.addressof(ms)), "snail_debug_anyorder")
            print("flags %d: %2d launches on one stream: %.4f ms total = %.4f ms per frame" % (flags, frames, ms.value, ms.value / frames), flush=True)
