"""How many node visits would a REJECT-ONLY test of the far child at push time save?  (round 5; test infrastructure: the oracle's
instrumented walk, single-threaded.)  The reference pushes the far child untested (src/bvh/traverse.cpp:71-74) and tests it when it is popped; a far
child none of whose lanes passes with the distances of the PUSH cannot pass later (distances only shrink, the range is the pushed one), so the push
could be dropped and its one LoopIteration counted.  Per workload and pushed range width: pops, pops that fail, pops that already fail at push time.
Usage: python tests/far_child_hist.py [atrium|stress|...] [resx resy] [refl]"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import oracle_lib as O
from tests.util import scene_pair, camera_for

name = sys.argv[1] if len(sys.argv) > 1 else "atrium"
resx, resy = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1920, 1080)
refl = "refl" in sys.argv
tv, hb, osc = scene_pair(name)
cam = camera_for(name, tv)
hist = np.zeros((2, 64, 4), dtype=np.uint64)
L = O.lib()
L.orc_debug_far_hist.argtypes = [ctypes.c_void_p]
L.orc_debug_far_hist(hist.ctypes.data)
if refl:
    bmin, bmax = osc.nodes[0]["bmin"], osc.nodes[0]["bmax"]
    c, e = (bmin + bmax) * 0.5, (bmax - bmin)
    lights = np.array([[c[0], c[1] + 0.35 * e[1], c[2], 1.0, 0.9, 0.8, 2.0 * float(e.max())]], dtype=np.float32)
    _, stats = osc.render_whitted(cam.as_array13(), resx, resy, lights, threads=1, reflections=True)
else:
    stats = osc.render_primary(cam.as_array13(), resx, resy, threads=1)[4]
L.orc_debug_far_hist(None)
print("%s %dx%d%s  TreeStats %s" % (name, resx, resy, " + mirrored bounce" if refl else "", [int(x) for x in stats]))
for k, lab in ((0, "shared-origin packets (primary)"), (1, "per-ray-origin packets (mirrored)")):
    h = hist[k].astype(np.int64)
    visits = int(h[:, 3].sum())
    if visits == 0:
        continue
    print(" %s: %d node visits, %d far-child pops, %d of them fail (%.1f %% of visits), %d fail at push time already (%.1f %% of visits)" % (
        lab, visits, int(h[:, 0].sum()), int(h[:, 1].sum()), 100.0 * h[:, 1].sum() / visits, int(h[:, 2].sum()), 100.0 * h[:, 2].sum() / visits))
    for lo, hi in ((0, 3), (4, 7), (8, 15), (16, 31), (32, 47), (48, 62), (63, 63)):
        s = h[lo:hi + 1].sum(axis=0)
        if s[3]:
            print("   range width %2d..%2d quads: %9d visits (%.1f %% of all), %8d pops, fail at pop %5.1f %%, fail at push %5.1f %% of the pops = %5.2f %% of ALL visits" % (
                lo + 1, hi + 1, s[3], 100.0 * s[3] / visits, s[0], 100.0 * s[1] / max(1, s[0]), 100.0 * s[2] / max(1, s[0]), 100.0 * s[2] / visits))
    nar = h[:32].sum(axis=0)
    print("   ranges <= 32 quads together: fail at push = %.2f %% of ALL visits, %.1f %% of the visits that start with such a range" % (100.0 * nar[2] / visits, 100.0 * nar[2] / max(1, nar[3])))
