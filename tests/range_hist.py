"""Histogram of the active quad range (last-first+1) at node and leaf visits of primary packets, from the oracle's
instrumented walk (test infrastructure; single-threaded).  Usage: python tests/range_hist.py [scene] [resx resy]"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import oracle_lib as O
from tests.util import scene_pair, camera_for

name = sys.argv[1] if len(sys.argv) > 1 else "atrium"
resx, resy = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1920, 1080)
tv, hb, osc = scene_pair(name)
cam = camera_for(name, tv)
hist = np.zeros(192, dtype=np.uint64)
L = O.lib()
L.orc_debug_range_hist.argtypes = [ctypes.c_void_p]
L.orc_debug_range_hist(hist.ctypes.data)
t, u, v, tid, stats = osc.render_primary(cam.as_array13(), resx, resy, threads=1)
L.orc_debug_range_hist(None)
for lab, h in (("inner(entry)", hist[:64]), ("leaf(entry)", hist[64:128]), ("leaf(after box test)", hist[128:])):
    tot = int(h.sum()); c = np.cumsum(h) / max(tot, 1)
    print("%-22s total %9d  mean width %.1f  <=4:%.2f <=8:%.2f <=16:%.2f <=32:%.2f <=48:%.2f ==64:%.2f" % (
        lab, tot, float((h * np.arange(1, 65)).sum()) / max(tot, 1), c[3], c[7], c[15], c[31], c[47], h[63] / max(tot, 1)))
print("stats", stats)
