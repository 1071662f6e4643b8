"""Timing of the TraverseShadow entry point (snail_trace_shadow_dev): N shadow packets as Scene::TraceLight builds them (tests/util.shadow_packets),
each with its own light position, traced R times.  Usage: python tools/time_shadow.py [scene] [packets] [repeats]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from snail_amd.scene import Scene, ShadowContext
from tests import util

name = sys.argv[1] if len(sys.argv) > 1 else "atrium"
npk = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 40
tv, hb, osc = util.scene_pair(name)
sc = Scene(hb, 0)
origin, dirs, idir, dist = util.shadow_packets(osc, npk, seed=7, size=64)
tt = torch.from_numpy
o, d, i, dist0 = tt(origin).cuda(), tt(dirs).cuda(), tt(idir).cuda(), tt(dist).cuda()
rays = int((dist > 0).sum())
def once():
    ctx = ShadowContext(o, d, i, dist0.clone(), size=64)
    sc.traverse_shadow(ctx)
    return ctx
for _ in range(5): once()
torch.cuda.synchronize()
best = 1e9
for rnd in range(3):
    t0 = time.perf_counter()
    for _ in range(reps): once()
    torch.cuda.synchronize()
    best = min(best, (time.perf_counter() - t0) / reps)
print("%s: %d shadow packets (%d live rays): %.3f ms per batch, %.1f Mrays/s" % (name, npk, rays, best * 1e3, rays / best / 1e6))
