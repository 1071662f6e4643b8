"""Load-balance study: per-packet cost of one primary frame (diagnostic entry snail_account_packets)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from snail_amd import HostBVH, scenes, FPSCamera, _lib
from snail_amd.scene import Scene
name = sys.argv[1] if len(sys.argv) > 1 else "atrium"
resx, resy = 1920, 1080
tv = scenes.scene_by_name(name); h = HostBVH.build(tv)
pos, ang, pitch = scenes.atrium_camera() if name.startswith("atrium") else scenes.stress_camera()
cam = FPSCamera(pos, ang, pitch).camera()
sc = Scene(h, 0)
pw, ph = (resx + 15) // 16, (resy + 15) // 16
out = np.zeros((ph * pw, 8), dtype=np.uint32)
cam13 = np.ascontiguousarray(cam.as_array13(), dtype=np.float32)
for rep in range(3):
    _lib.check(_lib.lib().snail_account_packets(sc._h, _lib.ptr(cam13), resx, resy, _lib.ptr(out)), "costs")
it, isect, cyc, start = (out[:, k].astype(np.float64) for k in range(4))
print("packets", len(out), "iters mean %.1f max %d" % (it.mean(), it.max()), "isect mean %.1f max %d" % (isect.mean(), isect.max()))
q = np.percentile(cyc, [5, 25, 50, 75, 90, 99, 100])
print("cycles: mean %.0f  p5 %.0f p25 %.0f p50 %.0f p75 %.0f p90 %.0f p99 %.0f max %.0f" % ((cyc.mean(),) + tuple(q)))
print("sum cycles / 1024 SIMDs = %.0f  (= %.1f us at 2.4 GHz if perfectly packed 1 wave/SIMD)" % (cyc.sum() / 1024, cyc.sum() / 1024 / 2400))
print("corr(cycles, iters) %.3f  corr(cycles, isect) %.3f" % (np.corrcoef(cyc, it)[0, 1], np.corrcoef(cyc, isect)[0, 1]))
A = np.stack([it, isect, np.ones_like(it)], axis=1)
coef, *_ = np.linalg.lstsq(A, cyc, rcond=None)
print("fit cycles ~ %.0f*iters + %.1f*isect + %.0f" % tuple(coef))
st = (start - start.min()) * 64 / 100.0  # s_memtime ticks at 100 MHz? print raw spread
print("start spread (raw>>6 units): min 0 max %.0f; fraction started in first 10%% of span: %.2f" % (st.max(), (st < 0.1 * st.max()).mean()))
m = cyc.reshape(ph, pw)
rows = m.reshape(ph // 4 if ph % 4 == 0 else ph, -1)
print("per packet-row mean kcycles:", " ".join("%.0f" % (x / 1000) for x in m.mean(axis=1)))

# ---- list-scheduling simulation: what would better dispatch orders buy? (durations = measured cycles) ----
import heapq
def simulate(order, group, slots):
    """greedy: `slots` concurrent groups; a group = `group` consecutive packets of `order`, occupying a slot for max(durations)"""
    d = cyc[order]
    n = (len(d) + group - 1) // group
    gd = [d[i * group:(i + 1) * group].max() for i in range(n)]
    h = [0.0] * slots
    heapq.heapify(h)
    end = 0.0
    for x in gd:
        t = heapq.heappop(h) + x
        end = max(end, t)
        heapq.heappush(h, t)
    return end
nat = np.arange(len(cyc))
# current launch: blocks of 2x2 packets, XCD-banded; approximate by natural order in groups of 4
print("sim (kcycles): natural/4-wave blocks %.0f | natural/1-wave blocks %.0f | LPT exact %.0f | LPT by iters %.0f | LPT noisy(0.76) %.0f | lower bounds: max %.0f, sum/4096 %.0f"
      % (simulate(nat, 4, 1024) / 1e3, simulate(nat, 1, 4096) / 1e3, simulate(np.argsort(-cyc), 1, 4096) / 1e3, simulate(np.argsort(-it), 1, 4096) / 1e3,
         simulate(np.argsort(-(cyc * (1 + 0.85 * np.random.RandomState(0).randn(len(cyc)) * cyc.std() / cyc.mean()))), 1, 4096) / 1e3, cyc.max() / 1e3, cyc.sum() / 4096 / 1e3))
rng = np.random.RandomState(1)
print("sim random order/1-wave blocks %.0f" % (simulate(rng.permutation(len(cyc)), 1, 4096) / 1e3))
