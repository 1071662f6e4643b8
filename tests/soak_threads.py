"""Concurrency soak (not collected by pytest; uses the oracle, hence lives under tests/): THREADS host threads hammer TWO scene handles -- one in the IEEE
arithmetic, one in the host's SSE arithmetic -- with a random mix of every kind of C-ABI call for SECONDS seconds, each result compared bit for bit with an
expectation the oracle computed beforehand.  The reference shares one const Scene over its render threads (src/render.cpp:214-267, src/thread_pool.cpp:151-180);
include/snail_hip.h ("Concurrency") promises the same of a SnailScene.
Usage: python tests/soak_threads.py [seconds] [threads] [seed]"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from snail_amd import FPSCamera, scenes
from snail_amd.scene import Scene
from snail_amd import render as R
from tests import oracle_lib as O, util

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
nthreads = int(sys.argv[2]) if len(sys.argv) > 2 else 8
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 1
rng = np.random.RandomState(seed)
resx, resy = 320, 192

cases = []   # (label, callable(thread stream) -> bool)
for name, mode, arith in (("atrium:0.05", O.MODE_IEEE, "ieee"), ("offgrid-in", O.MODE_SSE, "host_sse")):
    tv, hb, osc = util.scene_pair(name)
    sc = Scene(hb, 0)
    sc.set_arith(arith)
    base = util.camera_for(name, tv)
    bmin, bmax = osc.nodes[0]["bmin"], osc.nodes[0]["bmax"]
    c, e = (bmin + bmax) * 0.5, (bmax - bmin)
    for k in range(22):     # moving cameras and lights: more origins (44 per scene) than the origin-relative node cache holds (40)
        cam = FPSCamera(np.asarray(base.pos, dtype=np.float32) + np.float32(0.013 * k) * np.asarray(base.front, dtype=np.float32), *( (scenes.atrium_camera()[1:]) if name.startswith("atrium") else (0.0, 0.0))).camera()
        lights = np.array([[c[0] + 0.02 * k * e[0], c[1] + 0.35 * e[1], c[2], 1.0, 0.9, 0.8, 2.0 * float(e.max())]], dtype=np.float32)
        ref = osc.render_primary(cam.as_array13(), resx, resy, mode=mode)
        lit, lst = osc.render_whitted(cam.as_array13(), resx, resy, lights, mode=mode, reflections=(k % 2 == 1))
        depth = O.shade_depth(ref[0], mode=mode).reshape(resy, resx, 3)

        def dev_primary(st, sc=sc, cam=cam, ref=ref):
            with torch.cuda.stream(st):      # (the counters' and the frame's initial fills must be ON the launch's stream: torch's side streams do not wait for the null stream)
                stats = sc.new_stats()
                out = sc.alloc_frame(resx, resy)
            f = sc.trace_primary(cam, resx, resy, out=out, stats=stats, stream=st)
            st.synchronize()
            return all(np.array_equal(a.cpu().numpy().view(np.uint32), b.view(np.uint32)) for a, b in zip((f.t, f.u, f.v, f.tri_id), ref[:4])) and \
                np.array_equal(stats.cpu().numpy().astype(np.uint64), ref[4])

        def dev_whitted(st, sc=sc, cam=cam, lights=lights, lit=lit, lst=lst, refl=(k % 2 == 1)):
            with torch.cuda.stream(st):
                stats = sc.new_stats()
                out = torch.zeros((resy, resx, 3), dtype=torch.uint8, device="cuda")
            img = sc.render_whitted(cam, resx, resy, lights, out=out, stats=stats, stream=st, reflections=refl)
            st.synchronize()
            return np.array_equal(img.cpu().numpy(), lit) and np.array_equal(stats.cpu().numpy().astype(np.uint64), lst)

        def host_primary(st, sc=sc, cam=cam, ref=ref):
            t, u, v, tid, s4 = sc.trace_primary_host(cam, resx, resy)
            return np.array_equal(t.view(np.uint32), ref[0].view(np.uint32)) and np.array_equal(tid, ref[3]) and s4.tolist() == [int(x) for x in ref[4]]

        def host_image(st, sc=sc, cam=cam, lights=lights, lit=lit, lst=lst, depth=depth, ref=ref, refl=(k % 2 == 1)):
            if refl:
                img, s4 = sc.render_image_host(cam, resx, resy, lights7=lights, flags=sc.RENDER_REFLECTIONS)
                return np.array_equal(img, lit) and s4.tolist() == [int(x) for x in lst]
            img, s4 = sc.render_image_host(cam, resx, resy, flags=sc.RENDER_DEPTH)
            return np.array_equal(img, depth) and s4.tolist() == [int(x) for x in ref[4]]

        def host_tiles(st, sc=sc, cam=cam, depth=depth):
            tiles = R.divide_image(resx, resy)[::3]
            data, offs, _ = sc.render_tiles_host(cam, resx, resy, tiles, flags=sc.RENDER_DEPTH)
            return all(np.array_equal(data[o:o + len(w)], w) for o, w in zip(offs.tolist(), O.planar_encode(depth, tiles)))

        cases += [("%s dev primary %d" % (name, k), dev_primary), ("%s dev whitted %d" % (name, k), dev_whitted), ("%s host primary %d" % (name, k), host_primary),
                  ("%s host image %d" % (name, k), host_image), ("%s host tiles %d" % (name, k), host_tiles)]
    n_sh, n_ry = 9, 7
    so, sd, si, sdist = util.shadow_packets(osc, n_sh, 300 + len(cases))
    ro, rd, ri, rmask, rdist, robj, rbary = util.secondary_packets(osc, base, resx, resy, n_ry, 400 + len(cases), shared=False, masked=True)
    want_sh = sdist.copy(); sst = osc.trace_shadow(so, sd, si, want_sh, n_sh, 64, mode=mode)
    wd, wo, wb = rdist.copy(), robj.copy(), rbary.copy(); rst = osc.trace_rays(ro, rd, ri, rmask, wd, wo, wb, n_ry, 64, False, mode=mode)

    def host_shadow(st, sc=sc, so=so, sd=sd, si=si, sdist=sdist, want_sh=want_sh, sst=sst):
        d = sdist.copy()
        s4 = sc.trace_shadow_host(so, sd, si, d, len(so), 64)
        return np.array_equal(d.view(np.uint32), want_sh.view(np.uint32)) and [int(s4[i]) for i in (0, 1, 3)] == [int(sst[i]) for i in (0, 1, 3)]

    def host_rays(st, sc=sc, ro=ro, rd=rd, ri=ri, rmask=rmask, rdist=rdist, robj=robj, rbary=rbary, wd=wd, wo=wo, wb=wb, rst=rst):
        d, o, b = rdist.copy(), robj.copy(), rbary.copy()
        s4 = sc.trace_rays_host(ro, rd, ri, rmask, d, o, b, len(rdist) // 64, 64, False)
        return np.array_equal(d.view(np.uint32), wd.view(np.uint32)) and np.array_equal(o, wo) and np.array_equal(b.view(np.uint32), wb.view(np.uint32)) and \
            [int(s4[0]), int(s4[1])] == [int(rst[0]), int(rst[1])]

    cases += [("%s host shadow" % name, host_shadow), ("%s host rays" % name, host_rays)] * 3

print("%d cases, %d threads, %.0f s" % (len(cases), nthreads, seconds), flush=True)
bad, done = [], [0] * nthreads
stop = time.time() + seconds
gate = threading.Barrier(nthreads)


def body(k):
    r = np.random.RandomState(seed * 1000 + k)
    st = torch.cuda.Stream()
    gate.wait()
    while time.time() < stop and len(bad) < 10:
        label, fn = cases[r.randint(len(cases))]
        try:
            ok = fn(st)
        except Exception as ex:  # noqa: BLE001
            ok, label = False, "%s raised %r" % (label, ex)
        if not ok:
            bad.append((k, label))
        done[k] += 1


ts = [threading.Thread(target=body, args=(k,)) for k in range(nthreads)]
t0 = time.time()
for t in ts: t.start()
for t in ts: t.join()
print("done: %d calls in %.1f s on %d threads (%s per thread), %d mismatches" % (sum(done), time.time() - t0, nthreads, done, len(bad)))
for k, label in bad[:10]:
    print("  MISMATCH thread %d: %s" % (k, label))
sys.exit(1 if bad else 0)
