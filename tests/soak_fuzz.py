"""Long seeded fuzz run (not collected by pytest; uses the oracle, hence lives under tests/): random cameras, frame sizes, lights,
scenes and tree builders, GPU hit records / staged config-3 frames / counters against the oracle, bit for bit.
Usage: python tests/soak_fuzz.py [cases] [seed] [focus]   (focus = "refl": the stress scene with the mirrored bounce only, differences printed;
focus = "sse": every case in SNAIL_ARITH_HOST_SSE against the oracle's ORC_MODE_SSE -- this CPU's rcpps / rsqrtps on both sides;
focus = "ref" / "refsse": the REFERENCE'S OWN MESHES -- lancia, feline, barracks from tests/golden/<name>_tris.npz, scanned / modelled geometry with
full-mantissa coordinates -- under random cameras and lights, in the IEEE / the host's SSE arithmetic)"""
import math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from snail_amd import FPSCamera
from snail_amd.scene import Scene
from tests import oracle_lib as O, util

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
focus = sys.argv[3] if len(sys.argv) > 3 else ""
rng = np.random.RandomState(seed)
names = ["atrium:0.05", "stress:0.05", "box", "chain", "atrium:0.02", "offgrid"]   # offgrid: full-mantissa vertices, slivers, four decades of sizes, far from the origin
scn = {}
SSE = focus in ("sse", "refsse")
if focus in ("ref", "refsse"):
    from snail_amd import HostBVH
    names = ["lancia", "feline", "barracks"]
    for n in names:
        tv = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", n + "_tris.npz"))["tris"]
        scn[n] = (tv, Scene(HostBVH.build(tv), 0), O.OracleScene(tv))
else:
    for n in names:
        tv, hb, osc = util.scene_pair(n)
        scn[n] = (tv, Scene(hb, 0), osc)
if SSE:
    for n in names: scn[n][1].set_arith("host_sse")
MODE = O.MODE_SSE if SSE else O.MODE_IEEE
L_ = O.lib(); L_.orc_caller_mxcsr.restype = __import__("ctypes").c_uint
print("caller MXCSR 0x%04x (0x1f80 = default); float32 denormals in numpy: %s" % (L_.orc_caller_mxcsr(), "kept" if float(np.float32(1e-40) * np.float32(0.5)) != 0.0 else "FLUSHED"), flush=True)
bad = 0; t0 = time.time()
for case in range(cases):
    name = names[rng.randint(len(names))]
    if focus == "refl": name = "stress:0.05"
    tv, sc, osc = scn[name]
    bmin, bmax = osc.nodes[0]["bmin"], osc.nodes[0]["bmax"]
    c, e = (bmin + bmax) * 0.5, (bmax - bmin)
    pos = c + (rng.rand(3) - 0.5) * e * (0.9 if rng.rand() < 0.65 else 4.0)
    yaw = rng.rand() * 2 * math.pi if rng.rand() < 0.8 else [0.0, math.pi / 2, math.pi, 1.5 * math.pi][rng.randint(4)]
    pitch = (rng.rand() - 0.5) * 3.0 if rng.rand() < 0.8 else 0.0
    cam = FPSCamera(pos.astype(np.float32), yaw, pitch, plane_dist=float(np.exp(rng.uniform(-2.0, 2.0)))).camera()
    resx, resy = int(rng.randint(1, 260)), int(rng.randint(1, 200))
    want = osc.render_primary(cam.as_array13(), resx, resy, mode=MODE)
    st = sc.new_stats(); fr = sc.trace_primary(cam, resx, resy, stats=st); torch.cuda.synchronize()
    ok = all(np.array_equal(g.cpu().numpy().view(np.uint32), w.view(np.uint32)) for g, w in zip((fr.t, fr.u, fr.v, fr.tri_id), want[:4]))
    ok = ok and np.array_equal(st.cpu().numpy().astype(np.uint64), want[4])
    nl = int(rng.randint(0, 4)); refl = bool(rng.rand() < 0.5)
    if focus == "refl": refl = True
    lights = np.zeros((nl, 7), dtype=np.float32)
    for k in range(nl):
        lights[k, :3] = c + (rng.rand(3) - 0.5) * e * 1.2
        lights[k, 3:6] = rng.rand(3); lights[k, 6] = float(e.max()) * float(np.exp(rng.uniform(-2.5, 1.0)))
    wimg, wst = osc.render_whitted(cam.as_array13(), resx, resy, lights, mode=MODE, reflections=refl)
    st = sc.new_stats(); img = sc.render_whitted(cam, resx, resy, lights, stats=st, reflections=refl); torch.cuda.synchronize()
    ok2 = np.array_equal(img.cpu().numpy(), wimg) and np.array_equal(st.cpu().numpy().astype(np.uint64), wst)
    if not (ok and ok2):
        bad += 1
        if focus in ("refl", "sse"):
            g = img.cpu().numpy(); w = np.flatnonzero((g != wimg).any(axis=-1).ravel()) if g.shape == wimg.shape else []
            print("  pixels differing: %d of %d, first %s; stats gpu %s oracle %s" % (len(w), g.shape[0] * g.shape[1], w[:6], st.cpu().numpy().astype(np.uint64), wst), flush=True)
            # a second run of the same frame: does the device agree with itself?
            st2 = sc.new_stats(); img2 = sc.render_whitted(cam, resx, resy, lights, stats=st2, reflections=refl); torch.cuda.synchronize()
            print("  rerun equals first run: %s, equals oracle: %s" % (bool(torch.equal(img, img2)), np.array_equal(img2.cpu().numpy(), wimg)), flush=True)
        print("MISMATCH case %d: %s %dx%d primary_ok=%s whitted_ok=%s nl=%d refl=%s pos=%s yaw=%.4f pitch=%.4f" % (case, name, resx, resy, ok, ok2, nl, refl, pos, yaw, pitch), flush=True)
    if case % 100 == 99:
        print("%d cases, %d mismatches, %.0f s" % (case + 1, bad, time.time() - t0), flush=True)
print("done: %d cases, %d mismatches" % (cases, bad))
sys.exit(1 if bad else 0)
