"""Seeded soak of the generic entry points (not collected by pytest): random batches of secondary packets (shared / per-ray origins,
lane masks, packet sizes 1..64, optional non-finite poison -> M_EXACT deferral) and shadow packets, GPU against the oracle bit for bit.
Usage: python tests/soak_rays.py [batches] [seed] [focus]   (focus = "perray": per-ray-origin masked packet batches only, "perray1": those with 1..3 quads per packet; differences are printed;
"sse": every batch in SNAIL_ARITH_HOST_SSE against ORC_MODE_SSE -- the caller's rays as they are, 1 / det of the accepted hits in the host's arithmetic)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from snail_amd.scene import Scene, Context, ShadowContext
from tests import oracle_lib as O, util

batches = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
focus = sys.argv[3] if len(sys.argv) > 3 else ""
rng = np.random.RandomState(seed)
names = ["atrium:0.05", "stress:0.05", "chain"]
scn = {}
for n in names:
    tv, hb, osc = util.scene_pair(n)
    scn[n] = (tv, Scene(hb, 0), osc, util.camera_for(n, tv))
    if focus == "sse": scn[n][1].set_arith("host_sse")
MODE = O.MODE_SSE if focus == "sse" else O.MODE_IEEE
tt = torch.from_numpy
L_ = O.lib(); L_.orc_caller_mxcsr.restype = __import__("ctypes").c_uint
print("caller MXCSR 0x%04x (0x1f80 = default); float32 denormals in numpy: %s" % (L_.orc_caller_mxcsr(), "kept" if float(np.float32(1e-40) * np.float32(0.5)) != 0.0 else "FLUSHED"), flush=True)
bad = 0; t0 = time.time()
for b in range(batches):
    name = names[rng.randint(len(names))]
    tv, sc, osc, cam = scn[name]
    size = int([64, 64, 64, 16, 1, rng.randint(1, 65)][rng.randint(6)])
    npk = int(rng.randint(1, 40))
    if rng.rand() < 0.7 or focus.startswith("perray"):
        shared, masked, poison = bool(rng.rand() < 0.5), bool(rng.rand() < 0.5), bool(rng.rand() < 0.15)
        if focus.startswith("perray"): shared, masked, poison = False, True, False
        if focus == "perray1": size = int(rng.randint(1, 4))   # tiny packets (a masked placeholder ray in lane 0 is likely: what tells the packet's octant from lane 0's signs)
        if poison and size < 3: poison = False
        origin, dirs, idir, mask, dist, obj, bary = util.secondary_packets(osc, cam, 640, 368, npk, seed=int(rng.randint(1 << 30)), shared=shared, masked=masked,
                                                                           size=size, poison=poison)
        d2, o2, b2 = dist.copy(), obj.copy(), bary.copy()
        ost = osc.trace_rays(origin, dirs, idir, mask, d2, o2, b2, npk, size, shared, mode=MODE)
        ctx = Context(tt(origin).cuda(), tt(dirs).cuda(), tt(idir).cuda(), tt(dist.copy()).cuda(), tt(obj.copy()).cuda(), tt(bary.copy()).cuda(),
                      size=size, shared_origin=shared, mask=None if mask is None else tt(mask).cuda())
        st = sc.new_stats(); sc.traverse_primary(ctx, stats=st); torch.cuda.synchronize()
        s = st.cpu().numpy().astype(np.uint64)
        ok = (np.array_equal(ctx.object.cpu().numpy(), o2) and np.array_equal(ctx.distance.cpu().numpy().view(np.uint32), d2.view(np.uint32)) and
              np.array_equal(ctx.barycentric.cpu().numpy().view(np.uint32), b2.view(np.uint32)) and s[0] == ost[0] and s[1] == ost[1])
        what = "rays shared=%s masked=%s size=%d poison=%s npk=%d" % (shared, masked, size, poison, npk)
        if not ok:
            go, gd, gb = ctx.object.cpu().numpy(), ctx.distance.cpu().numpy(), ctx.barycentric.cpu().numpy()
            wo, wd, wb = np.flatnonzero(go.ravel() != o2.ravel()), np.flatnonzero(gd.ravel().view(np.uint32) != d2.ravel().view(np.uint32)), np.flatnonzero(gb.ravel().view(np.uint32) != b2.ravel().view(np.uint32))
            what += " | object differs at %d %s dist at %d %s bary at %d %s stats gpu %s oracle %s" % (len(wo), wo[:4], len(wd), wd[:4], len(wb), wb[:4], s[:4], [int(x) for x in ost[:4]])
            ctx2 = Context(tt(origin).cuda(), tt(dirs).cuda(), tt(idir).cuda(), tt(dist.copy()).cuda(), tt(obj.copy()).cuda(), tt(bary.copy()).cuda(),
                           size=size, shared_origin=shared, mask=None if mask is None else tt(mask).cuda())
            st2 = sc.new_stats(); sc.traverse_primary(ctx2, stats=st2); torch.cuda.synchronize()
            what += " | rerun: object equal to first run %s, to oracle %s; stats %s" % (bool(torch.equal(ctx2.object, ctx.object)), np.array_equal(ctx2.object.cpu().numpy(), o2), st2.cpu().numpy()[:2])
            if len(wd): what += " | dist gpu %s oracle %s obj gpu %s oracle %s" % (gd.ravel()[wd[:3]], d2.ravel()[wd[:3]], go.ravel()[wd[:3]], o2.ravel()[wd[:3]])
    else:
        origin, dirs, idir, dist = util.shadow_packets(osc, npk, seed=int(rng.randint(1 << 30)), size=size)
        d2 = dist.copy()
        ost = osc.trace_shadow(origin, dirs, idir, d2, npk, size, mode=MODE)
        ctx = ShadowContext(tt(origin).cuda(), tt(dirs).cuda(), tt(idir).cuda(), tt(dist.copy()).cuda(), size=size)
        st = sc.new_stats(); sc.traverse_shadow(ctx, stats=st); torch.cuda.synchronize()
        s = st.cpu().numpy().astype(np.uint64)
        ok = np.array_equal(ctx.distance.cpu().numpy().view(np.uint32), d2.view(np.uint32)) and s[0] == ost[0] and s[1] == ost[1] and s[3] == ost[3]
        what = "shadow size=%d npk=%d" % (size, npk)
    if not ok:
        bad += 1; print("MISMATCH batch %d %s: %s" % (b, name, what), flush=True)
    if b % 200 == 199: print("%d batches, %d mismatches, %.0f s" % (b + 1, bad, time.time() - t0), flush=True)
print("done: %d batches, %d mismatches" % (batches, bad))
sys.exit(1 if bad else 0)
