"""Helper of tests/test_gpu_parity.py::test_walk_variants_selected_by_scene_size (run as a child process with SNAIL_DEBUG_NO_PACK=1 or
SNAIL_DEBUG_FORCE_DEEP=1 in the environment -- the library reads them once): the walks that ordinary test scenes never select -- two-word
stack entries (scenes of more than 2^20 node slots: no record prefetch) and the second stack register pair (trees deeper than 62) -- on a
primary frame, the staged light pipeline with the mirrored bounce, and generic / shadow packets, bit-compared with the oracle.
Prints one JSON line."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from snail_amd.scene import Context, Scene, ShadowContext
from tests import oracle_lib as O
from tests import util


def main():
    name, resx, resy = "atrium:0.05", 328, 200
    tv, hb, osc = util.scene_pair(name)
    cam = util.camera_for(name, tv)
    sc = Scene(hb, 0)
    out = {"env": {k: os.environ.get(k) for k in ("SNAIL_DEBUG_NO_PACK", "SNAIL_DEBUG_FORCE_DEEP")}}
    st = sc.new_stats()
    fr = sc.trace_primary(cam, resx, resy, stats=st)
    torch.cuda.synchronize()
    t, u, v, tid, ost = osc.render_primary(cam.as_array13(), resx, resy, mode=O.MODE_IEEE)
    out["primary"] = bool(np.array_equal(fr.t.cpu().numpy().view(np.uint32), t.view(np.uint32)) and np.array_equal(fr.tri_id.cpu().numpy(), tid)
                          and np.array_equal(fr.u.cpu().numpy().view(np.uint32), u.view(np.uint32)) and np.array_equal(st.cpu().numpy().astype(np.uint64), ost))
    bmin, bmax = osc.nodes[0]["bmin"], osc.nodes[0]["bmax"]
    c, e = (bmin + bmax) * 0.5, (bmax - bmin)
    lights = np.array([[c[0], c[1] + 0.35 * e[1], c[2], 1.0, 0.9, 0.8, 2.0 * float(e.max())]], dtype=np.float32)
    want, wst = osc.render_whitted(cam.as_array13(), resx, resy, lights, mode=O.MODE_IEEE, reflections=True)
    st = sc.new_stats()
    got = sc.render_whitted(cam, resx, resy, lights, stats=st, reflections=True)
    torch.cuda.synchronize()
    out["whitted_refl"] = bool(np.array_equal(got.cpu().numpy(), want) and np.array_equal(st.cpu().numpy().astype(np.uint64), wst))
    dev = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a)).cuda()
    ok = True
    for shared, masked in ((True, False), (False, True), (False, False)):
        o, d, i, m, dist, obj, bary = util.secondary_packets(osc, cam, resx, resy, 6, 31, shared, masked)
        wd, wo, wb = dist.copy(), obj.copy(), bary.copy()
        wst = osc.trace_rays(o, d, i, m, wd, wo, wb, 6, 64, shared)
        ctx = Context(dev(o), dev(d), dev(i), dev(dist), dev(obj), dev(bary), 64, shared, dev(m))
        st = sc.new_stats()
        sc.traverse_primary(ctx, stats=st)
        torch.cuda.synchronize()
        ok = ok and np.array_equal(ctx.distance.cpu().numpy().view(np.uint32), wd.view(np.uint32)) and np.array_equal(ctx.object.cpu().numpy(), wo) \
            and np.array_equal(ctx.barycentric.cpu().numpy().view(np.uint32), wb.view(np.uint32)) and int(st[1]) == int(wst[1]) and int(st[0]) == int(wst[0])
    out["rays"] = bool(ok)
    so, sd, si, sdist = util.shadow_packets(osc, 6, 32)
    wd = sdist.copy()
    wst = osc.trace_shadow(so, sd, si, wd, 6, 64)
    sctx = ShadowContext(dev(so), dev(sd), dev(si), dev(sdist), 64)
    st = sc.new_stats()
    sc.traverse_shadow(sctx, stats=st)
    torch.cuda.synchronize()
    out["shadow"] = bool(np.array_equal(sctx.distance.cpu().numpy().view(np.uint32), wd.view(np.uint32)) and int(st[1]) == int(wst[1]) and int(st[3]) == int(wst[3]))
    sc.close()
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
