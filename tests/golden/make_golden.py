"""Generates the committed golden fixtures (run in the build container: `python tests/golden/make_golden.py`).

What can and cannot be pinned here (see oracle/snail_oracle.h): the reference's hot-path translation
units do not compile without the absent libfwk submodule, so the reference itself cannot be run to emit
hit records.  The fixtures are therefore:
  veclib_prims.json   outputs of the REFERENCE's own header-only veclib (oracle/_ref/veclib_probe, built
                      from /root/reference/veclib) on seeded bit patterns -- only the operations whose
                      result is CPU-independent (Min/Max/Condition, scalar Inv/RSqrt, Vec3 dot/cross);
  survey_digests.json the numbers SURVEY.md section 8(c) recorded from the reference in the survey session;
  oracle_*.npz        outputs of THIS repo's oracle in ORC_MODE_IEEE (CPU-independent), so that the GPU
                      box can check both the oracle build and the HIP path against committed bytes.
"""
import hashlib
import json
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from tests import oracle_lib as O   # noqa: E402
from tests import util              # noqa: E402


def expr_rows():
    """seeded rows of eight bit patterns for `veclib_probe exprs`: specials (zeros, infinities, NaNs, denormals, the SafeInv
    singularity -1e-8) in every lane position, unit-ish direction vectors, and values of wildly different magnitude"""
    rng = np.random.RandomState(4321)
    special = [0x00000000, 0x80000000, 0x7f800000, 0xff800000, 0x7fc00000, 0xffc00000, 0x3f800000, 0xbf800000,
               0x00000001, 0x80000001, 0x007fffff, 0x7f7fffff, 0x322bcc77, 0xb22bcc77, 0x437f0000, 0x3b808081]
    rows = []
    for i in range(len(special)):
        for j in range(0, len(special), 3):
            rows.append([special[(i + k) % len(special)] for k in range(4)] + [special[(j + 2 * k) % len(special)] for k in range(4)])
    f = (rng.randn(300, 8) * np.exp(rng.uniform(-12, 12, size=(300, 8)))).astype(np.float32)
    rows += f.view(np.uint32).tolist()
    d = rng.randn(300, 8).astype(np.float32)          # direction-like and colour-like values
    d[:, 4:] = np.abs(d[:, 4:]) * np.float32(0.6)
    rows += d.view(np.uint32).tolist()
    return rows


def run_exprs(probe, rows):
    text = "\n".join(" ".join("%08x" % v for v in r) for r in rows) + "\n"
    out = subprocess.run([probe, "exprs"], input=text, capture_output=True, text=True, check=True).stdout.split("\n")
    res = [[int(x, 16) for x in line.split()] for line in out if line.strip()]
    assert len(res) == len(rows) and all(len(r) == 57 for r in res)
    return res


def veclib_exprs():
    probe = os.path.join(ROOT, "oracle", "_ref", "veclib_probe")
    if not os.path.exists(probe):
        print("veclib_probe not built (reference checkout absent?) -- keeping existing veclib_exprs.json")
        return
    rows = expr_rows()
    res = run_exprs(probe, rows)
    json.dump({"layout": "words 0..44 of `veclib_probe exprs` (oracle/veclib_probe.cpp): ForWhich|ForAny<<4|ForAll<<5, Sqrt x4, Abs x4, dot x4, cross xyz x4, "
                         "Reflect xyz x4, Condition(Vec3q).x x4, Trunc(Clamp(*255)) x4; words 45..56 (FastInv, attenuation, SafeInv) are rcpps-based, "
                         "CPU specific, and compared live against the probe instead",
               "inputs": rows, "outputs": [r[:45] for r in res]}, open(os.path.join(HERE, "veclib_exprs.json"), "w"))
    print("veclib_exprs.json: %d rows" % len(rows))


def sha(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def veclib_prims():
    probe = os.path.join(ROOT, "oracle", "_ref", "veclib_probe")
    if not os.path.exists(probe):
        print("veclib_probe not built (reference checkout absent?) -- keeping existing veclib_prims.json")
        return
    rng = np.random.RandomState(1234)
    special = [0x00000000, 0x80000000, 0x7f800000, 0xff800000, 0x7fc00000, 0xffc00000, 0x3f800000, 0xbf800000,
               0x00000001, 0x80000001, 0x007fffff, 0x7f7fffff, 0x322bcc77, 0xb22bcc77]
    rows = []
    for i in range(len(special)):
        for j in range(len(special)):
            rows.append([special[i], special[j], special[(i + 3) % len(special)], special[(j + 5) % len(special)]])
    f = (rng.randn(400, 4) * np.exp(rng.uniform(-20, 20, size=(400, 4)))).astype(np.float32)
    rows += f.view(np.uint32).tolist()
    text = "\n".join(" ".join("%08x" % v for v in r) for r in rows) + "\n"
    out = subprocess.run([probe], input=text, capture_output=True, text=True, check=True).stdout.split("\n")
    res = [[int(x, 16) for x in line.split()] for line in out if line.strip()]
    assert len(res) == len(rows)
    # columns: 0 Inv4 1 RSqrt4 2 Min4 3 Max4 4 Cond4 | 5 invS 6 rsqrtS 7 minS 8 maxS 9 dot 10 cross.x
    keep = [[r[2], r[3], r[4], r[5], r[6], r[7], r[8], r[9], r[10]] for r in res]
    json.dump({"columns": ["Min4", "Max4", "Condition4(a<b,c,d)", "Inv(float)", "RSqrt(float)", "Min(float)", "Max(float)", "dot", "cross.x"],
               "inputs": rows, "outputs": keep,
               "note": "SSE Inv/RSqrt (rcpps/rsqrtps+NR) are CPU specific and are checked live against the probe instead"},
              open(os.path.join(HERE, "veclib_prims.json"), "w"))
    print("veclib_prims.json: %d rows" % len(rows))


def survey_digests():
    json.dump({
        "source": "SURVEY.md section 8(c) 'Observed' (reference run in the survey session; 1 thread, FPSCamera ang=pitch=0 far camera, useSah|noShadingData, 16x16 packets, default flip)",
        "box": {"res": [256, 256], "rays": 65536, "hits": 45369, "sum_id": 204078, "sum_t": 114934.835},
        "lancia": {"res": [1920, 1080], "hits": 81372, "sum_id": 532645744, "sum_t": 1310178.215, "nodes": 19785, "depth": 19, "tris": 30327},
        "feline": {"res": [1920, 1080], "hits_padded": 442567, "tris": 99732},
        "barracks": {"res": [1920, 1080], "hits": 373533},
    }, open(os.path.join(HERE, "survey_digests.json"), "w"), indent=1)


def oracle_frames():
    out = {}
    for name, resx, resy in (("box", 256, 256), ("atrium:0.05", 640, 368)):
        tv, hb, osc = util.scene_pair(name)
        cam = util.camera_for(name, tv)
        t, u, v, tid, st = osc.render_primary(cam.as_array13(), resx, resy, mode=O.MODE_IEEE)
        acc = osc.account_primary(cam.as_array13(), resx, resy, mode=O.MODE_IEEE)
        key = name.replace(":", "_").replace(".", "")
        out[key] = {"scene": name, "res": [resx, resy], "n_tris": int(len(tv)), "n_nodes": int(len(osc.nodes)), "depth": int(osc.depth),
                    "fnv_nodes": "%016x" % osc.fnv_nodes(), "fnv_tris": "%016x" % osc.fnv_tris(),
                    "hits": int(np.isfinite(t).sum()), "sum_id": int(tid.astype(np.int64).sum()),
                    "stats": [int(x) for x in st], "account": [int(x) for x in acc],
                    "sha_t": sha(t), "sha_u": sha(u), "sha_v": sha(v), "sha_id": sha(tid)}
        # every 16th pixel in both axes, full records
        np.savez_compressed(os.path.join(HERE, "oracle_%s_samples.npz" % key), t=t[::16, ::16], u=u[::16, ::16], v=v[::16, ::16], tid=tid[::16, ::16])
    json.dump(out, open(os.path.join(HERE, "oracle_frames.json"), "w"), indent=1)
    print("oracle_frames.json written")


def oracle_packets():
    """Whole-packet fixtures that carry their inputs: secondary-ray packets (all four template variants) and
    shadow packets on atrium:0.05, inputs + oracle IEEE outputs."""
    name = "atrium:0.05"
    tv, hb, osc = util.scene_pair(name)
    cam = util.camera_for(name, tv)
    data = {}
    for shared, masked in ((1, 0), (1, 1), (0, 0), (0, 1)):
        origin, dirs, idir, mask, dist, obj, bary = util.secondary_packets(osc, cam, 640, 368, 6, seed=100 + shared * 2 + masked, shared=bool(shared),
                                                                           masked=bool(masked), poison=(masked == 1))
        d2, o2, b2 = dist.copy(), obj.copy(), bary.copy()
        st = osc.trace_rays(origin, dirs, idir, mask, d2, o2, b2, 6, 64, bool(shared), mode=O.MODE_IEEE)
        k = "rays_s%d_m%d_" % (shared, masked)
        data.update({k + "origin": origin, k + "dir": dirs, k + "idir": idir, k + "dist_in": dist, k + "dist_out": d2, k + "obj_out": o2, k + "bary_out": b2,
                     k + "stats": st})
        if mask is not None:
            data[k + "mask"] = mask
    origin, dirs, idir, dist = util.shadow_packets(osc, 8, seed=77)
    d2 = dist.copy()
    st = osc.trace_shadow(origin, dirs, idir, d2, 8, 64)
    data.update({"shadow_origin": origin, "shadow_dir": dirs, "shadow_idir": idir, "shadow_dist_in": dist, "shadow_dist_out": d2, "shadow_stats": st})
    np.savez_compressed(os.path.join(HERE, "oracle_packets_atrium_005.npz"), **data)
    print("oracle_packets_atrium_005.npz written")


def whitted_lights(osc, cam, nl):
    bmin, bmax = osc.nodes[0]["bmin"], osc.nodes[0]["bmax"]
    c, e = (bmin + bmax) * 0.5, (bmax - bmin)
    return np.array([[c[0], c[1] + 0.35 * e[1], c[2], 1.0, 0.9, 0.8, 2.0 * float(e.max())],
                     [c[0] - 0.3 * e[0], c[1] + 0.1 * e[1], c[2] + 0.2 * e[2], 0.3, 0.5, 1.0, 0.6 * float(e.max())],
                     [cam.pos[0], cam.pos[1], cam.pos[2], 0.6, 0.6, 0.6, 0.25 * float(e.max())]], dtype=np.float32)[:nl]


def oracle_whitted():
    """BASELINE config 3 (primary + shadow packets, simple shading) on atrium:0.05: rgb8 frame digest + counters."""
    name, resx, resy = "atrium:0.05", 640, 368
    tv, hb, osc = util.scene_pair(name)
    cam = util.camera_for(name, tv)
    lights = whitted_lights(osc, cam, 2)
    frame, st = osc.render_whitted(cam.as_array13(), resx, resy, lights, mode=O.MODE_IEEE)
    json.dump({"scene": name, "res": [resx, resy], "lights": lights.tolist(), "sha_bgr": sha(frame), "stats": [int(x) for x in st],
               "mean_bgr": [float(x) for x in frame.reshape(-1, 3).mean(axis=0)]}, open(os.path.join(HERE, "oracle_whitted.json"), "w"), indent=1)
    np.savez_compressed(os.path.join(HERE, "oracle_whitted_samples.npz"), bgr=frame[::8, ::8])
    print("oracle_whitted.json written")
    # the same frame with the one-bounce reflections of gVals[7] (mirrored packets: per-ray origins, lane masks)
    frame, st = osc.render_whitted(cam.as_array13(), resx, resy, lights, mode=O.MODE_IEEE, reflections=True)
    json.dump({"scene": name, "res": [resx, resy], "lights": lights.tolist(), "sha_bgr": sha(frame), "stats": [int(x) for x in st],
               "mean_bgr": [float(x) for x in frame.reshape(-1, 3).mean(axis=0)]}, open(os.path.join(HERE, "oracle_whitted_refl.json"), "w"), indent=1)
    print("oracle_whitted_refl.json written")


def oracle_whitted_aa():
    """gVals[9], the tile renderer's 4x antialiasing (src/render.cpp:60-62, :71-110), on the frame of oracle_whitted.json at half its size:
    light pipeline, light pipeline + mirrored bounce, depth shading."""
    name, resx, resy = "atrium:0.05", 320, 192
    tv, hb, osc = util.scene_pair(name)
    cam = util.camera_for(name, tv)
    lights = whitted_lights(osc, cam, 2)
    out = {"scene": name, "res": [resx, resy], "lights": lights.tolist(), "modes": {}}
    for mode in ("lights", "refl", "depth"):
        frame, st = osc.render_whitted(cam.as_array13(), resx, resy, lights, mode=O.MODE_IEEE, reflections=mode == "refl", antialias=True, depth=mode == "depth")
        out["modes"][mode] = {"sha_bgr": sha(frame), "stats": [int(x) for x in st], "mean_bgr": [float(x) for x in frame.reshape(-1, 3).mean(axis=0)]}
    json.dump(out, open(os.path.join(HERE, "oracle_whitted_aa.json"), "w"), indent=1)
    print("oracle_whitted_aa.json written")


def oracle_transparency():
    name, resx, resy = "atrium:0.05", 320, 192
    tv, hb, osc = util.scene_pair(name)
    cam = util.camera_for(name, tv)
    xy, tp, ip, sel, lights = util.transparency_case(osc, cam, resx, resy, 5)
    out, st = osc.trace_transparency(cam.as_array13(), resx, resy, xy, tp, sel, lights)
    json.dump({"scene": name, "res": [resx, resy], "seed": 5, "sha_color": hashlib.sha256(out.tobytes()).hexdigest(), "stats": [int(x) for x in st]},
              open(os.path.join(HERE, "oracle_transparency.json"), "w"), indent=1)
    print("oracle_transparency.json", st)


if __name__ == "__main__":
    veclib_prims()
    veclib_exprs()
    survey_digests()
    oracle_frames()
    oracle_packets()
    oracle_whitted()
    oracle_whitted_aa()
    oracle_transparency()
