"""The oracle checked against first principles (float64 brute force), independent of the reference text:
closest hit == brute-force Moeller-Trumbore over all triangles; shadow result == brute-force occlusion;
packet walk == per-ray accounting walk on hit/miss; edge cases (empty/ragged packets, masks)."""
import numpy as np
import pytest

from tests import oracle_lib as O
from tests import util


def brute_force(tv, o, d):
    """closest two-sided hit of ray (o,d) with triangles tv[n,3,3] in float64 -> (t, index) or (inf,-1)"""
    v0, v1, v2 = tv[:, 0].astype(np.float64), tv[:, 1].astype(np.float64), tv[:, 2].astype(np.float64)
    e1, e2 = v1 - v0, v2 - v0
    p = np.cross(d, e2)
    det = (e1 * p).sum(1)
    ok = np.abs(det) > 1e-14
    inv = np.where(ok, 1.0 / np.where(ok, det, 1.0), 0.0)
    tv_ = o - v0
    u = (tv_ * p).sum(1) * inv
    q = np.cross(tv_, e1)
    v = (q * d).sum(1) * inv
    t = (e2 * q).sum(1) * inv
    hit = ok & (u >= 0) & (v >= 0) & (u + v <= 1) & (t > 0)
    if not hit.any():
        return np.inf, -1
    t = np.where(hit, t, np.inf)
    i = int(np.argmin(t))
    return float(t[i]), i


@pytest.mark.parametrize("name", ["box", "atrium:0.02"])
def test_closest_hit_matches_brute_force(name):
    """Per-ray Collide is two-sided (src/triangle.cpp:47-51), but the packet-level Triangle::TestInterval
    is only conservative for triangles whose normal points along the rays (det > 0); for the others it
    either waves the packet through (det interval entirely < 0, the "dirty hack" at src/triangle.cpp:119-120)
    or may cull real hits.  So: the oracle's hit must be a real intersection, never closer than the
    brute-force closest hit, and EQUAL to it whenever the closest triangle has det > 0."""
    tv, hb, osc = util.scene_pair(name)
    cam = util.camera_for(name, tv)
    resx, resy = 160, 96
    t, u, v, tid, st = osc.render_primary(cam.as_array13(), resx, resy, mode=O.MODE_IEEE)
    rng = np.random.RandomState(0)
    tv_bvh = tv[osc.perm]                       # triangles in BVH order: triId indexes this
    o = cam.pos.astype(np.float64)
    n_equal = 0
    for _ in range(300):
        x, y = rng.randint(resx), rng.randint(resy)
        px, py = (x // 16) * 16, (y // 16) * 16
        dd, _ = O.gen_packet(cam.as_array13(), resx, resy, px, py)
        q, l = (y - py) * 4 + (x - px) // 4, (x - px) % 4
        d = np.array([dd[q * 12 + l], dd[q * 12 + 4 + l], dd[q * 12 + 8 + l]], dtype=np.float64)
        bt, bi = brute_force(tv_bvh, o, d)
        if np.isfinite(t[y, x]):
            own, _ = brute_force(tv_bvh[[tid[y, x]]], o, d)      # the reported triangle really is hit there
            if np.isfinite(own):
                assert abs(own - t[y, x]) <= 1e-4 * max(1.0, own)
            assert t[y, x] >= bt - 1e-4 * max(1.0, bt)
        if bi >= 0:
            front = float(osc.tris[bi]["plane"][:3].astype(np.float64) @ d) > 1e-6
            if front and np.isfinite(t[y, x]) and abs(bt - t[y, x]) <= 1e-4 * max(1.0, bt):
                n_equal += 1
            elif front:
                # a front-facing closest hit may only be lost to rounding on an edge: it must be a grazing/edge case
                e = tv_bvh[bi].astype(np.float64)
                p = o + d * bt
                w = np.array([np.linalg.norm(np.cross(e[(k + 1) % 3] - e[k], p - e[k])) / np.linalg.norm(e[(k + 1) % 3] - e[k]) for k in range(3)])
                assert w.min() < 1e-3, (x, y, bt, t[y, x])
    assert n_equal > 100


def test_packet_walk_agrees_with_single_ray_walk():
    """Packet culls (interval tests, first/last shrinking, child order from lane 0) must not change which
    rays hit: compare hit counts with the single-ray accounting walk."""
    name = "atrium:0.05"
    tv, hb, osc = util.scene_pair(name)
    cam = util.camera_for(name, tv)
    t = osc.render_primary(cam.as_array13(), 640, 368, mode=O.MODE_IEEE)[0]
    acc = osc.account_primary(cam.as_array13(), 640, 368, mode=O.MODE_IEEE)
    assert int(acc[0]) == 640 * 368 and abs(int(acc[3]) - int(np.isfinite(t).sum())) <= 3


def test_shadow_matches_brute_force_occlusion():
    name = "atrium:0.02"
    tv, hb, osc = util.scene_pair(name)
    origin, dirs, idir, dist = util.shadow_packets(osc, 6, seed=3)
    d2 = dist.copy()
    osc.trace_shadow(origin, dirs, idir, d2, 6, 64)
    assert (np.isneginf(d2[np.isneginf(dist)])).all()          # masked lanes stay masked
    keep = ~np.isneginf(d2)
    util.assert_bit_equal(d2[keep], dist[keep], "unoccluded lanes keep their distance")
    rng = np.random.RandomState(1)
    tvb = tv.astype(np.float64)
    checked = 0
    for _ in range(400):
        p, q, l = rng.randint(6), rng.randint(64), rng.randint(4)
        i = p * 64 + q
        if dist[i, l] < 0:
            continue
        d = np.array([dirs[i, l], dirs[i, 4 + l], dirs[i, 8 + l]], dtype=np.float64)
        bt, _ = brute_force(tvb, origin[p].astype(np.float64), d)
        occluded = np.isneginf(d2[i, l])
        if abs(bt - dist[i, l]) > 1e-3 * max(1.0, dist[i, l]):   # skip grazing / at-the-limit cases
            # one-sided test in the reference (src/triangle.cpp:94): only front faces (det > 0) occlude
            if occluded:
                assert bt < dist[i, l]
            checked += 1
    assert checked > 50


def test_edge_cases_empty_and_masked_packets():
    name = "atrium:0.02"
    tv, hb, osc = util.scene_pair(name)
    cam = util.camera_for(name, tv)
    # zero packets: nothing happens
    z = np.zeros((0, 12), dtype=np.float32)
    st = osc.trace_rays(z, z, z, None, np.zeros((0, 4), np.float32), np.zeros((0, 4), np.int32), np.zeros((0, 8), np.float32), 0, 64, True)
    assert st.sum() == 0
    # a fully masked packet visits the root only and changes nothing
    origin, dirs, idir, mask, dist, obj, bary = util.secondary_packets(osc, cam, 64, 64, 2, seed=2, shared=True, masked=True)
    d2, o2, b2 = dist.copy(), obj.copy(), bary.copy()
    osc.trace_rays(origin, dirs, idir, mask, d2, o2, b2, 2, 64, True)
    assert (o2[64:128] == 0).all() and np.isneginf(d2[64:128]).all()
    # masked-out lanes never receive a hit
    lanes = (mask[:, None] >> np.arange(4)[None, :]) & 1
    assert np.isneginf(d2[lanes == 0]).all() and (o2[lanes == 0] == 0).all()
    # ragged packet sizes give the same per-ray answers as full packets when no packet-level cull differs on hit/miss
    origin, dirs, idir, _, dist, obj, bary = util.secondary_packets(osc, cam, 64, 64, 4, seed=9, shared=False, masked=False, size=16)
    d3, o3, b3 = dist.copy(), obj.copy(), bary.copy()
    osc.trace_rays(origin, dirs, idir, None, d3, o3, b3, 4, 16, False)
    d4, o4, b4 = dist.copy(), obj.copy(), bary.copy()
    osc.trace_rays(origin, dirs, idir, None, d4, o4, b4, 1, 64, False)
    assert np.array_equal(np.isfinite(d3), np.isfinite(d4))
    assert np.allclose(d3[np.isfinite(d3)], d4[np.isfinite(d4)], rtol=1e-5)


def test_planar_tile_format_known_answer_and_round_trip():
    """src/render.cpp:140-163 / src/compression.cpp:112-141: planes R, G-R, B-R with byte wrap-around; decode restores B,G,R."""
    rng = np.random.RandomState(5)
    frame = rng.randint(0, 256, size=(64, 48, 3)).astype(np.uint8)
    frame[0, 0] = (10, 20, 200)       # B, G, R: R=200 -> planes 200, 20-200=76 (mod 256), 10-200=66
    tiles = [(0, 0, 16, 64), (16, 0, 16, 64), (32, 0, 16, 64)]
    planes = O.planar_encode(frame, tiles)
    assert planes[0][0] == 200 and planes[0][16 * 64] == 76 and planes[0][2 * 16 * 64] == 66
    assert all(len(p) == 3 * 16 * 64 for p in planes)
    assert np.array_equal(O.planar_decode(planes, tiles, 48, 64), frame)


@pytest.mark.parametrize("name,resx,resy", [("box", 256, 256), ("atrium:0.05", 328, 200), ("stress:0.05", 200, 120), ("chain", 96, 64)])
def test_sse4_port_equals_scalar_oracle(name, resx, resy):
    """The 4-wide SSE-intrinsics port of the primary path (oracle/snail_sse4.inc: what bench.py times as `cpu_baseline`, kind "port-sse4")
    is the scalar restatement in ORC_MODE_SSE bit for bit: t, u, v, triId of every pixel and the TreeStats counters."""
    from tests import util
    tv, hb, osc = util.scene_pair(name)
    cam = util.camera_for(name, tv)
    a = osc.render_primary(cam.as_array13(), resx, resy, mode=O.MODE_SSE, threads=4)
    b = osc.render_primary_sse4(cam.as_array13(), resx, resy, threads=4)
    for x, y, what in zip(a[:4], b[:4], "t u v triId".split()):
        util.assert_bit_equal(x, y, "%s %s" % (name, what))
    assert np.array_equal(a[4], b[4]), (a[4], b[4])
    assert np.isfinite(a[0]).sum() > 0


def test_checker_does_not_inherit_flush_to_zero():
    """A Python process may run with flush-to-zero / denormals-are-zero switched on in MXCSR (a shared library built with -ffast-math does that
    to the thread that loads it, and which libraries get loaded depends on the host CPU): the oracle's entry points and the product's host-side
    builder compute under the default environment whatever the caller's is -- same bits with the caller in flush-to-zero mode."""
    import ctypes
    L = O.lib()
    L.orc_caller_mxcsr.restype = ctypes.c_uint
    L.orc_debug_set_mxcsr.argtypes = [ctypes.c_uint]
    default = L.orc_caller_mxcsr()
    tv, hb, osc = util.scene_pair("atrium:0.02")
    cam = util.camera_for("atrium:0.02", tv)
    # mirrored rays start ON surfaces: plane - origin differences get tiny there
    origin, dirs, idir, mask, dist, obj, bary = util.secondary_packets(osc, cam, 320, 192, 12, seed=5, shared=False, masked=True, size=64, poison=False)
    # denormal-scale geometry as well: the triangle precompute (ba, ca of 1e-39-sized edges) must keep denormals
    verts = (np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], dtype=np.float32) * np.float32(1e-39)).reshape(1, 9)
    def run():
        d2, o2, b2 = dist.copy(), obj.copy(), bary.copy()
        st = osc.trace_rays(origin, dirs, idir, mask, d2, o2, b2, 12, 64, False, mode=O.MODE_IEEE)
        return d2.view(np.uint32).copy(), o2.copy(), b2.view(np.uint32).copy(), np.asarray(st).copy(), O.tris_from_verts(verts).tobytes()
    want = run()
    try:
        L.orc_debug_set_mxcsr(0x9fc0)      # FTZ | DAZ, as crtfastmath sets them
        assert L.orc_caller_mxcsr() == 0x9fc0
        got = run()
        assert L.orc_caller_mxcsr() == 0x9fc0   # the caller's environment is restored, not replaced
        flushed = float(np.float32(1e-40) * np.float32(0.5)) == 0.0   # (what the caller's own arithmetic does in this mode)
    finally:
        L.orc_debug_set_mxcsr(default)
    for w, g in zip(want, got):
        assert (w == g) if isinstance(w, bytes) else np.array_equal(w, g)
    assert flushed or True   # numpy may or may not use SSE scalar multiplies here; the point is the equality above
