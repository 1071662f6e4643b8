"""GPU parity tests (run with -m gpu on an MI355X): the HIP path through the C-ABI vs the CPU oracle.

Bar (BASELINE.json north_star): triId bit-exact, t/u/v within 1e-4.  Because both sides implement the
two approximate operations of the path with veclib's scalar definitions (Inv = 1/x, RSqrt = 1/sqrt(x),
IEEE-rounded) and every other operation is a separately rounded fp32 mul/add/min/max, these tests hold
the HIP path to the stronger bar of BIT-EXACT t, u, v, triId and TreeStats counters against the oracle
in ORC_MODE_IEEE.  The oracle's ORC_MODE_SSE (what the reference executes on x86: rcpps/rsqrtps + one
Newton step) is compared with tolerance 1e-4 in test_sse_mode_tolerance."""
import numpy as np
import pytest

from tests import oracle_lib as O
from tests import util

pytestmark = pytest.mark.gpu

TOL = 1e-4   # north_star tolerance on t/u/v (relative to max(1,|t|))


@pytest.fixture(scope="module")
def torch_mod():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch


def gpu_scene(name):
    from snail_amd.scene import Scene
    tv, hbvh, osc = util.scene_pair(name)
    return tv, Scene(hbvh, 0), osc


def compare_frames(frame, oracle_out, what):
    t, u, v, tid, _ = oracle_out
    gt, gu, gv, gid = (x.cpu().numpy() for x in (frame.t, frame.u, frame.v, frame.tri_id))
    util.assert_bit_equal(gid, tid, what + " triId")
    util.assert_bit_equal(gt, t, what + " t")
    util.assert_bit_equal(gu, u, what + " u")
    util.assert_bit_equal(gv, v, what + " v")


@pytest.mark.parametrize("name,resx,resy", [
    ("box", 256, 256),            # BASELINE config 0
    ("box", 250, 130),            # ragged: not a multiple of 16 (and resx % 4 != 0 -> scalar stores)
    ("atrium:0.05", 640, 368),
    ("atrium:0.05", 328, 200),
])
def test_primary_frame_bit_exact(torch_mod, name, resx, resy):
    tv, sc, osc = gpu_scene(name)
    cam = util.camera_for(name, tv)
    stats = sc.new_stats()
    frame = sc.trace_primary(cam, resx, resy, stats=stats)
    torch_mod.cuda.synchronize()
    ref = osc.render_primary(cam.as_array13(), resx, resy, mode=O.MODE_IEEE)
    compare_frames(frame, ref, "%s %dx%d" % (name, resx, resy))
    assert np.array_equal(stats.cpu().numpy().astype(np.uint64), ref[4]), (stats.cpu().numpy(), ref[4])
    assert np.isfinite(ref[0]).sum() > 0
    sc.close()


def test_full_size_frame_bit_exact_and_properties(torch_mod):
    """BASELINE config 1 size: atrium (sponza stand-in, ~263 K triangles) at 1920x1080."""
    name = "atrium"
    tv, sc, osc = gpu_scene(name)
    cam = util.camera_for(name, tv)
    stats = sc.new_stats()
    frame = sc.trace_primary(cam, 1920, 1080, stats=stats)
    torch_mod.cuda.synchronize()
    ref = osc.render_primary(cam.as_array13(), 1920, 1080, mode=O.MODE_IEEE, threads=16)
    compare_frames(frame, ref, "atrium 1920x1080")
    st = stats.cpu().numpy()
    assert np.array_equal(st.astype(np.uint64), ref[4])
    assert st[2] == 1920 * 1088                       # padded to whole packets (src/render.cpp:67-68)
    # size-independent properties
    t = frame.t.cpu().numpy(); tid = frame.tri_id.cpu().numpy()
    u = frame.u.cpu().numpy(); v = frame.v.cpu().numpy()
    hit = np.isfinite(t)
    assert hit.mean() > 0.9                            # interior camera: almost every pixel hits
    assert (tid[~hit] == 0).all() and (u[~hit] == 0).all() and (v[~hit] == 0).all()
    assert (t[hit] > 0).all() and (tid[hit] >= 0).all() and (tid[hit] < len(tv)).all()
    assert (u[hit] >= -1e-4).all() and (v[hit] >= -1e-4).all() and ((u + v)[hit] <= 1 + 1e-4).all()
    # idempotence: a second launch into the same buffers gives the same bytes
    frame2 = sc.trace_primary(cam, 1920, 1080, out=frame)
    torch_mod.cuda.synchronize()
    util.assert_bit_equal(frame2.t.cpu().numpy(), t, "idempotence")
    # hit points lie on their triangle's plane: |n.(o + t d) - n.a| small
    hb = util.scene_pair(name)[1]
    ys, xs = np.nonzero(hit)
    sel = np.random.RandomState(0).choice(len(ys), 4096, replace=False)
    d, _ = O.gen_packet(cam.as_array13(), 1920, 1080, 0, 0)   # just to exercise the API; directions rebuilt below
    for i in sel[:256]:
        y, x = ys[i], xs[i]
        px, py = (x // 16) * 16, (y // 16) * 16
        dd, _ = O.gen_packet(cam.as_array13(), 1920, 1080, int(px), int(py))
        q = (y - py) * 4 + (x - px) // 4
        l = (x - px) % 4
        dirv = np.array([dd[q * 12 + l], dd[q * 12 + 4 + l], dd[q * 12 + 8 + l]], dtype=np.float64)
        P = cam.pos.astype(np.float64) + dirv * float(t[y, x])
        pl = hb.tris[tid[y, x]]["plane"].astype(np.float64)
        assert abs(pl[:3] @ P - pl[3]) < 1e-3 * max(1.0, abs(pl[3]))
    sc.close()


def test_rect_and_packet_list_match_full_frame(torch_mod):
    name = "atrium:0.05"
    tv, sc, osc = gpu_scene(name)
    cam = util.camera_for(name, tv)
    resx, resy = 640, 368
    full = sc.trace_primary(cam, resx, resy)
    # rect: only the rect is written, the rest keeps its previous contents
    part = sc.alloc_frame(resx, resy)
    part.t.fill_(-7.0)
    rect = (160, 64, 320, 128)
    sc.trace_primary(cam, resx, resy, rect=rect, out=part)
    torch_mod.cuda.synchronize()
    x0, y0, w, h = rect
    assert torch_mod.equal(part.t[y0:y0 + h, x0:x0 + w], full.t[y0:y0 + h, x0:x0 + w])
    assert torch_mod.equal(part.tri_id[y0:y0 + h, x0:x0 + w], full.tri_id[y0:y0 + h, x0:x0 + w])
    outside = part.t.clone(); outside[y0:y0 + h, x0:x0 + w] = -7.0
    assert (outside == -7.0).all()
    # packet list in a shuffled order + scatter == full frame
    xs, ys = np.meshgrid(np.arange(0, resx, 16), np.arange(0, resy, 16))
    xy = np.stack([xs.ravel(), ys.ravel()], axis=1).astype(np.int32)
    np.random.RandomState(3).shuffle(xy)
    dxy = torch_mod.from_numpy(xy).cuda()
    planes = sc.trace_packets(cam, resx, resy, dxy)
    fr = sc.alloc_frame(resx, resy)
    sc.packets_to_frame(dxy, planes, fr)
    torch_mod.cuda.synchronize()
    for a, b in ((fr.t, full.t), (fr.u, full.u), (fr.v, full.v), (fr.tri_id, full.tri_id)):
        assert torch_mod.equal(a, b)
    # packet-major layout is the reference's quad order: quad ty*4+k = pixels x+4k.. of row y+ty
    p0 = planes[0][0].cpu().numpy().reshape(16, 4, 4)
    x, y = xy[0]
    np.testing.assert_array_equal(p0.reshape(16, 16), full.t[y:y + 16, x:x + 16].cpu().numpy())
    sc.close()


def test_host_pointer_entry_point(torch_mod):
    name = "box"
    tv, sc, osc = gpu_scene(name)
    cam = util.camera_for(name, tv)
    t, u, v, tid, stats = sc.trace_primary_host(cam, 256, 256)
    ref = osc.render_primary(cam.as_array13(), 256, 256, mode=O.MODE_IEEE)
    util.assert_bit_equal(t, ref[0], "t"); util.assert_bit_equal(tid, ref[3], "id")
    util.assert_bit_equal(u, ref[1], "u"); util.assert_bit_equal(v, ref[2], "v")
    assert np.array_equal(stats, ref[4])
    sc.close()


@pytest.mark.parametrize("shared,masked,size,poison", [
    (True, False, 64, False), (True, True, 64, False), (False, False, 64, False), (False, True, 64, False),
    (False, True, 16, False), (True, False, 1, False),
    (True, False, 64, True), (False, True, 64, True),       # non-finite idir -> EXACT instantiation
])
def test_trace_rays_bit_exact(torch_mod, shared, masked, size, poison):
    name = "atrium:0.05"
    tv, sc, osc = gpu_scene(name)
    cam = util.camera_for(name, tv)
    npk = 24
    origin, dirs, idir, mask, dist, obj, bary = util.secondary_packets(osc, cam, 640, 368, npk, seed=11 + size, shared=shared, masked=masked,
                                                                       size=size, poison=poison)
    d2, o2, b2 = dist.copy(), obj.copy(), bary.copy()
    ost = osc.trace_rays(origin, dirs, idir, mask, d2, o2, b2, npk, size, shared, mode=O.MODE_IEEE)
    from snail_amd.scene import Context
    tt = torch_mod.from_numpy
    ctx = Context(tt(origin).cuda(), tt(dirs).cuda(), tt(idir).cuda(), tt(dist.copy()).cuda(), tt(obj.copy()).cuda(), tt(bary.copy()).cuda(),
                  size=size, shared_origin=shared, mask=None if mask is None else tt(mask).cuda())
    stats = sc.new_stats()
    sc.traverse_primary(ctx, stats=stats)
    torch_mod.cuda.synchronize()
    util.assert_bit_equal(ctx.object.cpu().numpy(), o2, "object")
    util.assert_bit_equal(ctx.distance.cpu().numpy(), d2, "distance")
    util.assert_bit_equal(ctx.barycentric.cpu().numpy(), b2, "barycentric")
    st = stats.cpu().numpy().astype(np.uint64)
    assert st[0] == ost[0] and st[1] == ost[1], (st, ost)
    assert (o2 != 0).any() or poison
    # host-pointer variant gives the same bytes
    d3, o3, b3 = dist.copy(), obj.copy(), bary.copy()
    sc.trace_rays_host(origin, dirs, idir, mask, d3, o3, b3, npk, size, shared)
    util.assert_bit_equal(d3, d2, "host distance"); util.assert_bit_equal(o3, o2, "host object")
    sc.close()


@pytest.mark.parametrize("size", [64, 16])
def test_trace_shadow_bit_exact(torch_mod, size):
    name = "atrium:0.05"
    tv, sc, osc = gpu_scene(name)
    npk = 32
    origin, dirs, idir, dist = util.shadow_packets(osc, npk, seed=5, size=size)
    d2 = dist.copy()
    ost = osc.trace_shadow(origin, dirs, idir, d2, npk, size)
    from snail_amd.scene import ShadowContext
    tt = torch_mod.from_numpy
    ctx = ShadowContext(tt(origin).cuda(), tt(dirs).cuda(), tt(idir).cuda(), tt(dist.copy()).cuda(), size=size)
    stats = sc.new_stats()
    sc.traverse_shadow(ctx, stats=stats)
    torch_mod.cuda.synchronize()
    util.assert_bit_equal(ctx.distance.cpu().numpy(), d2, "shadow distance")
    st = stats.cpu().numpy().astype(np.uint64)
    assert st[0] == ost[0] and st[1] == ost[1] and st[3] == ost[3], (st, ost)
    occl = np.isneginf(d2) & ~np.isneginf(dist)
    assert occl.any() and (~np.isneginf(d2)).any()          # both outcomes are exercised
    d3 = dist.copy()
    sc.trace_shadow_host(origin, dirs, idir, d3, npk, size)
    util.assert_bit_equal(d3, d2, "host shadow distance")
    sc.close()


def test_accounting_walk_matches_oracle(torch_mod):
    name = "atrium:0.05"
    tv, sc, osc = gpu_scene(name)
    cam = util.camera_for(name, tv)
    got = sc.account_primary(cam, 640, 368)
    want = osc.account_primary(cam.as_array13(), 640, 368, mode=O.MODE_IEEE)
    assert np.array_equal(got, want), (got, want)
    sc.close()


def test_sse_mode_tolerance(torch_mod):
    """HIP (IEEE Inv/RSqrt) vs the oracle in SSE mode (rcpps/rsqrtps + Newton, as the reference runs on
    x86): triId equal except where two candidate hits are closer than the tolerance (tie rule, SURVEY
    section 7), t/u/v within 1e-4."""
    name = "atrium:0.05"
    tv, sc, osc = gpu_scene(name)
    cam = util.camera_for(name, tv)
    frame = sc.trace_primary(cam, 640, 368)
    torch_mod.cuda.synchronize()
    t, u, v, tid, _ = osc.render_primary(cam.as_array13(), 640, 368, mode=O.MODE_SSE)
    gt, gu, gv, gid = (x.cpu().numpy() for x in (frame.t, frame.u, frame.v, frame.tri_id))
    assert np.array_equal(np.isfinite(gt), np.isfinite(t))
    hit = np.isfinite(t)
    scale = np.maximum(1.0, np.abs(t[hit]))
    assert (np.abs(gt[hit] - t[hit]) <= TOL * scale).all()
    same = gid == tid
    assert same[hit].mean() > 0.999
    assert (np.abs(gu - u)[hit & same] <= TOL).all() and (np.abs(gv - v)[hit & same] <= TOL).all()
    sc.close()


def test_invalid_arguments_fail_loudly(torch_mod):
    from snail_amd import SnailError
    name = "box"
    tv, sc, osc = gpu_scene(name)
    cam = util.camera_for(name, tv)
    with pytest.raises(SnailError):
        sc.trace_primary(cam, 256, 256, rect=(8, 0, 64, 64))      # rect origin not on the packet grid
    with pytest.raises(SnailError):
        sc.trace_primary(cam, 0, 256)
    sc.close()
