"""The C-ABI handle under the reference's concurrency contract (run with -m gpu on an MI355X).

The reference hands ONE `const Scene<AccStruct>` to `threads` pthread workers, each of which calls TraversePrimary / TraverseShadow on it
(src/render.cpp:214-267, src/thread_pool.cpp:151-180, src/scene_trace.cpp:119-120,:560-563); include/snail_hip.h ("Concurrency") promises
the same of a SnailScene.  Here several host threads are inside the library on one handle at once -- ctypes releases the GIL for the
duration of a foreign call -- and every result and every TreeStats total must be the oracle's, whatever the interleaving."""
import threading

import numpy as np
import pytest

from snail_amd import FPSCamera, scenes
from tests import oracle_lib as O
from tests import util

pytestmark = pytest.mark.gpu

THREADS = 8


@pytest.fixture(scope="module")
def torch_mod():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch


def run_threads(fn, n=THREADS):
    """fn(k) on n threads started together; re-raises the first failure."""
    errs = [None] * n
    gate = threading.Barrier(n)

    def body(k):
        try:
            gate.wait()
            fn(k)
        except BaseException as e:  # noqa: BLE001 - reported below
            errs[k] = e

    ts = [threading.Thread(target=body, args=(k,)) for k in range(n)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    for e in errs:
        if e is not None:
            raise e


@pytest.mark.parametrize("host_sse", [False, True])
def test_host_pointer_entry_points_from_eight_threads(torch_mod, host_sse):
    """snail_trace_shadow / snail_trace_rays (one packet per call = the adapter's immediate path, and batched), snail_trace_primary of a rect,
    snail_account_primary and snail_render_image, all at once from 8 threads on ONE scene: each call returns the
    oracle's bits and its OWN counters (the summed TreeStats equal the oracle's)."""
    from snail_amd.scene import Scene
    MODE = O.MODE_SSE if host_sse else O.MODE_IEEE
    name = "atrium:0.05"
    tv, hb, osc = util.scene_pair(name)
    cam = util.camera_for(name, tv)
    sc = Scene(hb, 0)
    if host_sse:
        sc.set_arith("host_sse")
    resx, resy = 320, 192
    n_sh, n_ry = 4 * THREADS + 3, 2 * THREADS + 1
    so, sd, si, sdist = util.shadow_packets(osc, n_sh, 171)
    ro, rd, ri, rmask, rdist, robj, rbary = util.secondary_packets(osc, cam, resx, resy, n_ry, 172, shared=False, masked=True)
    # expectations, serially, from the oracle
    want_sh = sdist.copy()
    sh_stats = [osc.trace_shadow(so[p:p + 1], sd[p * 64:(p + 1) * 64], si[p * 64:(p + 1) * 64], want_sh[p * 64:(p + 1) * 64], 1, 64, mode=MODE) for p in range(n_sh)]
    wd, wo, wb = rdist.copy(), robj.copy(), rbary.copy()
    ry_stats = [osc.trace_rays(ro[p * 64:(p + 1) * 64], rd[p * 64:(p + 1) * 64], ri[p * 64:(p + 1) * 64], rmask[p * 64:(p + 1) * 64], wd[p * 64:(p + 1) * 64],
                               wo[p * 64:(p + 1) * 64], wb[p * 64:(p + 1) * 64], 1, 64, False, mode=MODE) for p in range(n_ry)]
    ref = osc.render_primary(cam.as_array13(), resx, resy, mode=MODE)
    rects = [(0, 0, 160, 96), (160, 0, 160, 96), (0, 96, 160, 96), (160, 96, 160, 96), (0, 0, 320, 64), (0, 64, 320, 128), (16, 16, 48, 160), (64, 32, 256, 128)]
    rect_stats = [osc.render_primary(cam.as_array13(), resx, resy, rect=r, mode=MODE, threads=1)[4] for r in rects]
    want_img = O.shade_depth(ref[0], mode=MODE).reshape(resy, resx, 3)
    acct = sc.account_primary(cam, resx, resy)            # (device accounting walk, checked against the oracle's elsewhere)
    got_sh, got_d, got_o, got_b = sdist.copy(), rdist.copy(), robj.copy(), rbary.copy()
    sh_got_stats, ry_got_stats = [None] * n_sh, [None] * n_ry
    rounds = 3

    def body(k):
        for rnd in range(rounds):
            for p in range(k, n_sh, THREADS):                 # one synchronous call per 256-ray packet: HipBVH::TraverseShadow
                d = sdist[p * 64:(p + 1) * 64].copy()
                st = sc.trace_shadow_host(so[p:p + 1], sd[p * 64:(p + 1) * 64], si[p * 64:(p + 1) * 64], d, 1, 64)
                got_sh[p * 64:(p + 1) * 64] = d
                sh_got_stats[p] = st
            for p in range(k, n_ry, THREADS):                 # HipBVH::TraversePrimary(Context<0,1>)
                d, o, b = rdist[p * 64:(p + 1) * 64].copy(), robj[p * 64:(p + 1) * 64].copy(), rbary[p * 64:(p + 1) * 64].copy()
                st = sc.trace_rays_host(ro[p * 64:(p + 1) * 64], rd[p * 64:(p + 1) * 64], ri[p * 64:(p + 1) * 64], rmask[p * 64:(p + 1) * 64], d, o, b, 1, 64, False)
                got_d[p * 64:(p + 1) * 64], got_o[p * 64:(p + 1) * 64], got_b[p * 64:(p + 1) * 64] = d, o, b
                ry_got_stats[p] = st
            # a rect of the frame through the host-buffer entry point, this thread's own rect and counters
            x0, y0, w, h = rects[k]
            t, u, v, tid, st = sc.trace_primary_host(cam, resx, resy, rect=rects[k])
            util.assert_bit_equal(t[y0:y0 + h, x0:x0 + w], ref[0][y0:y0 + h, x0:x0 + w], "thread %d rect t" % k)
            util.assert_bit_equal(tid[y0:y0 + h, x0:x0 + w], ref[3][y0:y0 + h, x0:x0 + w], "thread %d rect triId" % k)
            util.assert_bit_equal(u[y0:y0 + h, x0:x0 + w], ref[1][y0:y0 + h, x0:x0 + w], "thread %d rect u" % k)
            assert st.tolist() == [int(x) for x in rect_stats[k]], (k, st, rect_stats[k])
            if k % 2 == 0:                                    # the image renderer (calls on one handle take turns; still each its own bytes and counters)
                img, ist = sc.render_image_host(cam, resx, resy, flags=sc.RENDER_DEPTH)
                assert np.array_equal(img, want_img), "thread %d image" % k
                assert ist.tolist() == [int(x) for x in ref[4]]
            else:                                             # the all-batched shadow call + the accounting walk
                d = sdist.copy()
                st = sc.trace_shadow_host(so, sd, si, d, n_sh, 64)
                util.assert_bit_equal(d, want_sh, "thread %d shadow batch" % k)
                assert [int(st[0]), int(st[1]), int(st[3])] == [sum(int(s[i]) for s in sh_stats) for i in (0, 1, 3)]
                assert sc.account_primary(cam, resx, resy).tolist() == acct.tolist()

    run_threads(body)
    util.assert_bit_equal(got_sh, want_sh, "shadow packets, 8 threads")
    util.assert_bit_equal(got_d, wd, "secondary packets, 8 threads, dist")
    util.assert_bit_equal(got_o, wo, "secondary packets, 8 threads, obj")
    util.assert_bit_equal(got_b, wb, "secondary packets, 8 threads, bary")
    for p in range(n_sh):
        assert [int(sh_got_stats[p][i]) for i in (0, 1, 3)] == [int(sh_stats[p][i]) for i in (0, 1, 3)], ("shadow stats", p)
    for p in range(n_ry):
        assert [int(x) for x in ry_got_stats[p][:2]] == [int(x) for x in ry_stats[p][:2]], ("ray stats", p)
    sc.close()


def test_dev_entry_points_from_threads_on_their_own_streams(torch_mod):
    """Four host threads, each with its own HIP stream and its own moving camera and light, issue *_dev launches on ONE scene with no
    synchronisation between them: more distinct origins than the origin-relative node cache holds (40), the 8 scratch slots recycled across
    threads, staged light frames beside plain primary frames.  Every frame equals what the same call gives alone afterwards -- and the first
    frame of each thread the oracle's."""
    torch = torch_mod
    from snail_amd.scene import Scene
    name = "atrium:0.05"
    tv, hb, osc = util.scene_pair(name)
    sc = Scene(hb, 0)
    resx, resy = 320, 192
    pos, ang, pitch = scenes.atrium_camera()
    bmin, bmax = osc.nodes[0]["bmin"], osc.nodes[0]["bmax"]
    c, e = (bmin + bmax) * 0.5, (bmax - bmin)
    nthreads, frames = 4, 12

    def cam_of(k, f):
        p = np.asarray(pos, dtype=np.float32) + np.float32(0.01) * np.array([f * (k + 1), 0.3 * k, -f], dtype=np.float32)
        return FPSCamera(p, ang + 0.02 * k, pitch).camera()

    def light_of(k, f):
        return np.array([[c[0] + 0.01 * f * (k + 1), c[1] + 0.35 * e[1], c[2] + 0.02 * k, 1.0, 0.9, 0.8, 2.0 * float(e.max())]], dtype=np.float32)

    outs = [[None] * frames for _ in range(nthreads)]
    lit = [[None] * frames for _ in range(nthreads)]
    stats = [sc.new_stats() for _ in range(nthreads)]

    def body(k):
        st = torch.cuda.Stream()
        for f in range(frames):
            with torch.cuda.stream(st):      # (the buffers' initial fills on the launches' stream: torch's side streams do not wait for the null stream)
                fr = sc.alloc_frame(resx, resy)
                img = torch.zeros((resy, resx, 3), dtype=torch.uint8, device="cuda")
            outs[k][f] = sc.trace_primary(cam_of(k, f), resx, resy, out=fr, stats=stats[k], stream=st)
            lit[k][f] = sc.render_whitted(cam_of(k, f), resx, resy, light_of(k, f), out=img, stream=st, reflections=(f % 4 == 3))
        st.synchronize()

    run_threads(body, nthreads)
    torch.cuda.synchronize()
    for k in range(nthreads):
        alone_stats = sc.new_stats()
        for f in range(frames):
            alone = sc.trace_primary(cam_of(k, f), resx, resy, stats=alone_stats)
            alone_lit = sc.render_whitted(cam_of(k, f), resx, resy, light_of(k, f), reflections=(f % 4 == 3))
            torch.cuda.synchronize()
            for a, b, n in ((outs[k][f].t, alone.t, "t"), (outs[k][f].u, alone.u, "u"), (outs[k][f].v, alone.v, "v"), (outs[k][f].tri_id, alone.tri_id, "triId")):
                util.assert_bit_equal(a.cpu().numpy(), b.cpu().numpy(), "thread %d frame %d %s" % (k, f, n))
            assert torch.equal(lit[k][f], alone_lit), "thread %d lit frame %d" % (k, f)
        assert torch.equal(stats[k], alone_stats), (k, stats[k], alone_stats)
        ref = osc.render_primary(cam_of(k, 0).as_array13(), resx, resy, mode=O.MODE_IEEE)
        util.assert_bit_equal(outs[k][0].t.cpu().numpy(), ref[0], "thread %d frame 0 vs the oracle" % k)
        util.assert_bit_equal(outs[k][0].tri_id.cpu().numpy(), ref[3], "thread %d frame 0 triId vs the oracle" % k)
        want, _ = osc.render_whitted(cam_of(k, 0).as_array13(), resx, resy, light_of(k, 0), mode=O.MODE_IEEE, reflections=False)
        assert np.array_equal(lit[k][0].cpu().numpy(), want), "thread %d lit frame 0 vs the oracle" % k
    sc.close()
