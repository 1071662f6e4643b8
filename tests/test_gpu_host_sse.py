"""GPU parity tests of SNAIL_ARITH_HOST_SSE (run with -m gpu on an MI355X): the HIP path in the arithmetic the reference's x86 build
executes -- veclib's SSE Inv / RSqrt / FastInv = rcpps / rsqrtps of the HOST CPU + one Newton step (veclib/sse/base.h:84-92,
veclib/sse/f32.h:98-102) -- against the oracle in ORC_MODE_SSE, which runs those very instructions on the same CPU.

Bar: BIT-EXACT t, u, v, triId, TreeStats counters and picture bytes.  north_star asks for triId bit-exact and t/u/v within 1e-4 against
"the reference CPU veclib/SSE path"; in this mode nothing is left to a tolerance: device-generated rays included."""
import os

import numpy as np
import pytest

from tests import oracle_lib as O
from tests import util
from tests.test_gpu_parity import compare_frames, gpu_scene

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_mod():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch


def sse_scene(name):
    tv, sc, osc = gpu_scene(name)
    sc.set_arith("host_sse")
    assert sc.arith() == "host_sse"
    return tv, sc, osc


def lights_for(osc, cam, n):
    bmin, bmax = osc.nodes[0]["bmin"], osc.nodes[0]["bmax"]
    c, e = (bmin + bmax) * 0.5, (bmax - bmin)
    return np.array([[c[0], c[1] + 0.35 * e[1], c[2], 1.0, 0.9, 0.8, 2.0 * float(e.max())],
                     [c[0] - 0.3 * e[0], c[1] + 0.1 * e[1], c[2] + 0.2 * e[2], 0.3, 0.5, 1.0, 0.6 * float(e.max())],
                     [cam.pos[0], cam.pos[1], cam.pos[2], 0.6, 0.6, 0.6, 0.25 * float(e.max())]], dtype=np.float32)[:n]


def test_device_reproduces_the_hosts_rcpps_and_rsqrtps_for_every_float(torch_mod):
    """dev_sse::rcpHost / rsqrtHost over all 2^32 bit patterns against the instructions of this box's CPU (checksums per 65536 inputs)."""
    import ctypes as C
    from snail_amd._lib import check, debug_lib
    L = debug_lib()
    for fn in (0, 1):
        bad, first = C.c_uint64(1), C.c_uint32(0)
        check(L.snail_debug_hostsse_device_check(fn, min(16, os.cpu_count() or 1), C.byref(bad), C.byref(first)), "snail_debug_hostsse_device_check")
        assert bad.value == 0, (fn, bad.value, hex(first.value << 16))


@pytest.mark.parametrize("name,resx,resy", [
    ("box", 256, 256),            # BASELINE config 0
    ("box", 250, 130),            # ragged
    ("atrium:0.05", 640, 368),
    ("chain", 256, 144),          # depth-63 tree: the DEEP kernels of the second arithmetic
    ("stress:0.05", 320, 192),
    ("patches", 200, 120),        # the narrow-range leaf forms at every width
    ("offgrid", 640, 400),        # geometry off every grid (scenes.offgrid), far camera
    ("offgrid-in", 328, 200),     # ... and from inside
])
def test_primary_frame_bit_exact_in_host_sse(torch_mod, name, resx, resy):
    tv, sc, osc = sse_scene(name)
    cam = util.camera_for(name, tv)
    stats = sc.new_stats()
    frame = sc.trace_primary(cam, resx, resy, stats=stats)
    torch_mod.cuda.synchronize()
    ref = osc.render_primary(cam.as_array13(), resx, resy, mode=O.MODE_SSE)
    compare_frames(frame, ref, "host_sse %s %dx%d" % (name, resx, resy))
    assert np.array_equal(stats.cpu().numpy().astype(np.uint64), ref[4]), (stats.cpu().numpy(), ref[4])
    # the two arithmetics are different functions: the IEEE frame of the same scene handle differs in some bits, and switching back restores it
    sc.set_arith("ieee")
    f2 = sc.trace_primary(cam, resx, resy)
    torch_mod.cuda.synchronize()
    compare_frames(f2, osc.render_primary(cam.as_array13(), resx, resy, mode=O.MODE_IEEE), "back to ieee")
    if name != "box":
        assert not np.array_equal(f2.t.cpu().numpy().view(np.uint32), ref[0].view(np.uint32))
    sc.close()


@pytest.mark.parametrize("name", ["atrium", "stress"])
def test_full_size_frame_bit_exact_in_host_sse(torch_mod, name):
    """BASELINE configs 1 and 5 at 1920x1080, rays generated on the device: every hit record and the TreeStats equal the oracle's in
    ORC_MODE_SSE bit for bit -- no hit/miss flip, no triId mismatch, no pixel outside any tolerance."""
    tv, sc, osc = sse_scene(name)
    cam = util.camera_for(name, tv)
    stats = sc.new_stats()
    frame = sc.trace_primary(cam, 1920, 1080, stats=stats)
    torch_mod.cuda.synchronize()
    ref = osc.render_primary(cam.as_array13(), 1920, 1080, mode=O.MODE_SSE, threads=16)
    compare_frames(frame, ref, "host_sse %s 1920x1080" % name)
    assert np.array_equal(stats.cpu().numpy().astype(np.uint64), ref[4]), (stats.cpu().numpy(), ref[4])
    assert np.isfinite(ref[0]).mean() > 0.5
    sc.close()


@pytest.mark.parametrize("shared,masked,size,poison", [(True, False, 64, False), (False, True, 64, False), (False, False, 23, False), (True, True, 16, False),
                                                       (True, False, 64, True), (False, True, 64, True)])
def test_generic_packets_bit_exact_in_host_sse(torch_mod, shared, masked, size, poison):
    """TraversePrimary<so,mask> with the caller's rays (dir / idir as the reference's SSE code hands them over): 1 / det of the accepted
    hits is the only approximate operation left, and it is the host's."""
    name = "atrium:0.05"
    tv, sc, osc = sse_scene(name)
    cam = util.camera_for(name, tv)
    npk = 24
    origin, dirs, idir, mask, dist, obj, bary = util.secondary_packets(osc, cam, 640, 368, npk, seed=11 + size, shared=shared, masked=masked, size=size, coherent=size == 64,
                                                                       poison=poison)
    if poison:   # what the reference's SSE SafeInv hands over at its singularity: rcpps(0) = inf, and the Newton step makes it NaN (0 * inf) -> the M_EXACT kernels of dev_sse
        idir[~np.isfinite(idir)] = np.nan
        idir[3 * size + 1, 0:4] = np.nan
    d2, o2, b2 = dist.copy(), obj.copy(), bary.copy()
    st2 = osc.trace_rays(origin, dirs, idir, mask, d2, o2, b2, npk, size, shared, mode=O.MODE_SSE)
    d3, o3, b3 = dist.copy(), obj.copy(), bary.copy()
    sc.trace_rays_host(origin, dirs, idir, mask, d3, o3, b3, npk, size, shared)
    util.assert_bit_equal(d3, d2, "t"); util.assert_bit_equal(o3, o2, "triId"); util.assert_bit_equal(b3, b2, "barycentric")
    assert (o2 != 0).any() and st2[0] > 0
    # ... and it is not the IEEE result (some accepted hit's 1 / det differs in its last bits)
    d4, o4, b4 = dist.copy(), obj.copy(), bary.copy()
    osc.trace_rays(origin, dirs, idir, mask, d4, o4, b4, npk, size, shared, mode=O.MODE_IEEE)
    assert not np.array_equal(d4.view(np.uint32), d2.view(np.uint32))
    sc.close()


@pytest.mark.parametrize("name,resx,resy,nl,refl", [("atrium:0.05", 640, 368, 2, False), ("atrium:0.05", 640, 368, 2, True), ("box", 256, 256, 1, True),
                                                     ("stress:0.05", 320, 192, 3, True), ("chain", 128, 96, 1, True), ("atrium:0.05", 250, 130, 0, False)])
def test_whitted_frames_byte_exact_in_host_sse(torch_mod, name, resx, resy, nl, refl):
    """Config 3's pipeline (+ the mirrored bounce) in the second arithmetic: shadow-ray generation (Inv(distance), SafeInv), the mirrored
    rays' SafeInv, the attenuation's FastInv = raw rcpps, 1 / det in every walk."""
    tv, sc, osc = sse_scene(name)
    cam = util.camera_for(name, tv)
    lights = lights_for(osc, cam, nl)
    want, wst = osc.render_whitted(cam.as_array13(), resx, resy, lights, mode=O.MODE_SSE, reflections=refl)
    stats = sc.new_stats()
    got = sc.render_whitted(cam, resx, resy, lights, stats=stats, reflections=refl).cpu().numpy()
    assert np.array_equal(got, want), (int((got != want).sum()), got.shape)
    assert np.array_equal(stats.cpu().numpy().astype(np.uint64), wst), (stats.cpu().numpy(), wst)
    assert want.max() > 0
    sc.close()


@pytest.mark.parametrize("refl", [False, True])
def test_config3_full_size_frame_byte_exact_in_host_sse(torch_mod, refl):
    name = "atrium"
    tv, sc, osc = sse_scene(name)
    cam = util.camera_for(name, tv)
    lights = lights_for(osc, cam, 1)
    want, wst = osc.render_whitted(cam.as_array13(), 1920, 1080, lights, mode=O.MODE_SSE, threads=16, reflections=refl)
    stats = sc.new_stats()
    got = sc.render_whitted(cam, 1920, 1080, lights, stats=stats, reflections=refl).cpu().numpy()
    assert np.array_equal(got, want), int((got != want).sum())
    assert np.array_equal(stats.cpu().numpy().astype(np.uint64), wst), (stats.cpu().numpy(), wst)
    sc.close()


@pytest.mark.parametrize("mode", ["lights", "refl", "depth"])
def test_tile_renderer_in_host_sse(torch_mod, mode):
    """The host-pointer tile API (depth shading fused into the traversal kernel; 4x antialiasing with the depth / colour reduction)."""
    from snail_amd.scene import Scene
    name, resx, resy = "atrium:0.05", 250, 130
    tv, sc, osc = sse_scene(name)
    cam = util.camera_for(name, tv)
    lights = lights_for(osc, cam, 1)
    for aa in (False, True):
        flags = (Scene.RENDER_AA4 if aa else 0) | (Scene.RENDER_REFLECTIONS if mode == "refl" else 0) | (Scene.RENDER_DEPTH if mode == "depth" else 0)
        want, wst = osc.render_whitted(cam.as_array13(), resx, resy, lights, mode=O.MODE_SSE, reflections=mode == "refl", antialias=aa, depth=mode == "depth")
        img, st = sc.render_image_host(cam, resx, resy, lights, flags)
        assert np.array_equal(img, want), (aa, int((img != want).sum()))
        assert np.array_equal(st, wst), (aa, st, wst)
    # the tile list dealt over two handles of the scene: both in the host's arithmetic -> the same bytes; mixed arithmetics are refused
    from snail_amd import render as R
    tiles = R.divide_image(resx, resy)
    sc2 = Scene(sc.bvh, 0)
    with pytest.raises(Exception, match="another arithmetic"):
        sc.render_tiles_host(cam, resx, resy, tiles, lights, 0, scenes=[sc, sc2])
    sc2.set_arith("host_sse")
    want, wst = osc.render_whitted(cam.as_array13(), resx, resy, lights, mode=O.MODE_SSE, reflections=mode == "refl", depth=mode == "depth")
    fl = (Scene.RENDER_REFLECTIONS if mode == "refl" else 0) | (Scene.RENDER_DEPTH if mode == "depth" else 0)
    data, offsets, tst = sc.render_tiles_host(cam, resx, resy, tiles, lights, fl, scenes=[sc, sc2])
    for k, wp in enumerate(O.planar_encode(want, tiles)):
        assert np.array_equal(data[offsets[k]:offsets[k] + len(wp)], wp), ("tile", k)
    assert np.array_equal(tst, wst), (tst, wst)
    sc2.close()
    # the stand-alone depth shading of existing hit records, special values included
    t = torch_mod.tensor([[float("inf"), 1e-6, float("nan"), 1.0, 3.0e38, 1.0e-39] + [2.0] * 250], dtype=torch_mod.float32, device="cuda")
    b = sc.shade_depth(t, arith="host_sse").cpu().numpy().reshape(-1, 3)
    assert np.array_equal(b, O.shade_depth(t.cpu().numpy(), mode=O.MODE_SSE))
    sc.close()


def test_transparency_stage_in_host_sse(torch_mod):
    name, resx, resy = "atrium:0.05", 320, 192
    tv, sc, osc = sse_scene(name)
    cam = util.camera_for(name, tv)
    xy, tp, ip, sel, lights = util.transparency_case(osc, cam, resx, resy, 5, mode=O.MODE_SSE)
    want, wst = osc.trace_transparency(cam.as_array13(), resx, resy, xy, tp, sel, lights[:1], mode=O.MODE_SSE)
    dev = lambda a: torch_mod.from_numpy(np.ascontiguousarray(a)).cuda()
    stats = sc.new_stats()
    got = sc.trace_transparency(cam, resx, resy, dev(xy), dev(tp), dev(ip), dev(sel), lights[:1], stats=stats)
    torch_mod.cuda.synchronize()
    util.assert_bit_equal(got.cpu().numpy(), want, "transColor")
    assert np.array_equal(stats.cpu().numpy().astype(np.uint64), wst), (stats.cpu().numpy(), wst)
    sc.close()


def test_batched_ordered_launches_in_host_sse(torch_mod):
    """The bench's launch form (several frames per launch, a fed-back dispatch order) in the second arithmetic."""
    name, resx, resy = "atrium:0.05", 328, 200
    tv, sc, osc = sse_scene(name)
    cam = util.camera_for(name, tv)
    n = sc.primary_slots(resx, resy)
    cost = torch_mod.zeros(n, dtype=torch_mod.int32, device="cuda")
    sc.trace_primary(cam, resx, resy, slot_cost=cost)
    order = sc.order_from_cost(cost)
    outs = [sc.alloc_frame(resx, resy) for _ in range(3)]
    sc.trace_primary_batch([cam] * 3, resx, resy, outs, order=order)
    torch_mod.cuda.synchronize()
    ref = osc.render_primary(cam.as_array13(), resx, resy, mode=O.MODE_SSE)
    for f in outs:
        compare_frames(f, ref, "batched host_sse")
    sc.close()


def test_tables_of_another_cpu_reproduce_that_cpus_results(torch_mod):
    """snail_arith_set_tables: SNAIL_ARITH_HOST_SSE with the tables of ANOTHER CPU (tests/golden/rcp_tables.npz: the build container's Xeon, on
    which the survey ran the reference, and the pool's EPYC 9575F).  With the Xeon's tables, whatever this box's CPU is:
      * box 256x256 from the survey's camera gives the numbers the survey RECORDED FROM THE REFERENCE ITSELF (SURVEY.md section 8c,
        tests/golden/survey_digests.json): 45 369 hits, sum(triId) 204 078, sum(t) 114 934.835 -- a pin of the HIP path that does not pass
        through the oracle;
      * the bench's atrium frame hashes to the digest committed from the build container (the oracle in ORC_MODE_SSE on that Xeon).
    Back on this host's own tables the frame hashes to this CPU's committed digest (if the file knows the CPU)."""
    import hashlib
    import json
    from snail_amd import HostBVH, scenes, survey_camera
    from snail_amd.scene import Scene, host_sse_tables, set_arith_tables
    from tests.golden import full_size as FS
    gold_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    tabs = np.load(os.path.join(gold_dir, "rcp_tables.npz"))
    own = host_sse_tables()
    own_key = hashlib.sha256(own.tobytes()).hexdigest()[:16]
    sha = lambda t: hashlib.sha256(np.ascontiguousarray(t.cpu().numpy()).tobytes()).hexdigest()
    full = json.load(open(os.path.join(gold_dir, "oracle_full_size.json")))["host_sse"]
    try:
        set_arith_tables(tabs["xeon_skylake_sp"])
        assert np.array_equal(host_sse_tables(), tabs["xeon_skylake_sp"])
        xeon_key = hashlib.sha256(np.ascontiguousarray(tabs["xeon_skylake_sp"]).tobytes()).hexdigest()[:16]
        # (1) the survey's box digest, recorded from the reference on that CPU
        d = json.load(open(os.path.join(gold_dir, "survey_digests.json")))["box"]
        tv = scenes.box_scene()
        sc = Scene(HostBVH.build(tv), 0)
        sc.set_arith("host_sse")
        fr = sc.trace_primary(survey_camera(tv), 256, 256)
        torch_mod.cuda.synchronize()
        t, tid = fr.t.cpu().numpy(), fr.tri_id.cpu().numpy()
        hit = np.isfinite(t)
        assert int(hit.sum()) == d["hits"] and int(tid[hit].astype(np.int64).sum()) == d["sum_id"], (int(hit.sum()), int(tid[hit].astype(np.int64).sum()))
        assert abs(float(t[hit].astype(np.float64).sum()) - d["sum_t"]) < 0.0006
        sc.close()
        # (2) the bench's frame: the digest the build container committed for ITS CPU, reproduced here
        tv, hb, _ = util.scene_pair("atrium")
        sc = Scene(hb, 0)
        sc.set_arith("host_sse")
        cam = FS.bench_camera("atrium")
        want = full[xeon_key]["atrium_1920x1080_c1"]
        fr = sc.trace_primary(cam, 1920, 1080)
        assert (sha(fr.t), sha(fr.u), sha(fr.v), sha(fr.tri_id)) == (want["sha_t"], want["sha_u"], want["sha_v"], want["sha_id"])
        # back to this host's tables: in force from the next set_arith on
        set_arith_tables(None)
        assert np.array_equal(host_sse_tables(), own)
        sc.set_arith("host_sse")
        fr = sc.trace_primary(cam, 1920, 1080)
        if own_key in full:
            want = full[own_key]["atrium_1920x1080_c1"]
            assert (sha(fr.t), sha(fr.u), sha(fr.v), sha(fr.tri_id)) == (want["sha_t"], want["sha_u"], want["sha_v"], want["sha_id"])
        if own_key != xeon_key:
            assert sha(fr.t) != full[xeon_key]["atrium_1920x1080_c1"]["sha_t"]        # two CPUs, two sets of last bits
        sc.close()
        with pytest.raises(Exception, match="reciprocals|must lie"):
            set_arith_tables(np.zeros((3, 4096), dtype=np.uint32))
    finally:
        set_arith_tables(None)


def test_gpu_built_tree_in_host_sse(torch_mod):
    """The arithmetic is a property of the scene handle, whatever built its tree: a linear BVH built on the device (snail_scene_create_lbvh), walked
    in the host's SSE arithmetic, against the oracle's walk of the SAME tree in ORC_MODE_SSE."""
    from snail_amd.scene import Scene
    name = "atrium:0.05"
    tv, _, _ = util.scene_pair(name)
    cam = util.camera_for(name, tv)
    sc = Scene.from_lbvh(tv, 0, max_leaf_tris=4)
    sc.set_arith("host_sse")
    osc2 = O.OracleScene.__new__(O.OracleScene)
    osc2.tris = np.ascontiguousarray(sc.bvh.tris.view(O.TRI_DTYPE)); osc2.nodes = np.ascontiguousarray(sc.bvh.nodes.view(O.NODE_DTYPE))
    osc2.depth = sc.bvh.depth; osc2.perm = sc.perm
    resx, resy = 320, 192
    want = osc2.render_primary(cam.as_array13(), resx, resy, mode=O.MODE_SSE)
    stats = sc.new_stats()
    fr = sc.trace_primary(cam, resx, resy, stats=stats)
    torch_mod.cuda.synchronize()
    compare_frames(fr, want, "lbvh host_sse")
    assert np.array_equal(stats.cpu().numpy().astype(np.uint64), want[4])
    lights = lights_for(osc2, cam, 1)
    wimg, wst = osc2.render_whitted(cam.as_array13(), resx, resy, lights, mode=O.MODE_SSE, reflections=True)
    st = sc.new_stats()
    img = sc.render_whitted(cam, resx, resy, lights, stats=st, reflections=True).cpu().numpy()
    assert np.array_equal(img, wimg) and np.array_equal(st.cpu().numpy().astype(np.uint64), wst)
    sc.close()


def test_bench_checks_itself_on_a_cpu_without_committed_digests(torch_mod, tmp_path):
    """bench.py's `verified` on a host whose rcpps / rsqrtps tables the digest file does not know (simulated: the Xeon's tables with one entry replaced by
    its neighbour's -- tables of no CPU): the timed host_sse frame is checked LIVE against the oracle computing over the same tables (ORC_MODE_TABLE),
    no committed digest and no detour through another arithmetic involved; with the host's own tables the live check runs this CPU's instructions
    (ORC_MODE_SSE) and the committed digest is the second check."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tabs = np.load(os.path.join(root, "tests", "golden", "rcp_tables.npz"))["xeon_skylake_sp"].copy()
    tabs[0, 1001] = tabs[0, 1002]          # (the Xeon's entries come in equal pairs: 1000 = 1001 > 1002) still non-increasing, in (0.5, 1]
    assert tabs[0, 1001] != tabs[0, 1000]
    f = str(tmp_path / "nobody.npy")
    np.save(f, tabs)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    base = [sys.executable, os.path.join(root, "bench.py"), "--steps", "8", "--warmup", "2", "--settle-ms", "0", "--no-cpu-baseline", "--lone-frames", "0", "--arith", "host_sse"]
    r = subprocess.run(base + ["--arith-tables", f], capture_output=True, text=True, timeout=300, cwd=root, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    v = d["verification"]
    assert d["config"]["arith"] == "host_sse" and d["verified"] is True, v
    assert v["live_oracle"] is True and "ORC_MODE_TABLE" in v["live_oracle_how"] and v.get("committed") is None and "no committed digest" in v["note"] and v["buffers_identical"], v
    # config 3 with the bounce on this host's own tables: live against this CPU's instructions
    r = subprocess.run(base + ["--config", "3", "--reflections"], capture_output=True, text=True, timeout=300, cwd=root, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    v = d["verification"]
    assert d["verified"] is True and v["live_oracle"] is True and "ORC_MODE_SSE" in v["live_oracle_how"] and v.get("committed") in (True, None), v


def test_two_scenes_on_one_device_compute_with_two_cpus_tables(torch_mod):
    """The arithmetic is a property of the SCENE HANDLE: its rcpps / rsqrtps tables live in a buffer of its own that every launch is handed
    (dev_sse::hostTab()), filled by snail_scene_set_arith from the tables in force at that moment.  Two scenes on ONE device -- the Xeon's tables on one,
    the EPYC's on the other (tests/golden/rcp_tables.npz) -- with frames interleaved on two streams: each equals the oracle computing over ITS tables
    (ORC_MODE_TABLE; primary hit records, the config-3 frame with the bounce) and, at the bench's size, the digest committed for ITS CPU; a later
    snail_arith_set_tables changes neither until that scene's next set_arith."""
    import hashlib
    import json
    from snail_amd.scene import Scene, host_sse_tables, set_arith_tables
    from tests.golden import full_size as FS
    gold_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    tabs = np.load(os.path.join(gold_dir, "rcp_tables.npz"))
    names = ["xeon_skylake_sp", "epyc_9575f"]
    key = {n: hashlib.sha256(np.ascontiguousarray(tabs[n]).tobytes()).hexdigest()[:16] for n in names}
    assert key[names[0]] != key[names[1]]
    full = json.load(open(os.path.join(gold_dir, "oracle_full_size.json")))["host_sse"]
    sha = lambda t: hashlib.sha256(np.ascontiguousarray(t.cpu().numpy()).tobytes()).hexdigest()
    small, resx, resy = "atrium:0.05", 640, 368
    tv, hb, osc = util.scene_pair(small)
    cam = util.camera_for(small, tv)
    lights = lights_for(osc, cam, 1)
    tvb, hbb, _ = util.scene_pair("atrium")
    camb = FS.bench_camera("atrium")
    try:
        sc, big = {}, {}
        for n in names:                              # set the tables, then the scenes' arithmetic: each scene keeps what was in force at ITS call
            set_arith_tables(tabs[n])
            sc[n], big[n] = Scene(hb, 0), Scene(hbb, 0)
            sc[n].set_arith("host_sse"); big[n].set_arith("host_sse")
        set_arith_tables(None)                       # ... and whatever is set afterwards does not reach them
        st = {n: torch_mod.cuda.Stream() for n in names}
        frames = {n: [] for n in names}; lit = {n: [] for n in names}; bigf = {n: [] for n in names}
        for i in range(4):                           # interleaved, two streams, no synchronisation in between
            for n in names:
                frames[n].append(sc[n].trace_primary(cam, resx, resy, stream=st[n]))
                lit[n].append(sc[n].render_whitted(cam, resx, resy, lights, stream=st[n], reflections=True))
                bigf[n].append(big[n].trace_primary(camb, 1920, 1080, stream=st[n]))
        torch_mod.cuda.synchronize()
        for n in names:
            O.set_tables(tabs[n])
            ref = osc.render_primary(cam.as_array13(), resx, resy, mode=O.MODE_TABLE)
            want, _ = osc.render_whitted(cam.as_array13(), resx, resy, lights, mode=O.MODE_TABLE, reflections=True)
            for f, img in zip(frames[n], lit[n]):
                compare_frames(f, ref, "scene with %s's tables" % n)
                assert np.array_equal(img.cpu().numpy(), want), n
            w = full[key[n]]["atrium_1920x1080_c1"]
            for f in bigf[n]:
                assert (sha(f.t), sha(f.u), sha(f.v), sha(f.tri_id)) == (w["sha_t"], w["sha_u"], w["sha_v"], w["sha_id"]), n
        # the two CPUs' last bits differ, so the two scenes' frames do
        assert not torch_mod.equal(frames[names[0]][0].t, frames[names[1]][0].t)
        for n in names:
            sc[n].close(); big[n].close()
    finally:
        set_arith_tables(None)
        O.set_tables(O.tables_of_this_cpu())


@pytest.mark.parametrize("host_sse", [False, True])
def test_lights_whose_lookups_meet_special_inputs(torch_mod, host_sse):
    """The light kernel's main pass takes the table look-ups in their short form after ONE test of all their inputs per packet and hands a packet with
    a special input (0, denormal, inf, NaN, a flushed result) to the deferred pass, which applies the full rule -- here with lights that produce such
    inputs: a radius so small that 16 (d / r)^2 overflows (FastInv(inf) = 0), one so large that it underflows to a denormal / zero (FastInv -> inf: the
    attenuation saturates the pixel), inside the packet's hit bounds so that the packet-level cull keeps them.  One light (the fused epilogue) and three
    (k_final's), with and without the bounce, equal the oracle byte for byte in either arithmetic."""
    from snail_amd.scene import Scene
    MODE = O.MODE_SSE if host_sse else O.MODE_IEEE
    name, resx, resy = "atrium:0.05", 320, 192
    tv, hb, osc = util.scene_pair(name)
    cam = util.camera_for(name, tv)
    sc = Scene(hb, 0)
    if host_sse:
        sc.set_arith("host_sse")
    bmin, bmax = osc.nodes[0]["bmin"], osc.nodes[0]["bmax"]
    c, e = (bmin + bmax) * 0.5, (bmax - bmin)
    pos = [c[0], c[1] + 0.2 * e[1], c[2]]
    for radius in (1e-30, 1e-18, 1e18, 3e38):
        one = np.array([pos + [1.0, 0.9, 0.8, radius]], dtype=np.float32)
        three = np.array([pos + [1.0, 0.9, 0.8, radius], [c[0] - 0.3 * e[0], c[1] + 0.1 * e[1], c[2] + 0.2 * e[2], 0.3, 0.5, 1.0, 0.6 * float(e.max())],
                          [c[0] + 0.2 * e[0], c[1] + 0.3 * e[1], c[2], 0.5, 0.5, 0.5, radius]], dtype=np.float32)
        for lights in (one, three):
            for refl in (False, True):
                want, wst = osc.render_whitted(cam.as_array13(), resx, resy, lights, mode=MODE, reflections=refl)
                stats = sc.new_stats()
                got = sc.render_whitted(cam, resx, resy, lights, stats=stats, reflections=refl).cpu().numpy()
                assert np.array_equal(got, want), (radius, len(lights), refl, int((got != want).sum()))
                assert np.array_equal(stats.cpu().numpy().astype(np.uint64), wst), (radius, len(lights), refl)
    sc.close()
