"""The reference's OWN meshes in front of the HIP kernels (run with -m gpu on an MI355X).

/root/reference does not exist on the GPU box; tests/golden/<name>_tris.npz holds the post-ingest float32 triangle soup of
/root/reference/scenes/{lancia,feline,barracks}.obj and tests/golden/ref_meshes.json the oracle's digests of the survey's frames, both
written in the build container by tests/golden/make_ref_mesh_fixture.py.  Unlike the procedural scenes (vertices on a 1/1024 grid) these are
scanned / modelled meshes with full-mantissa coordinates.  Three kinds of evidence per mesh:
  * the HIP frame against the COMMITTED oracle digests (IEEE arithmetic: CPU independent);
  * the HIP frame against the oracle LIVE on this box, bit for bit, in both arithmetics (host_sse: this CPU's rcpps / rsqrtps on both sides);
  * with the rcpps / rsqrtps tables of the CPU the survey ran the reference on given to the library (tests/golden/rcp_tables.npz), the numbers
    SURVEY.md section 8(c) recorded FROM THE REFERENCE ITSELF -- lancia: 81 372 hits, sum(triId) 532 645 744, sum(t) 1 310 178.215 -- straight
    from the HIP frame: a pin that does not pass through the oracle."""
import hashlib
import json
import os

import numpy as np
import pytest

from snail_amd import HostBVH, survey_camera
from tests import oracle_lib as O
from tests import util

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
MESHES = ["lancia", "feline", "barracks"]


@pytest.fixture(scope="module")
def torch_mod():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch


def sha(a):
    if hasattr(a, "cpu"):
        a = a.cpu().numpy()
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def load(name):
    g = json.load(open(os.path.join(GOLD, "ref_meshes.json")))[name]
    tv = np.load(os.path.join(GOLD, name + "_tris.npz"))["tris"]
    assert hashlib.sha256(np.ascontiguousarray(tv).tobytes()).hexdigest() == g["sha_tris_npz"]
    return g, tv


def bench_light(scene_like):
    """bench.py --config 3's light for the tree's box (an OracleScene or a HostBVH: both carry .nodes)"""
    from tests.golden.full_size import bench_light as BL
    return BL(scene_like.nodes[0]["bmin"], scene_like.nodes[0]["bmax"])


def frame_numbers(fr):
    t, tid = fr.t.cpu().numpy(), fr.tri_id.cpu().numpy()
    hit = np.isfinite(t)
    return int(hit.sum()), int(tid[hit].astype(np.int64).sum()), float(t[hit].astype(np.float64).sum())


def check_against_digest(sc, cam, g, what):
    resx, resy = g["res"] if "res" in g else (1920, 1080)
    stats = sc.new_stats()
    fr = sc.trace_primary(cam, resx, resy, stats=stats)
    assert (sha(fr.t), sha(fr.u), sha(fr.v), sha(fr.tri_id)) == (g["sha_t"], g["sha_u"], g["sha_v"], g["sha_id"]), what
    assert [int(x) for x in stats.cpu().numpy()] == g["stats"], what
    hits, sid, st = frame_numbers(fr)
    assert (hits, sid) == (g["hits"], g["sum_id"]) and abs(st - g["sum_t"]) < 0.0006, what
    return fr


@pytest.mark.parametrize("name", MESHES)
def test_reference_mesh_ieee_against_committed_digests_and_live_oracle(torch_mod, name):
    """SNAIL_ARITH_IEEE: hit records, TreeStats and the config-3 frames (one light; with and without the mirrored bounce) of the survey's
    1920x1080 view equal the digests the build container committed AND the oracle run here, bit for bit; the product's SAH builder yields the
    oracle's tree on this mesh byte for byte."""
    from snail_amd.scene import Scene
    g, tv = load(name)
    hb = HostBVH.build(tv)
    osc = O.OracleScene(tv)
    assert (len(tv), hb.n_nodes, hb.depth) == (g["tris"], g["nodes"], g["depth"])
    assert np.array_equal(hb.nodes.view(np.uint8), osc.nodes.view(np.uint8)) and ("%016x" % osc.fnv_nodes(), "%016x" % osc.fnv_tris()) == (g["fnv_nodes"], g["fnv_tris"])
    cam = survey_camera(tv)
    assert [float(x) for x in cam.as_array13()] == g["camera"]
    sc = Scene(hb, 0)
    resx, resy = g["res"]
    fr = check_against_digest(sc, cam, dict(g["ieee"], res=g["res"]), name + " ieee")
    ref = osc.render_primary(cam.as_array13(), resx, resy, mode=O.MODE_IEEE, threads=16)
    for a, b, n in ((fr.t, ref[0], "t"), (fr.u, ref[1], "u"), (fr.v, ref[2], "v"), (fr.tri_id, ref[3], "triId")):
        util.assert_bit_equal(a.cpu().numpy(), b, "%s live %s" % (name, n))
    lights = bench_light(osc)
    for key, refl in (("c3", False), ("c3r", True)):
        stats = sc.new_stats()
        img = sc.render_whitted(cam, resx, resy, lights, stats=stats, reflections=refl)
        assert sha(img) == g["ieee"][key]["sha_bgr"] and [int(x) for x in stats.cpu().numpy()] == g["ieee"][key]["stats"], (name, key)
    assert sc.account_primary(cam, resx, resy)[3] == g["ieee"]["hits_padded"]
    sc.close()


@pytest.mark.parametrize("name", MESHES)
def test_reference_mesh_host_sse_against_live_oracle(torch_mod, name):
    """SNAIL_ARITH_HOST_SSE with this box's own tables against the oracle's ORC_MODE_SSE, which executes this CPU's rcpps / rsqrtps: hit records,
    TreeStats and the config-3 frame with the bounce, bit for bit, rays generated on the device."""
    from snail_amd.scene import Scene, host_sse_tables
    g, tv = load(name)
    hb = HostBVH.build(tv)
    osc = O.OracleScene(tv)
    cam = survey_camera(tv)
    sc = Scene(hb, 0)
    sc.set_arith("host_sse")
    resx, resy = g["res"]
    stats = sc.new_stats()
    fr = sc.trace_primary(cam, resx, resy, stats=stats)
    ref = osc.render_primary(cam.as_array13(), resx, resy, mode=O.MODE_SSE, threads=16)
    for a, b, n in ((fr.t, ref[0], "t"), (fr.u, ref[1], "u"), (fr.v, ref[2], "v"), (fr.tri_id, ref[3], "triId")):
        util.assert_bit_equal(a.cpu().numpy(), b, "%s host_sse live %s" % (name, n))
    assert [int(x) for x in stats.cpu().numpy()] == [int(x) for x in ref[4]]
    lights = bench_light(osc)
    want, wst = osc.render_whitted(cam.as_array13(), resx, resy, lights, mode=O.MODE_SSE, threads=16, reflections=True)
    stats = sc.new_stats()
    img = sc.render_whitted(cam, resx, resy, lights, stats=stats, reflections=True).cpu().numpy()
    assert np.array_equal(img, want) and [int(x) for x in stats.cpu().numpy()] == [int(x) for x in wst]
    own_key = hashlib.sha256(host_sse_tables().tobytes()).hexdigest()[:16]
    if own_key in g["host_sse"]:          # a CPU the fixture was made on: the committed digests as well
        check_against_digest(sc, cam, dict(g["host_sse"][own_key], res=g["res"]), name + " host_sse committed")
    sc.close()


@pytest.mark.parametrize("name", MESHES)
def test_reference_mesh_reproduces_the_surveys_numbers_with_the_survey_cpus_tables(torch_mod, name):
    """The pin that bypasses the oracle: with the rcpps / rsqrtps tables of the CPU the survey ran the reference on (the build container's Xeon,
    tests/golden/rcp_tables.npz) the HIP path -- on whatever CPU this box has -- produces the numbers SURVEY.md section 8(c) recorded from the
    reference itself: lancia 81 372 hits / sum(triId) 532 645 744 / sum(t) 1 310 178.215 (a 30 K-triangle scanned mesh: the generator's RSqrt and
    SafeInv, the SAH tree's triangle permutation, the walk, Inv(det) and every strict-< tie as the reference computed them), barracks 373 533
    hits, feline 442 567 hits over all traced rays; and the frames hash to the digests the build container committed for that CPU."""
    from snail_amd.scene import Scene, host_sse_tables, set_arith_tables
    g, tv = load(name)
    tabs = np.load(os.path.join(GOLD, "rcp_tables.npz"))
    xeon = np.ascontiguousarray(tabs["xeon_skylake_sp"])
    xeon_key = hashlib.sha256(xeon.tobytes()).hexdigest()[:16]
    assert xeon_key in g["host_sse"], "the fixture was made on the survey's CPU"
    sv = g["survey"]
    try:
        set_arith_tables(xeon)
        assert np.array_equal(host_sse_tables(), xeon)
        sc = Scene(HostBVH.build(tv), 0)
        sc.set_arith("host_sse")
        cam = survey_camera(tv)
        resx, resy = g["res"]
        fr = check_against_digest(sc, cam, dict(g["host_sse"][xeon_key], res=g["res"]), name + " host_sse, survey CPU's tables")
        hits, sid, st = frame_numbers(fr)
        if "hits" in sv:
            assert hits == sv["hits"], (hits, sv)
        if "sum_id" in sv:
            assert sid == sv["sum_id"] and abs(st - sv["sum_t"]) < 0.0006, (sid, st, sv)
        if "nodes" in sv:
            assert (sc.bvh.n_nodes, sc.bvh.depth, len(tv)) == (sv["nodes"], sv["depth"], sv["tris"])
        if "hits_padded" in sv:   # the survey counted over every traced ray, the rows that pad 1080 to whole packets included
            fp = sc.trace_frame_packets_host(cam, resx, resy)
            assert int(np.isfinite(fp[0]).sum()) == sv["hits_padded"], (int(np.isfinite(fp[0]).sum()), sv)
        lights = bench_light(sc.bvh)
        for key, refl in (("c3", False), ("c3r", True)):
            stats = sc.new_stats()
            img = sc.render_whitted(cam, resx, resy, lights, stats=stats, reflections=refl)
            want = g["host_sse"][xeon_key][key]
            assert sha(img) == want["sha_bgr"] and [int(x) for x in stats.cpu().numpy()] == want["stats"], (name, key)
        sc.close()
    finally:
        set_arith_tables(None)
