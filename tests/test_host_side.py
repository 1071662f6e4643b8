"""CPU tests of the product's host side (no GPU, no compute through the HIP kernels): the C-ABI library
loads and exports every symbol include/snail_hip.h declares; the host SAH builder inside libsnailhip.so
is bit-identical to the oracle's; scene ingest mirrors the loader's order/winding rules; tile plan."""
import os
import re

import numpy as np
import pytest

import snail_amd
from snail_amd import HostBVH, scenes
from snail_amd import render as R
from tests import oracle_lib as O
from tests import util

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_capi_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "snail_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(snail_[a-z_0-9]+)\s*\(", hdr)))
    assert len(declared) >= 15
    lib = snail_amd.lib()
    for name in declared:
        assert hasattr(lib, name), "libsnailhip.so does not export " + name
    from snail_amd._lib import SIGNATURES, DEBUG_SIGNATURES
    assert sorted(SIGNATURES) == declared          # the Python binding covers the whole header
    # the contract carries no workbench: nothing named snail_debug_* in the product header, the product library or its binding
    assert not [n for n in declared if "debug" in n]
    assert not hasattr(lib, "snail_debug_recip_check") and not hasattr(lib, "snail_debug_delay_dev")
    # ... and the workbench build exports the product C-ABI plus everything include/snail_hip_debug.h declares
    dhdr = open(os.path.join(ROOT, "include", "snail_hip_debug.h")).read()
    ddecl = sorted(set(re.findall(r"\b(snail_debug_[a-z_0-9]+)\s*\(", dhdr)))
    assert sorted(DEBUG_SIGNATURES) == ddecl and len(ddecl) >= 6
    from snail_amd._lib import debug_lib
    dl = debug_lib()
    for name in declared + ddecl:
        assert hasattr(dl, name), "libsnailhip_debug.so does not export " + name


def test_missing_extension_fails_loudly(monkeypatch, tmp_path):
    from snail_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(snail_amd.SnailError):
        _lib.lib()


def test_scene_needs_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from snail_amd.scene import Scene
    tv, hb, osc = util.scene_pair("box")
    with pytest.raises(snail_amd.SnailError):
        Scene(hb, 0)


@pytest.mark.parametrize("name", ["box", "atrium:0.05", "stress:0.01", "chain"])
def test_builder_bit_identical_to_oracle(name):
    tv, hb, osc = util.scene_pair(name)
    assert hb.tris.tobytes() == osc.tris.tobytes()
    assert hb.nodes.tobytes() == osc.nodes.tobytes()
    assert hb.depth == osc.depth and np.array_equal(hb.perm, osc.perm)
    if name == "chain":
        leaf = (hb.nodes["sub"] & 0x80000000) != 0
        assert hb.depth == 63 and hb.nodes["aux"][leaf].max() > 64        # forced leaf at maxDepth-1 with a long triangle list


@pytest.mark.parametrize("name", ["box", "atrium:0.05"])
def test_tree_invariants(name):
    tv, hb, osc = util.scene_pair(name)
    n = len(tv)
    assert sorted(hb.perm.tolist()) == list(range(n))                       # a permutation of the input
    # triangles were only permuted, never altered
    fresh = HostBVH.triangles(tv)
    assert hb.tris.tobytes() == fresh[hb.perm].tobytes()
    nodes = hb.nodes
    leaf = (nodes["sub"] & 0x80000000) != 0
    first = (nodes["sub"] & 0x7fffffff)[leaf]
    count = nodes["aux"][leaf]
    order = np.argsort(first)
    assert first[order][0] == 0 and (first[order][1:] == (first[order] + count[order])[:-1]).all()   # leaves tile [0,n)
    assert int(count.sum()) == n and count.max() <= max(4, count.max())
    inner = ~leaf
    child = nodes["sub"][inner]
    assert (child >= 1).all() and (child + 1 < len(nodes)).all()
    assert ((nodes["aux"][inner] & 0xffff) <= 2).all() and ((nodes["aux"][inner] >> 16) <= 1).all()
    # every child box lies inside... not required by the builder (leaf boxes are recomputed), but leaves must bound their triangles
    for i in np.nonzero(leaf)[0][:200]:
        f, c = int(nodes["sub"][i] & 0x7fffffff), int(nodes["aux"][i])
        t = hb.tris[f:f + c]
        p = np.stack([t["a"], t["a"] + t["ba"], t["a"] + t["ca"]], axis=1).reshape(-1, 3)
        assert (p.min(0) >= nodes["bmin"][i] - 1e-6).all() and (p.max(0) <= nodes["bmax"][i] + 1e-6).all()
    assert hb.depth <= 64


def test_triangle_record_matches_reference_formulae():
    tv = scenes.box_scene()
    t = HostBVH.triangles(tv)
    for k in range(len(tv)):
        v0, v1, v2 = tv[k].astype(np.float32)
        assert np.array_equal(t["a"][k], v0) and np.array_equal(t["ba"][k], v1 - v0) and np.array_equal(t["ca"][k], v2 - v0)
        n = np.cross((v1 - v0).astype(np.float64), (v2 - v0).astype(np.float64))
        assert abs(t["t0"][k] - np.linalg.norm(n)) < 1e-5 and abs(t["it0"][k] * t["t0"][k] - 1) < 1e-6
        assert np.allclose(t["plane"][k][:3], n / np.linalg.norm(n), atol=1e-6)
        assert abs(t["plane"][k][3] - (n / np.linalg.norm(n)) @ v0) < 1e-5


def test_obj_ingest_rules(tmp_path):
    p = tmp_path / "m.obj"
    p.write_text("""# quad, degenerate face, negative indices, v/vt/vn forms
v 0 0 0
v 1 0 0
v 1 1 0
v 0 1 0
v 0 0 1
vt 0 0
vn 0 0 1
f 1 2 3 4
f 1/1/1 2/1/1 2/1/1
f 1//1 3//1 5//1
f -5 -4 -1
""")
    tv = scenes.load_obj(str(p), flip=False)
    V = np.array([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0], [0, 0, 1]], dtype=np.float32)
    # quad -> (v0,v1,v2),(v2,v3,v0); the degenerate face (index 2) is replaced by the LAST face (swap-with-last)
    want = np.array([V[[0, 1, 2]], V[[2, 3, 0]], V[[0, 1, 4]], V[[0, 2, 4]]])
    assert np.array_equal(tv, want)
    flipped = scenes.load_obj(str(p), flip=True)
    assert np.array_equal(flipped, want[:, [1, 0, 2]])                      # FlipNormals swaps v0 <-> v1
    # '%f' round trip of a dyadic-grid scene is exact
    a = scenes.drop_degenerate(scenes.atrium(detail=0.01))
    q = tmp_path / "a.obj"
    scenes.save_obj(str(q), a)
    assert np.array_equal(scenes.load_obj(str(q), flip=False), a)
    # sscanf("%f") rounds the text to float32 once: a decimal just above a float32 midpoint, which a double parse rounds onto
    # the midpoint and then (ties-to-even) down
    lo = np.float32(1.0); hi = np.nextafter(lo, np.float32(2.0))
    txt = "1.00000005960464481"          # midpoint of (lo, hi) = 1.000000059604644775390625, plus 3.5e-17
    assert float(np.float32(float(txt))) == float(lo)            # the double-rounding answer
    r = tmp_path / "r.obj"
    r.write_text("v %s 0 0\nv 0 1 0\nv 0 0 1\nf 1 2 3\n" % txt)
    assert scenes.load_obj(str(r), flip=False)[0, 0, 0] == hi    # the reference's answer
    # -swapYZ (src/base_scene.cpp:337-343) after the flip (src/rtracer.cpp:555-557)
    sw = scenes.load_obj(str(p), flip=True, swap_yz=True)
    assert np.array_equal(sw, want[:, [1, 0, 2]][:, :, [0, 2, 1]])


def test_box_restatement_matches_reference_asset(reference_scenes):
    assert np.array_equal(scenes.load_obj(os.path.join(reference_scenes, "box.obj")), scenes.box_scene())


def test_tile_plan_partitions_the_frame():
    for resx, resy, n in ((1920, 1080, 1), (2720, 1528, 2), (3840, 2160, 8), (250, 130, 3)):
        plan = R.ShardPlan.make(resx, resy, n)
        assert len(plan.tiles) == ((resx + 15) // 16) * ((resy + 63) // 64)
        allp = np.concatenate(plan.packets, axis=0)
        want = {(x, y) for y in range(0, resy, 16) for x in range(0, resx, 16)}
        got = [tuple(p) for p in allp.tolist()]
        assert len(got) == len(set(got)) == len(want) and set(got) == want          # every packet exactly once
        sizes = [len(p) for p in plan.packets]
        assert max(sizes) - min(sizes) <= 4 * 2                                     # balanced to within two tiles
        for r in range(n):
            pp = plan.padded_packets(r)
            assert len(pp) == plan.padded and np.array_equal(pp[:len(plan.packets[r])], plan.packets[r])
        assert plan.total_rays() == len(want) * 256
    # identical on every rank without communication
    a, b = R.ShardPlan.make(640, 384, 4), R.ShardPlan.make(640, 384, 4)
    assert np.array_equal(a.owner, b.owner)
    # shuffled round-robin: each group of n consecutive tiles is a permutation of the ranks (src/server.cpp:239-248)
    own = R.assign_tiles(64, 8)
    assert all(sorted(own[i:i + 8].tolist()) == list(range(8)) for i in range(0, 64, 8))


def test_tile_plan_rank0_share_and_pad_entries():
    """rank0_share (the reference's server renders nothing, src/server.cpp:233-265): every packet still exactly once, rank 0 gets about
    that fraction of a fair share, the others stay balanced; pad entries of a short rank lie outside the image (y = PAD_Y), so that
    nothing is traced or counted twice and the scatter skips them."""
    for share, n in ((0.0, 8), (0.25, 8), (0.5, 4), (1.0, 4), (0.0, 2)):
        plan = R.ShardPlan.make(1920, 1080, n, rank0_share=share)
        got = [tuple(p) for p in np.concatenate([p.reshape(-1, 2) for p in plan.packets], axis=0).tolist()]
        assert len(got) == len(set(got)) == 120 * 68
        sizes = [len(p) for p in plan.packets]
        fair = 8160 / (n - 1 + share)
        assert abs(sizes[0] - share * fair) <= 8 + 0.05 * fair, (share, n, sizes)
        assert max(sizes[1:]) - min(sizes[1:]) <= 8
        assert plan.padded == max(sizes)
        for r in range(n):
            pp = plan.padded_packets(r)
            assert pp.shape == (plan.padded, 2) and np.array_equal(pp[:sizes[r]], plan.packets[r].reshape(-1, 2))
            assert (pp[sizes[r]:, 1] == R.PAD_Y).all() and R.PAD_Y > 1 << 20
    assert np.array_equal(R.assign_tiles(64, 8, rank0_share=1.0), R.assign_tiles(64, 8))      # the default is the reference's round-robin


def test_scene_create_rejects_runaway_trees():
    """snail_scene_create's host-side validation needs no GPU: a tree with a back-edge or a shared subtree (every wave would loop for
    ever) or deeper than declared (the VGPR-lane stack would wrap) is refused before anything touches a device."""
    from snail_amd import _lib
    from tests import util
    tv, hb, _ = util.scene_pair("atrium:0.02")
    L = _lib.lib()
    inner = [i for i in range(hb.n_nodes) if not (int(hb.nodes["sub"][i]) & 0x80000000)]
    def create(nodes, depth):
        h = L.snail_scene_create(_lib.ptr(nodes), len(nodes), _lib.ptr(hb.tris), hb.n_tris, depth, 0)
        msg = L.snail_last_error().decode()
        if h:
            L.snail_scene_destroy(h)
        return h, msg
    cyc = hb.nodes.copy(); cyc["sub"][inner[-1]] = cyc["sub"][0]
    h, msg = create(cyc, hb.depth); assert not h and "reachable twice" in msg, msg
    sh = hb.nodes.copy(); sh["sub"][inner[2]] = sh["sub"][inner[1]]
    h, msg = create(sh, hb.depth); assert not h and "reachable twice" in msg, msg
    h, msg = create(hb.nodes, hb.depth - 1); assert not h and "deeper than the declared depth" in msg, msg
    h, msg = create(hb.nodes, 65); assert not h and "depth" in msg, msg
    oob = hb.nodes.copy(); oob["sub"][inner[0]] = hb.n_nodes
    h, msg = create(oob, hb.depth); assert not h and "child" in msg, msg


def test_order_flags_are_validated_before_anything_is_launched():
    """snail_order_from_cost_hint_dev (include/snail_hip.h: SNAIL_ORDER_AUTO / SNAIL_ORDER_SORTED) refuses unknown order flags and null buffers on the
    host, before any device call; an empty input is a no-op."""
    import ctypes
    from snail_amd import _lib
    L = _lib.lib()
    buf = (ctypes.c_int32 * 8)()
    p = ctypes.cast(buf, ctypes.c_void_p)
    assert L.snail_order_from_cost_hint_dev(p, 0, p, 0, None) == 0
    assert L.snail_order_from_cost_hint_dev(p, 8, p, 2, None) == 1 and "order flags" in L.snail_last_error().decode()
    assert L.snail_order_from_cost_hint_dev(None, 8, p, 1, None) == 1 and "null buffer" in L.snail_last_error().decode()
    assert L.snail_order_from_cost_dev(p, 8, None, None) == 1


def build_adapter_mock(tmp_path):
    import subprocess
    exe = str(tmp_path / "adapter_mock")
    libdir = os.path.join(ROOT, "snail_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-pthread", os.path.join(ROOT, "tests", "cpp", "adapter_mock.cpp"), "-o", exe,
                           "-L" + libdir, "-lsnailhip", "-Wl,-rpath," + libdir])
    return exe


def test_cpp_adapter_compiles_and_links(tmp_path):
    """include/snail_adapter.hpp (AccStruct-shaped HipBVH + TraceFrame) against mock types with the reference's
    member names; links against libsnailhip.so; runs without touching the GPU when given no arguments."""
    import subprocess
    exe = build_adapter_mock(tmp_path)
    out = subprocess.run([exe], capture_output=True, text=True, check=True).stdout
    assert "compiled and linked" in out


def test_tile_layout_offsets():
    """Scene.tile_layout: first packet and byte offset of each tile's planes when tiles are stored back to back (render.tile_packets order)."""
    from snail_amd import render as R
    from snail_amd.scene import Scene
    tiles = R.divide_image(40, 100)                       # 16x64 tiles with partial right / bottom tiles
    first, off, total = Scene.tile_layout(tiles)
    npk = [((w + 15) // 16) * ((h + 15) // 16) for _, _, w, h in tiles.tolist()]
    assert first.tolist() == np.concatenate([[0], np.cumsum(npk)[:-1]]).tolist()
    assert off.tolist() == np.concatenate([[0], np.cumsum([3 * w * h for _, _, w, h in tiles.tolist()])[:-1]]).tolist()
    assert total == 3 * 40 * 100 and len(R.tile_packets(tiles)) == sum(npk)


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus N` WITHOUT torch.distributed.run (the shape of the driver's scaling command) must start its N ranks itself
    -- as a child process, before anything touches a GPU -- and relay rank 0's JSON line and the exit status.  --dry-run stops after the
    rendezvous and one all-reduce (gloo here), so this runs without a GPU."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--dry-run"], capture_output=True, text=True,
                       timeout=300, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout           # ONE JSON line: rank 0's
    d = json.loads(lines[0])
    assert d == {"dry_run": True, "ranks": 2, "n_gpus": 2, "backend": "gloo"}
    # the status of a failing child is relayed too (here: the ranks refuse to run without a GPU)
    import torch
    if not torch.cuda.is_available():
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "1", "--warmup", "0"], capture_output=True,
                           text=True, timeout=300, env=env, cwd=ROOT)
        assert r.returncode != 0 and "needs an MI355X" in (r.stdout + r.stderr)


def test_kernel_resource_budget():
    """The register budget the measurements rest on, from the compiler's own report (`make -C snail_amd/csrc asm`,
    -Rpass-analysis=kernel-resource-usage): the hand-written node loop pins s[68:91] through clobber lists and the primary kernel sits at
    its occupancy step -- a compiler or source change that spills, or that costs a wave, must fail HERE and not show up as a slower bench."""
    import subprocess
    csrc = os.path.join(ROOT, "snail_amd", "csrc")
    r = subprocess.run(["make", "-C", csrc, "asm"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    rep = open(os.path.join(csrc, "snail_hip.resources")).read()
    kernels = {}
    cur = None
    for line in rep.splitlines():
        m = re.search(r"remark:\s+Function Name: (\S+)", line)
        if m:
            cur = kernels.setdefault(m.group(1), {})
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\S+) \[-Rpass", line)
        if m and cur is not None:
            cur[m.group(1).strip()] = m.group(2)
    # the kernels exist once per arithmetic (namespaces dev and dev_sse, snail_dev.inc): the default arithmetic's kernels are held to the
    # budget the measurements rest on; the host-SSE kernels (a table look-up + a Newton step where the others have v_rcp_f32) to their own
    for ns, budget in (("_ZN3dev", json_budget()), ("_ZN7dev_sse", json_budget_host_sse())):
      for name, want in budget.items():
        hits = [k for k in kernels if name in k and k.startswith(ns)]
        assert hits, "no kernel matching %s%s in the resource report" % (ns, name)
        for k in hits:
            got = kernels[k]
            assert int(got["VGPRs"]) <= want["vgprs"], (k, got)
            assert int(got["Occupancy"]) >= want["waves"], (k, got)
            assert int(got["VGPRs Spill"]) <= want.get("vgpr_spill", 0) and int(got["SGPRs Spill"]) <= want.get("sgpr_spill", 0), (k, got)
            # (the asm statements clobber s68..s91 by name; a build in which the allocator ran out of room shows up as SGPR spills above)


def json_budget_host_sse():
    """the table arithmetic's kernels: the IEEE budget (round 5: the light kernel's look-ups take their short form after one test per packet, and a
    culled packet shares the traced packet's epilogue -- 8 spilled registers where round 4 had 31; the mirrored-ray generator of k_final 51 VGPRs where it had 81)"""
    return {
        "k_primaryILb0E": {"vgprs": 80, "waves": 6},
        "k_lightILb0ELi0E": {"vgprs": 80, "waves": 6, "vgpr_spill": 10, "sgpr_spill": 10},
        "k_lightILb0ELi1E": {"vgprs": 80, "waves": 6, "vgpr_spill": 4, "sgpr_spill": 4},
        "k_finalILi0ELi1E": {"vgprs": 80, "waves": 6},
        "k_raysILb0ELb1ELb0ELb0E": {"vgprs": 96, "waves": 5},
        "k_raysILb0E": {"vgprs": 112, "waves": 4, "sgpr_spill": 8},   # (scalar spills go to VGPR lanes, not to memory)
    }


def json_budget():
    """mangled-name fragment -> {vgprs (max), waves per SIMD (min), vgpr_spill (max)}; DESIGN.md section 3 quotes these"""
    return {
        "k_primaryILb0E": {"vgprs": 80, "waves": 6},
        "k_shadowILb0E": {"vgprs": 88, "waves": 5},                # the compiler's own allocation: nothing spilled
        # six-wave budget; round 4: the kernel also finishes its packet when the frame has ONE light (the tail of TraceLight, the colour and the
        # store, k_final's work): what is spilled is the sample set-up before the walk and that epilogue after it, one register inside the walk as before
        "k_lightILb0ELi0E": {"vgprs": 80, "waves": 6, "vgpr_spill": 10, "sgpr_spill": 10},
        "k_lightILb0ELi1E": {"vgprs": 80, "waves": 6, "vgpr_spill": 4, "sgpr_spill": 4},
        "k_finalILi0ELi1E": {"vgprs": 80, "waves": 6},
        "k_raysILb0ELb1ELb0ELb0E": {"vgprs": 96, "waves": 5},      # mirrored packets (per-ray origins, masks): the compiler's own allocation, nothing spilled
        "k_raysILb0E": {"vgprs": 112, "waves": 4},                 # every generic-packet kernel of ordinary trees: no scratch
    }


def test_counters_carry_the_kernel_hash():
    """profiles/traffic.json stores the hash of the kernel sources its PMC counters were measured on; bench.py compares it with the
    sources it runs and says `counters_stale` when they differ (the roofline must not price a new kernel with an old instruction count)."""
    import importlib.util
    import json
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    d = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    assert re.fullmatch(r"[0-9a-f]{16}", d.get("_kernel_sha16", "")), "tools/make_traffic.py stores _kernel_sha16"
    tr = b.pmc_counters("atrium_1920x1080_n1_c1")
    assert tr is not None and tr["_stale"] == (d["_kernel_sha16"] != b.kernel_source_sha16())


def test_host_sse_tables_reproduce_this_cpus_rcpps_and_rsqrtps():
    """SNAIL_ARITH_HOST_SSE rests on one claim: this CPU's rcpps / rsqrtps are functions of the sign, the exponent and the top 12 mantissa
    bits (snail_amd/csrc/host_sse.h).  The library proves the block structure when it takes the tables; here the table RULE -- the same inline
    functions the device runs -- is compared with the instructions over ALL 2^32 float bit patterns, for both instructions (no GPU needed)."""
    import ctypes as C
    import numpy as np
    from snail_amd._lib import check, lib
    L = lib()
    tab = np.zeros(3 * 4096, dtype=np.uint32)
    check(L.snail_host_sse_tables(tab.ctypes.data_as(C.c_void_p)), "snail_host_sse_tables")
    t = tab.reshape(3, 4096)
    assert ((t >> 23) >= 126).all() and ((t >> 23) <= 127).all()                  # every entry in (0.5, 1]
    assert (np.diff(t.astype(np.int64), axis=1) <= 0).all()                       # monotone: 1 / x and 1 / sqrt(x) fall
    assert (t & 0x7ff).max() == 0                                                 # 12 significant mantissa bits on every CPU seen so far
    threads = min(8, os.cpu_count() or 1)
    for fn in (0, 1):
        bad, first = C.c_uint64(1), C.c_uint32(0)
        check(L.snail_host_sse_check(fn, 0, 1 << 32, threads, C.byref(bad), C.byref(first)), "snail_host_sse_check")
        assert bad.value == 0, (fn, bad.value, hex(first.value))
    # the oracle's SSE mode runs the instructions themselves: Inv / RSqrt of the oracle = table value + Newton, on a sample
    from tests import oracle_lib as O
    rng = np.random.RandomState(5)
    xs = np.concatenate([rng.uniform(1e-3, 1e3, 2000), -rng.uniform(1e-3, 1e3, 500), [1.0, 2.0, 0.5, 3.0e38, 2.0e-38]]).astype(np.float32)
    for x in xs:
        b = int(np.float32(x).view(np.uint32))
        e, m = (b >> 23) & 255, b & 0x7fffff
        base = int(t[0][m >> 11])
        oe = (base >> 23) + 127 - e
        r = np.uint32((b & 0x80000000) | ((oe << 23) | (base & 0x7fffff) if oe > 0 else 0)).view(np.float32)
        want = np.float32(np.float32(r + r) - np.float32(np.float32(np.float32(x) * r) * r))
        got = np.float32(O.lib().orc_inv(C.c_float(float(x)), O.MODE_SSE))
        assert got.view(np.uint32) == want.view(np.uint32), (x, got, want)


def test_arith_tables_of_another_cpu_are_taken_and_given_back():
    """snail_arith_set_tables (no GPU needed for the host side): the committed tables of the two CPUs seen so far are accepted and returned by
    snail_host_sse_tables while in force, garbage is refused, NULL restores this host's own -- and the build container's own tables ARE the
    committed `xeon_skylake_sp` set when it runs on that CPU (the survey's)."""
    import ctypes as C
    import numpy as np
    from snail_amd._lib import check, lib
    from snail_amd.scene import host_sse_tables, set_arith_tables
    tabs = np.load(os.path.join(ROOT, "tests", "golden", "rcp_tables.npz"))
    own = host_sse_tables()
    try:
        for name in ("xeon_skylake_sp", "epyc_9575f"):
            set_arith_tables(tabs[name])
            assert np.array_equal(host_sse_tables(), tabs[name])
            bad = C.c_uint64(0)
            check(lib().snail_host_sse_check(0, 0x3f800000, 1 << 23, 2, C.byref(bad), None), "snail_host_sse_check")
            assert (bad.value == 0) == np.array_equal(tabs[name][0], own[0])     # the check keeps comparing with THIS host's instruction
        for garbage in (np.zeros((3, 4096), dtype=np.uint32), own[:, ::-1].copy()):
            with pytest.raises(Exception, match="must lie"):
                set_arith_tables(garbage)
        assert np.array_equal(host_sse_tables(), tabs["epyc_9575f"])             # a refused call changes nothing
    finally:
        set_arith_tables(None)
    assert np.array_equal(host_sse_tables(), own)
    with open("/proc/cpuinfo") as f:
        model = next((l for l in f if l.startswith("model name")), "")
    if "Xeon(R) Processor @ 2.10GHz" in model:
        assert np.array_equal(own, tabs["xeon_skylake_sp"])


def test_header_is_a_c_header_and_the_host_entry_points_run_from_c(tmp_path):
    """include/snail_hip.h compiles as PLAIN C (gcc -std=c99 -Wall -Werror -pedantic), every function it declares links against libsnailhip.so, and the entry
    points that need no GPU run from a C host: the SAH builder on the reference's cube, the rcpps / rsqrtps table probe, error reporting by status + text
    (tests/c/abi_c.c).  Any FFI a host language of the reference's world would use -- cgo, JNI, ctypes -- binds exactly this."""
    import subprocess
    exe = str(tmp_path / "abi_c")
    libdir = os.path.join(ROOT, "snail_amd")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", os.path.join(ROOT, "tests", "c", "abi_c.c"), "-o", exe, "-L" + libdir, "-lsnailhip",
                           "-Wl,-rpath," + libdir])
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0 and "C ABI ok: 9 nodes, depth 4" in r.stdout, (r.returncode, r.stdout, r.stderr)
    # ... and the list of addresses taken there is the header's list of declarations
    decl = set(re.findall(r"\b(snail_[a-z0-9_]+)\s*\(", open(os.path.join(ROOT, "include", "snail_hip.h")).read())) - {"snail_hip"}
    used = set(re.findall(r"ADDR\((snail_[a-z0-9_]+)\)", open(os.path.join(ROOT, "tests", "c", "abi_c.c")).read()))
    from snail_amd._lib import SIGNATURES
    assert used == set(SIGNATURES), (sorted(set(SIGNATURES) - used), sorted(used - set(SIGNATURES)))
    assert set(SIGNATURES) <= decl
