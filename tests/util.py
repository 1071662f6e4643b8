"""Shared helpers for the parity tests: scene construction (product builder AND oracle builder),
cameras, seeded secondary-ray packets in the reference's Context layout."""
from __future__ import annotations

import functools
import math

import numpy as np

from snail_amd import FPSCamera, HostBVH, scenes, survey_camera
from tests import oracle_lib as O


@functools.lru_cache(maxsize=8)
def scene_pair(name: str):
    """(tri_verts, HostBVH built by the product, OracleScene built by the oracle)."""
    tv = scenes.scene_by_name(name)
    return tv, HostBVH.build(tv), O.OracleScene(tv)


def camera_for(name: str, tv):
    if name.startswith("atrium"):
        pos, ang, pitch = scenes.atrium_camera()
        return FPSCamera(pos, ang, pitch).camera()
    if name.startswith("stress"):
        pos, ang, pitch = scenes.stress_camera()
        return FPSCamera(pos, ang, pitch).camera()
    if name.startswith("patches"):   # head-on, one board cell = 16 x 16 pixels at 96 x 96 (board 6 x 6, plane_dist 1: the board spans +-3 at distance 3 * ...)
        return FPSCamera(np.array([3.0, 3.0, -6.0], dtype=np.float32), 0.0, 0.0).camera()
    if name.startswith("offgrid-in"):   # inside the blob, looking +z through it (the plain name: the survey's far camera)
        return FPSCamera((np.asarray(scenes.OFFGRID_ORIGIN, dtype=np.float32) + np.array([0.0, 0.0, -9.0], dtype=np.float32)), 0.0, 0.0).camera()
    if name.startswith("chain"):
        return FPSCamera(np.array([-0.25, 0.004, 0.002], dtype=np.float32), -math.pi / 2, 0.0).camera()   # looking down +x through the chain
    return survey_camera(tv)


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32 if a.dtype.itemsize == 4 else np.uint8)


def assert_bit_equal(a, b, what=""):
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    ne = bits(a) != bits(b)
    assert not ne.any(), "%s: %d of %d elements differ (first at %s: %r vs %r)" % (
        what, int(ne.sum()), ne.size, np.argwhere(ne)[0], a[tuple(np.argwhere(ne)[0])], b[tuple(np.argwhere(ne)[0])])


def secondary_packets(oscene, cam, resx, resy, n_packets, seed, shared, masked, size=64, poison=False, coherent=False):
    """Build seeded packets the way the reference's secondary rays look (src/scene_trace.cpp:603-634):
    origins on first-hit points of primary packets, directions = mirrored primaries + jitter; idir = 1/(d+1e-8);
    masks with random dead lanes; distance = +inf / -inf(masked), object = 0 (src/scene_trace.cpp:112-115).
    Returns numpy arrays in Context layout."""
    rng = np.random.RandomState(seed)
    nq = n_packets * size
    origin = np.zeros((n_packets if shared else nq, 12), dtype=np.float32)
    dirs = np.zeros((nq, 12), dtype=np.float32)
    bmin, bmax = oscene.nodes[0]["bmin"], oscene.nodes[0]["bmax"]
    centre, ext = (bmin + bmax) * 0.5, (bmax - bmin)
    for p in range(n_packets):
        base = centre + (rng.rand(3) - 0.5) * ext * 0.6
        main = rng.randn(3)
        main /= np.linalg.norm(main)
        for q in range(size):
            # coherent: neighbouring quads look in neighbouring directions from nearly one point, as a mirrored packet's do -- box tests then
            # narrow the quad range, which the scattered form hardly ever does
            d = main[None, :] + (0.002 if coherent else 0.08) * rng.randn(4, 3) + (0.012 if coherent else 0.004) * np.array([[q % 4, q // 4, 0]])
            d /= np.linalg.norm(d, axis=1, keepdims=True)
            dirs[p * size + q] = d.T.reshape(-1).astype(np.float32)
            if not shared:
                o = base[None, :] + (0.0005 if coherent else 0.02) * ext * rng.randn(4, 3)
                origin[p * size + q] = o.T.reshape(-1).astype(np.float32)
        if shared:
            origin[p] = np.repeat(base.astype(np.float32), 4)
    if poison:
        # axis-parallel and the SafeInv singularity dir == -1e-8 (src/rtbase.h:117-120): idir = 1/0 = inf
        dirs[0, 0:4] = 0.0
        dirs[1, 4:8] = np.float32(-0.00000001)
        dirs[2 * size // 3, 8:12] = np.float32(-0.00000001)
    idir = (np.float32(1.0) / (dirs + np.float32(0.00000001))).astype(np.float32)
    mask = None
    dist = np.full((nq, 4), np.inf, dtype=np.float32)
    if masked:
        mask = rng.randint(0, 16, size=nq).astype(np.uint8)
        mask[rng.rand(nq) < 0.2] = 0
        mask[rng.rand(nq) < 0.3] = 15
        if n_packets > 1:
            mask[size:2 * size] = 0          # one fully dead packet
        lanes = (mask[:, None] >> np.arange(4)[None, :]) & 1
        dist[lanes == 0] = -np.inf
    obj = np.zeros((nq, 4), dtype=np.int32)
    bary = np.zeros((nq, 8), dtype=np.float32)
    return origin, dirs, idir, mask, dist, obj, bary


def shadow_packets(oscene, n_packets, seed, size=64):
    """Shadow packets as Scene::TraceLight builds them (src/scene_trace.cpp:538-558): dir = (P - L)/|P - L|,
    idir = SafeInv(dir), distance = |P - L| * 0.9999 for lit candidates, -inf for masked lanes."""
    rng = np.random.RandomState(seed)
    nq = n_packets * size
    bmin, bmax = oscene.nodes[0]["bmin"], oscene.nodes[0]["bmax"]
    centre, ext = (bmin + bmax) * 0.5, (bmax - bmin)
    origin = np.zeros((n_packets, 3), dtype=np.float32)
    dirs = np.zeros((nq, 12), dtype=np.float32)
    dist = np.zeros((nq, 4), dtype=np.float32)
    for p in range(n_packets):
        L = (centre + (rng.rand(3) - 0.5) * ext * np.array([0.5, 0.2, 0.5]) + np.array([0, 0.3 * ext[1], 0])).astype(np.float32)
        origin[p] = L
        tgt = centre + (rng.rand(3) - 0.5) * ext
        for q in range(size):
            P = (tgt[None, :] + 0.03 * ext[None, :] * rng.randn(4, 3) + 0.01 * ext * np.array([[q % 4, 0, q // 4]])).astype(np.float32)
            lv = (P - L[None, :]).astype(np.float32)
            ln = np.sqrt((lv * lv).sum(axis=1)).astype(np.float32)
            d = (lv / ln[:, None]).astype(np.float32)
            dirs[p * size + q] = d.T.reshape(-1)
            dist[p * size + q] = (ln * np.float32(0.9999)).astype(np.float32)
    dead = rng.rand(nq, 4) < 0.25
    dist[dead] = -np.inf
    if n_packets > 2:
        dist[2 * size:3 * size] = -np.inf      # fully masked packet
    idir = (np.float32(1.0) / (dirs + np.float32(0.00000001))).astype(np.float32)
    return origin, dirs, idir, dist


def frame_to_packets(plane, xy):
    """numpy [resy,resx] -> packet-major [n,256] in the reference's quad order (pixels outside the image = 0)."""
    resy, resx = plane.shape
    out = np.zeros((len(xy), 256), dtype=plane.dtype)
    for i, (x, y) in enumerate(np.asarray(xy).tolist()):
        blk = np.zeros((16, 16), dtype=plane.dtype)
        h, w = min(16, resy - y), min(16, resx - x)
        blk[:h, :w] = plane[y:y + h, x:x + w]
        out[i] = blk.reshape(-1)          # row ty, then 4 quads of 4 pixels = row-major 16x16
    return out


def transparency_case(osc, cam, resx, resy, seed, mode=O.MODE_IEEE):
    """Inputs of a Scene::TraceTransparency call as the reference's RayTrace makes it (src/scene_trace.cpp:468-477): the packets of a frame,
    their hit distances and triangle ids, a seeded selector (random lane sets, some packets fully selected -- the RayGroup<0,0> branch of
    :631 --, some not at all) and one light."""
    t, u, v, tid, _ = osc.render_primary(cam.as_array13(), resx, resy, mode=mode)
    xy = np.array([(x, y) for y in range(0, resy, 16) for x in range(0, resx, 16)], dtype=np.int32)
    tp, ip = frame_to_packets(t, xy), frame_to_packets(tid, xy)
    if resx % 16 or resy % 16:      # rays of edge packets outside the image: a miss, as the device hands them back
        inside = frame_to_packets(np.ones_like(t), xy) > 0
        tp[~inside] = np.inf
    rng = np.random.RandomState(seed)
    sel = rng.randint(0, 16, size=(len(xy), 64)).astype(np.uint8)
    sel[::3] = 15
    sel[1::7] = 0
    bmin, bmax = osc.nodes[0]["bmin"], osc.nodes[0]["bmax"]
    c, e = (bmin + bmax) * 0.5, (bmax - bmin)
    lights = np.array([[c[0], c[1] + 0.35 * e[1], c[2], 1.0, 0.9, 0.8, 2.0 * float(e.max())]], dtype=np.float32)
    return xy, tp, ip, sel, lights


def non_nested_tree(hb, levels=(2, 9), shrink=0.35, every=2):
    """A caller tree whose child boxes stick out of their parents': the product builder's tree with the boxes of every `every`-th inner
    node on `levels` shrunk towards their centre (children untouched).  Such a tree is valid input of snail_scene_create ("caller arrays
    verbatim") and the reference walks it by rescanning the whole inherited quad range at every box (src/bounding_box.cpp:71-139)."""
    from snail_amd import HostBVH
    nodes = hb.nodes.copy()
    todo, k, changed = [(0, 0)], 0, 0
    while todo:
        i, lvl = todo.pop()
        sub = int(nodes["sub"][i])
        if sub & 0x80000000:
            continue
        todo += [(sub, lvl + 1), (sub + 1, lvl + 1)]
        if levels[0] <= lvl <= levels[1]:
            k += 1
            if k % every == 0:
                c = (nodes["bmin"][i] + nodes["bmax"][i]) * np.float32(0.5)
                nodes["bmin"][i] = (c + (nodes["bmin"][i] - c) * np.float32(1.0 - shrink)).astype(np.float32)
                nodes["bmax"][i] = (c + (nodes["bmax"][i] - c) * np.float32(1.0 - shrink)).astype(np.float32)
                changed += 1
    assert changed > 8
    return HostBVH(hb.tris, nodes, hb.depth, hb.perm)


def collapsed_tree(hb, max_leaf):
    """The product builder's tree with every subtree of at most `max_leaf` triangles collapsed into ONE leaf (triangles of a subtree are
    contiguous in the reference's layout), re-laid out breadth first so that children stay adjacent pairs starting at odd indices.
    Leaves of 5 .. max_leaf triangles: what the reference's builder produces only at BVH::maxDepth or where the SAH refuses to split."""
    from snail_amd import HostBVH
    nodes = hb.nodes
    sub = nodes["sub"].astype(np.int64)
    leaf = (sub & 0x80000000) != 0
    first = np.zeros(len(nodes), dtype=np.int64); cnt = np.zeros(len(nodes), dtype=np.int64)
    order = []
    stack = [0]
    while stack:                      # post-order by explicit stack
        i = stack.pop(); order.append(i)
        if not leaf[i]: stack += [int(sub[i]), int(sub[i]) + 1]
    for i in reversed(order):
        if leaf[i]: first[i], cnt[i] = int(sub[i] & 0x7fffffff), int(nodes["aux"][i])
        else:
            a, b = int(sub[i]), int(sub[i]) + 1
            assert first[a] + cnt[a] == first[b]
            first[i], cnt[i] = first[a], cnt[a] + cnt[b]
    out = [nodes[0].copy()]; queue = [(0, 0)]; depth = 0; level = {0: 0}
    while queue:
        i, slot = queue.pop(0)
        if leaf[i] or cnt[i] <= max_leaf:
            out[slot]["sub"] = np.uint32(0x80000000 | int(first[i])); out[slot]["aux"] = int(cnt[i])
            depth = max(depth, level[slot])
            continue
        a = int(sub[i]); k = len(out)
        out.append(nodes[a].copy()); out.append(nodes[a + 1].copy())
        out[slot]["sub"] = k
        level[k] = level[k + 1] = level[slot] + 1
        queue += [(a, k), (a + 1, k + 1)]
    new = np.array(out, dtype=nodes.dtype)
    assert int(new["aux"][(new["sub"] & 0x80000000) != 0].max()) > 4
    return HostBVH(hb.tris, new, depth, hb.perm)


def interleave16(b):
    """block -> slot of the kernels' built-in dispatch order (dev::interleave16: 16 consecutive slots per XCD turn)"""
    b = np.asarray(b)
    xcd, j = b & 7, b >> 3
    return (((j >> 4) << 3) + xcd) * 16 + (j & 15)


def check_derived_order(cost, order, adaptive=3, exact=False):
    """A dispatch order the library derived from per-slot costs (snail_order_from_cost_dev, the *_reorder_dev launches; up to 49152 slots): a permutation, and
    -- by the rule of dev::orderSortBlock -- the SORTED one (cost classes of max >> shift <= 4095, descending) when the costs are heavy-tailed (the class of the
    slot at the 99th percentile, times the slot count, at least `adaptive` times the sum of the costs) or the caller declared them exact (SNAIL_ORDER_SORTED), the kernels' BUILT-IN order otherwise.  Returns which."""
    c = np.minimum(np.maximum(np.asarray(cost, dtype=np.int64), 0), 65535)
    o = np.asarray(order, dtype=np.int64)
    n = len(c)
    assert np.array_equal(np.sort(o), np.arange(n)), "order is not a permutation"
    shift = 0
    while (int(c.max()) >> shift) > 4095:
        shift += 1
    cls = 4095 - np.minimum(c >> shift, 4095)                       # class 0 = heaviest
    counts = np.bincount(cls, minlength=4096)
    start = np.concatenate([[0], np.cumsum(counts)[:-1]])
    want = max(n // 100, 1)
    p99 = 4095
    for k in np.flatnonzero(counts):
        if start[k] < want <= start[k] + counts[k]:
            p99 = int(k)
    heavy_tailed = ((4095 - p99) << shift) * n >= adaptive * int(c.sum())
    if heavy_tailed or exact:
        assert (np.diff(cls[o]) >= 0).all(), "heavy-tailed or exact costs: cost classes must descend"
        return "sorted"
    n128 = (n + 127) // 128 * 128
    nat = interleave16(np.arange(n128))
    assert np.array_equal(o, nat[nat < n]), "costs without a heavy tail: the built-in dispatch order"
    return "built-in"
