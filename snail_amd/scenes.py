"""Host-side scene ingest and deterministic procedural scenes (numpy, float32, no FMA).

Scene ingest mirrors the parts of the reference that fix triangle ORDER and WINDING fed to the BVH
builder (they define triId):
  * quad faces become (v0,v1,v2),(v2,v3,v0)        -- src/formats/wavefront_obj.cpp:113-166
  * degenerate faces are dropped by swap-with-last  -- src/base_scene.cpp:173-184 (Object::Repair)
  * `rtracer` flips winding by default (swap v0,v1) -- src/rtracer.cpp:554-556, src/base_scene.cpp:326-335
  * object order x face order, no object transform  -- src/base_scene.cpp:39-57 (ToTriVector)

The reference checkout lacks sponza.obj / abrams.obj (.MISSING_LARGE_BLOBS), so configs 2-5 of
BASELINE.json run on the deterministic stand-ins generated here (`atrium`, `stress`); a real OBJ
dropped into scenes/ can be loaded with `load_obj` instead.
"""
from __future__ import annotations

import math
import os

import numpy as np

F32 = np.float32


# --------------------------------------------------------------------------------------------------
# OBJ ingest
# --------------------------------------------------------------------------------------------------
def _repair(tri_idx: np.ndarray, verts: np.ndarray) -> np.ndarray:
    """Object::Repair (src/base_scene.cpp:173-184): drop faces whose float32 cross product has all
    components < 1e-8 in magnitude, by swap-with-last (which perturbs face order)."""
    v0, v1, v2 = verts[tri_idx[:, 0]], verts[tri_idx[:, 1]], verts[tri_idx[:, 2]]
    e1, e2 = (v1 - v0).astype(F32), (v2 - v0).astype(F32)
    nx = (e1[:, 1] * e2[:, 2]).astype(F32) - (e1[:, 2] * e2[:, 1]).astype(F32)
    ny = (e1[:, 2] * e2[:, 0]).astype(F32) - (e1[:, 0] * e2[:, 2]).astype(F32)
    nz = (e1[:, 0] * e2[:, 1]).astype(F32) - (e1[:, 1] * e2[:, 0]).astype(F32)
    eps = F32(0.00000001)
    bad = (np.abs(nx) < eps) & (np.abs(ny) < eps) & (np.abs(nz) < eps)
    if not bad.any():
        return tri_idx
    order = list(range(len(tri_idx)))
    badl = bad.tolist()
    n = 0
    while n < len(order):
        if badl[order[n]]:
            order[n] = order[-1]
            order.pop()
        else:
            n += 1
    return tri_idx[np.asarray(order, dtype=np.int64)]


def _libc_strtof():
    """sscanf("%f") of the reference (src/formats/wavefront_obj.cpp:98-101) rounds the decimal text to float32 ONCE; Python's
    float() rounds to float64 first, and the second rounding to float32 can land on the other neighbour when the text sits
    within 2^-29 (relative) of a float32 midpoint.  The C library's strtof is the same conversion as sscanf's."""
    import ctypes
    import ctypes.util
    try:
        libc = ctypes.CDLL(ctypes.util.find_library("c") or "libc.so.6")
        fn = libc.strtof
        fn.restype = ctypes.c_float
        fn.argtypes = [ctypes.c_char_p, ctypes.c_void_p]
        return lambda tok: fn(tok.encode("ascii", "replace"), None)
    except (OSError, AttributeError):   # no C library to bind: double rounding (differs only in the rare midpoint case)
        return lambda tok: float(np.float32(float(tok)))


def load_obj(path: str, flip: bool = True, swap_yz: bool = False) -> np.ndarray:
    """LoadWavefrontObj -> Repair -> (FlipNormals) -> ToTriVector: returns float32 [n,3,3] vertex
    positions in the order the reference hands them to BVH::Construct."""
    verts: list[tuple[float, float, float]] = []
    faces: list[tuple[int, int, int]] = []
    strtof = _libc_strtof()
    with open(path, "r", errors="replace") as fh:
        for line in fh:
            parts = line.split()
            if not parts:
                continue
            if parts[0] == "v" and len(parts) >= 4:
                verts.append((strtof(parts[1]), strtof(parts[2]), strtof(parts[3])))
            elif parts[0] == "f" and len(parts) >= 4:
                idx = []
                for tok in parts[1:5]:
                    i = int(tok.split("/")[0])
                    if i < 0:
                        i = len(verts) + i + 1
                    idx.append(i - 1)
                faces.append((idx[0], idx[1], idx[2]))
                if len(idx) == 4:
                    faces.append((idx[2], idx[3], idx[0]))
    v = np.asarray(verts, dtype=np.float64).astype(F32)
    f = np.asarray(faces, dtype=np.int64)
    if len(f) and (f.min() < 0 or f.max() >= len(v)):
        raise ValueError("Wrong vertex index in %s" % path)
    f = _repair(f, v)
    if flip:          # Object::FlipNormals (src/base_scene.cpp:326-335): swap the first two corners
        f = f[:, [1, 0, 2]]
    if swap_yz:       # Object::SwapYZ (src/base_scene.cpp:337-343)
        v = np.ascontiguousarray(v[:, [0, 2, 1]])
    return np.ascontiguousarray(v[f])


def save_obj(path: str, tri_verts: np.ndarray) -> None:
    """Write triangles as an OBJ with 10 decimals: exact for the generators' 1/1024 coordinate grid, so
    the text round-trips bit-exactly through sscanf("%f") (src/formats/wavefront_obj.cpp:98-101)."""
    tv = np.asarray(tri_verts, dtype=F32).reshape(-1, 3)
    with open(path, "w") as fh:
        for p in tv:
            fh.write("v %.10f %.10f %.10f\n" % (p[0], p[1], p[2]))
        for i in range(len(tv) // 3):
            fh.write("f %d %d %d\n" % (3 * i + 1, 3 * i + 2, 3 * i + 3))


# --------------------------------------------------------------------------------------------------
# box: numerical restatement of scenes/box.obj:15-36 (8 vertices / 12 faces)
# --------------------------------------------------------------------------------------------------
_BOX_V = [
    (1.0, -1.0, -1.0), (1.0, -1.0, 1.0), (-1.0, -1.0, 1.0), (-1.0, -1.0, -1.0),
    (1.0, 1.0, -1.0), (0.999999, 1.0, 1.000001), (-1.0, 1.0, 1.0), (-1.0, 1.0, -1.0),
]
_BOX_F = [
    (5, 1, 4), (5, 4, 8), (3, 7, 8), (3, 8, 4), (2, 6, 3), (6, 7, 3),
    (1, 5, 2), (5, 6, 2), (5, 8, 6), (8, 7, 6), (1, 2, 3), (1, 3, 4),
]


def box_scene(flip: bool = True) -> np.ndarray:
    v = np.asarray(_BOX_V, dtype=np.float64).astype(F32)
    f = np.asarray(_BOX_F, dtype=np.int64) - 1
    if flip:
        f = f[:, [1, 0, 2]]
    return np.ascontiguousarray(v[f])


# --------------------------------------------------------------------------------------------------
# procedural stand-ins.  All coordinates are snapped to a 1/1024 grid so 10-decimal text round-trips.
# --------------------------------------------------------------------------------------------------
def _snap(a: np.ndarray) -> np.ndarray:
    return (np.round(np.asarray(a, dtype=np.float64) * 1024.0) / 1024.0).astype(F32)


def _grid_quads(p00, du, dv, nu, nv, disp=None):
    """Tessellated parallelogram -> [2*nu*nv,3,3].  disp(u,v)->(n,3) optional displacement."""
    us, vs = np.meshgrid(np.arange(nu + 1), np.arange(nv + 1), indexing="ij")
    p = (np.asarray(p00, dtype=np.float64)[None, None, :]
         + us[..., None] * np.asarray(du, dtype=np.float64)[None, None, :]
         + vs[..., None] * np.asarray(dv, dtype=np.float64)[None, None, :])
    if disp is not None:
        p = p + disp(us / nu, vs / nv)
    a, b, c, d = p[:-1, :-1], p[1:, :-1], p[1:, 1:], p[:-1, 1:]
    t1 = np.stack([a, b, c], axis=-2).reshape(-1, 3, 3)
    t2 = np.stack([c, d, a], axis=-2).reshape(-1, 3, 3)
    out = np.empty((t1.shape[0] * 2, 3, 3), dtype=np.float64)
    out[0::2], out[1::2] = t1, t2
    return out


def _cylinder(cx, cz, y0, y1, radius, nseg, nrows, flute=0.0):
    ang = np.arange(nseg + 1) * (2.0 * math.pi / nseg)
    ys = np.linspace(y0, y1, nrows + 1)
    rr = radius * (1.0 + flute * np.cos(ang * 12.0))
    x = cx + rr[:, None] * np.cos(ang)[:, None] + 0.0 * ys[None, :]
    z = cz + rr[:, None] * np.sin(ang)[:, None] + 0.0 * ys[None, :]
    y = 0.0 * ang[:, None] + ys[None, :]
    p = np.stack([x, y, z], axis=-1)
    a, b, c, d = p[:-1, :-1], p[1:, :-1], p[1:, 1:], p[:-1, 1:]
    t1 = np.stack([a, b, c], axis=-2).reshape(-1, 3, 3)
    t2 = np.stack([c, d, a], axis=-2).reshape(-1, 3, 3)
    return np.concatenate([t1, t2], axis=0)


def _arch(x0, x1, z, y0, rise, thick, nseg, nrows):
    """Half-circle arch band between two columns (a curved strip extruded in z)."""
    t = np.linspace(0.0, math.pi, nseg + 1)
    xm, r = 0.5 * (x0 + x1), 0.5 * (x1 - x0)
    xs = xm - r * np.cos(t)
    ys = y0 + rise * np.sin(t)
    zs = np.linspace(z - thick, z + thick, nrows + 1)
    p = np.stack([xs[:, None] + 0 * zs[None, :], ys[:, None] + 0 * zs[None, :], 0 * xs[:, None] + zs[None, :]], axis=-1)
    a, b, c, d = p[:-1, :-1], p[1:, :-1], p[1:, 1:], p[:-1, 1:]
    t1 = np.stack([a, b, c], axis=-2).reshape(-1, 3, 3)
    t2 = np.stack([c, d, a], axis=-2).reshape(-1, 3, 3)
    return np.concatenate([t1, t2], axis=0)


ATRIUM_DETAIL = 0.94   # -> 263 124 triangles after Repair (sponza.obj has ~262 K)


def atrium(seed: int = 1, detail: float = ATRIUM_DETAIL) -> np.ndarray:
    """`atrium-262k`: sponza stand-in: a 48 x 16 x 20 nave with a
    tessellated floor/ceiling/walls, two storeys of colonnades (fluted cylinders), arches between
    columns, and draped cloth quads.  detail=ATRIUM_DETAIL gives ~263 K triangles; deterministic in `seed`."""
    rng = np.random.RandomState(seed)
    k = math.sqrt(detail)
    parts = []
    L, W, H = 48.0, 20.0, 16.0
    x0, z0 = -L / 2, -W / 2

    def n(v):
        return max(2, int(round(v * k)))

    # floor with gentle cobble displacement, ceiling, four walls
    def cobble(u, v):
        h = 0.03 * np.sin(u * 150.0) * np.sin(v * 70.0)
        return np.stack([0 * h, h, 0 * h], axis=-1)

    parts.append(_grid_quads((x0, 0, z0), (L / n(176), 0, 0), (0, 0, W / n(88)), n(176), n(88), cobble))
    parts.append(_grid_quads((x0, H, z0), (0, 0, W / n(40)), (L / n(96), 0, 0), n(40), n(96)))
    parts.append(_grid_quads((x0, 0, z0), (0, H / n(48), 0), (L / n(128), 0, 0), n(48), n(128)))
    parts.append(_grid_quads((x0, 0, -z0), (L / n(128), 0, 0), (0, H / n(48), 0), n(128), n(48)))
    parts.append(_grid_quads((x0, 0, z0), (0, 0, W / n(56)), (0, H / n(48), 0), n(56), n(48)))
    parts.append(_grid_quads((-x0, 0, z0), (0, H / n(48), 0), (0, 0, W / n(56)), n(48), n(56)))
    # gallery floors (second storey) along both sides
    for zs in (z0, -z0 - 4.0):
        parts.append(_grid_quads((x0, 7.0, zs), (L / n(96), 0, 0), (0, 0, 4.0 / n(8)), n(96), n(8)))
        parts.append(_grid_quads((x0, 6.5, zs), (0, 0, 4.0 / n(8)), (L / n(96), 0, 0), n(8), n(96)))
    # colonnades: 2 rows x 2 storeys x 12 columns, arches between neighbours
    ncol = 12
    xs = np.linspace(x0 + 3.0, -x0 - 3.0, ncol)
    for zrow in (z0 + 4.0, -z0 - 4.0):
        for (ya, yb, rad) in ((0.0, 5.0, 0.55), (7.0, 12.0, 0.45)):
            for cx in xs:
                parts.append(_cylinder(cx, zrow, ya, yb, rad, n(40), n(36), flute=0.04))
                parts.append(_cylinder(cx, zrow, ya, ya + 0.4, rad * 1.5, n(24), 2))
                parts.append(_cylinder(cx, zrow, yb, yb + 0.4, rad * 1.5, n(24), 2))
            for i in range(ncol - 1):
                parts.append(_arch(xs[i], xs[i + 1], zrow, yb + 0.4, 1.1, 0.35, n(36), n(6)))
    # draped cloths hanging across the nave
    for i in range(6):
        cx = x0 + 6.0 + i * 7.0 + float(rng.uniform(-0.5, 0.5))
        ph = float(rng.uniform(0, 6.28))

        def drape(u, v, ph=ph):
            sag = -2.2 * np.sin(np.pi * v) - 0.15 * np.sin(u * 25.0 + ph) * np.sin(v * 31.0)
            return np.stack([0.12 * np.sin(v * 40.0 + ph), sag, 0 * sag], axis=-1)

        parts.append(_grid_quads((cx, 13.5, z0 + 4.0), (2.5 / n(24), 0, 0), (0, 0, (W - 8.0) / n(96)), n(24), n(96), drape))
    # a central fountain-ish lathe object and scattered blocks
    parts.append(_cylinder(0.0, 0.0, 0.0, 1.2, 2.2, n(96), n(10), flute=0.02))
    parts.append(_cylinder(0.0, 0.0, 1.2, 3.0, 0.5, n(48), n(16)))
    tris = np.concatenate(parts, axis=0)
    return np.ascontiguousarray(_snap(tris))


def atrium_camera() -> "tuple":
    """Interior camera looking down the nave (pos, ang, pitch) for FPSCamera semantics."""
    return (np.array([-21.0, 4.2, 1.3], dtype=F32), -math.pi / 2 + 0.12, -0.08)


def stress(seed: int = 7, detail: float = 1.0) -> np.ndarray:
    """`stress-1M`: two merged generator meshes (~1.0 M triangles at detail=1): a densely displaced
    terrain plus a cloud of tessellated tori, for the deep-BVH study (config 5)."""
    rng = np.random.RandomState(seed)
    k = math.sqrt(detail)

    def n(v):
        return max(2, int(round(v * k)))

    parts = []

    def terrain(u, v):
        h = (1.8 * np.sin(u * 9.0) * np.cos(v * 7.0) + 0.6 * np.sin(u * 41.0 + 1.3) * np.sin(v * 37.0)
             + 0.15 * np.sin(u * 173.0) * np.cos(v * 191.0))
        return np.stack([0 * h, h, 0 * h], axis=-1)

    parts.append(_grid_quads((-40, 0, -40), (80.0 / n(560), 0, 0), (0, 0, 80.0 / n(560)), n(560), n(560), terrain))
    for i in range(24):
        c = rng.uniform(-30, 30, size=3)
        c[1] = rng.uniform(3, 14)
        R, r = rng.uniform(1.5, 3.5), rng.uniform(0.3, 0.9)
        nu, nv = n(128), n(60)
        u = np.arange(nu + 1) * (2 * math.pi / nu)
        v = np.arange(nv + 1) * (2 * math.pi / nv)
        uu, vv = np.meshgrid(u, v, indexing="ij")
        rot = rng.uniform(0, math.pi)
        x = (R + r * np.cos(vv)) * np.cos(uu)
        y = r * np.sin(vv)
        z = (R + r * np.cos(vv)) * np.sin(uu)
        y2 = y * math.cos(rot) - z * math.sin(rot)
        z2 = y * math.sin(rot) + z * math.cos(rot)
        p = np.stack([x + c[0], y2 + c[1], z2 + c[2]], axis=-1)
        a, b, cc, d = p[:-1, :-1], p[1:, :-1], p[1:, 1:], p[:-1, 1:]
        parts.append(np.concatenate([np.stack([a, b, cc], axis=-2).reshape(-1, 3, 3),
                                     np.stack([cc, d, a], axis=-2).reshape(-1, 3, 3)], axis=0))
    tris = np.concatenate(parts, axis=0)
    return np.ascontiguousarray(_snap(tris))


def stress_camera() -> "tuple":
    return (np.array([-34.0, 9.0, -30.0], dtype=F32), -0.85, 0.2)   # looking across the terrain, slightly down


def chain(n: int = 400, ratio: float = 1.2) -> np.ndarray:
    """Adversarial scene for the builder and the traversal stack: a geometric chain of shrinking triangles.
    The SAH sweep peels one triangle per level, so the tree reaches BVH::maxDepth - 1 = 63 (where a leaf is
    forced, src/bvh/tree.cpp:54) and that last leaf holds far more than 4 (here > 64) triangles."""
    tris = []
    for i in range(n):
        s = ratio ** (-i)
        x = 10.0 * s
        y, z = 2.0 * s * ((i % 3) - 1), 2.0 * s * (((i // 3) % 3) - 1)      # sway so that many triangles are visible
        tris.append([[x, y, z], [x + 0.3 * s, y, z + 2.0 * s], [x, y + 2.0 * s, z]])
    return np.asarray(tris, dtype=F32)


def patches(n: int = 6) -> np.ndarray:
    """Test scene for the narrow-range leaf forms: an n x n board of axis-aligned squares in the plane z = 0 whose sides grow from a few
    percent of a cell to a whole cell, each tessellated into 2 x 2 x 2 triangles, plus a clump of 100 coincident triangles in front of one cell
    (tests collapse its subtree into ONE leaf of more than 64 triangles that a narrow range reaches).  Seen head-on at 16 pixels per cell the
    squares cover 1 .. 64 quads of a packet."""
    tris = []
    for j in range(n):
        for i in range(n):
            k = j * n + i
            side = 0.04 + 0.96 * k / (n * n - 1)
            x0, y0 = i + 0.5 - side / 2, j + 0.5 - side / 2
            h = side / 2
            for b in range(2):
                for a in range(2):
                    xa, ya = x0 + a * h, y0 + b * h
                    tris.append([[xa, ya, 0.0], [xa + h, ya, 0.0], [xa + h, ya + h, 0.0]])
                    tris.append([[xa + h, ya + h, 0.0], [xa, ya + h, 0.0], [xa, ya, 0.0]])
    for _ in range(100):
        tris.append([[0.30, 0.30, -0.25], [0.55, 0.32, -0.25], [0.40, 0.52, -0.25]])
    return np.asarray(tris, dtype=F32)


OFFGRID_ORIGIN = (137.731, -61.377, 419.173)


def offgrid(seed: int = 11, n_blob: int = 4000) -> np.ndarray:
    """Geometry OFF every grid (the procedural scenes above are snapped to k/1024, where `ba`, `ca`, box extents and many `bmin - o` terms are
    exact; a scanned mesh like the reference's lancia.obj is not): every coordinate is a float64 random value rounded once to float32 -- full
    mantissas -- around an origin far from zero (|x| ~ 60 .. 420: one ulp is 4e-6 .. 3e-5, so sums and differences round), with
      * a blob of randomly oriented triangles whose sizes span four decades (1e-3 .. 10 units),
      * slivers (aspect ratio ~1e4) in random directions,
      * a wavy sheet tessellated with irrational steps (coherent surfaces for the packets to narrow on),
      * shingles: near-coplanar triangles overlapping their neighbours with offsets of a few ulps (near-ties in t between candidates).
    Deterministic in `seed`; ~10 K triangles."""
    rng = np.random.RandomState(seed)
    O = np.asarray(OFFGRID_ORIGIN, dtype=np.float64)
    parts = []
    # blob: centre in a 30-unit cube, size log-uniform over 1e-3 .. 10
    c = O[None, :] + rng.uniform(-15.0, 15.0, size=(n_blob, 3))
    size = np.exp(rng.uniform(math.log(1e-3), math.log(10.0), size=n_blob))
    e = rng.randn(n_blob, 3, 3)
    parts.append(c[:, None, :] + e * size[:, None, None] * 0.5)
    # slivers: a long edge of 2 .. 8 units, the third vertex 1e-4 .. 1e-3 of that off its middle
    ns = 1500
    a = O[None, :] + rng.uniform(-15.0, 15.0, size=(ns, 3))
    d = rng.randn(ns, 3); d /= np.linalg.norm(d, axis=1, keepdims=True)
    ln = rng.uniform(2.0, 8.0, size=ns)
    w = rng.randn(ns, 3); w -= (w * d).sum(axis=1, keepdims=True) * d; w /= np.linalg.norm(w, axis=1, keepdims=True)
    b = a + d * ln[:, None]
    m = a + d * (ln * rng.uniform(0.2, 0.8, size=ns))[:, None] + w * (ln * np.exp(rng.uniform(math.log(1e-4), math.log(1e-3), size=ns)))[:, None]
    parts.append(np.stack([a, b, m], axis=1))
    # wavy sheet, steps pi / 29 and e / 23, through the middle of the blob
    nu, nv = 44, 36
    us, vs = np.meshgrid(np.arange(nu + 1) * (math.pi / 29.0), np.arange(nv + 1) * (math.e / 23.0), indexing="ij")
    p = np.stack([O[0] - 2.3 + us, O[1] - 1.9 + vs + 0.21 * np.sin(us * 3.1), O[2] + 0.37 * np.sin(us * 2.3) * np.cos(vs * 1.7) + 0.05 * us], axis=-1)
    qa, qb, qc, qd = p[:-1, :-1], p[1:, :-1], p[1:, 1:], p[:-1, 1:]
    parts.append(np.concatenate([np.stack([qa, qb, qc], axis=-2).reshape(-1, 3, 3), np.stack([qc, qd, qa], axis=-2).reshape(-1, 3, 3)], axis=0))
    # shingles: rows of overlapping triangles in ONE plane (a random tilt), each a few ulps of the plane's distance above the last
    nsh = 1200
    nrm = np.array([0.31, 0.83, -0.46]); nrm /= np.linalg.norm(nrm)
    t1 = np.cross(nrm, [0.0, 0.0, 1.0]); t1 /= np.linalg.norm(t1); t2 = np.cross(nrm, t1)
    base = O + np.array([3.7, -2.1, -6.3])
    k = np.arange(nsh)
    org = base[None, :] + t1[None, :] * (0.173 * (k % 40))[:, None] + t2[None, :] * (0.291 * (k // 40))[:, None] + nrm[None, :] * (k * 6.0e-5)[:, None]
    parts.append(np.stack([org, org + t1 * 0.41 + t2 * 0.07, org + t1 * 0.11 + t2 * 0.53], axis=1))
    return np.ascontiguousarray(np.concatenate(parts, axis=0).astype(F32))


def drop_degenerate(tri_verts: np.ndarray) -> np.ndarray:
    """Apply Object::Repair to an explicit triangle soup (generators can emit zero-area faces)."""
    tv = np.asarray(tri_verts, dtype=F32)
    flat = tv.reshape(-1, 3)
    idx = np.arange(len(flat), dtype=np.int64).reshape(-1, 3)
    keep = _repair(idx, flat)
    return np.ascontiguousarray(flat[keep])


def scene_by_name(name: str, scenes_dir: str | None = None) -> np.ndarray:
    """Resolve a workload name: 'box', 'atrium', 'atrium:<detail>', 'stress', 'stress:<detail>', or
    a path / file name of an OBJ (looked up in scenes_dir)."""
    if name == "box":
        return box_scene()
    if name.startswith("atrium"):
        d = float(name.split(":")[1]) if ":" in name else ATRIUM_DETAIL
        return drop_degenerate(atrium(detail=d))
    if name.startswith("stress"):
        d = float(name.split(":")[1]) if ":" in name else 1.0
        return drop_degenerate(stress(detail=d))
    if name.startswith("chain"):
        return chain()
    if name.startswith("patches"):
        return patches()
    if name.startswith("offgrid"):
        return drop_degenerate(offgrid())
    path = name
    if not os.path.exists(path) and scenes_dir:
        path = os.path.join(scenes_dir, name)
    return load_obj(path)
