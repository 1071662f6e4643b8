"""ctypes binding of oracle/liboracle.so -- TEST INFRASTRUCTURE (the checker, never the product)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "liboracle.so")

MODE_IEEE, MODE_SSE, MODE_TABLE = 0, 1, 2      # MODE_TABLE: the SSE definitions over the rcpps / rsqrtps tables given to set_tables()

TRI_DTYPE = np.dtype([("a", "<f4", 3), ("ba", "<f4", 3), ("ca", "<f4", 3), ("t0", "<f4"), ("it0", "<f4"),
                      ("pad", "<i4"), ("plane", "<f4", 4)])
NODE_DTYPE = np.dtype([("bmin", "<f4", 3), ("bmax", "<f4", 3), ("sub", "<u4"), ("aux", "<i4")])
assert TRI_DTYPE.itemsize == 64 and NODE_DTYPE.itemsize == 32

_lib = None


def build():
    srcs = [os.path.join(ORACLE_DIR, f) for f in ("snail_oracle.cpp", "snail_sse4.inc", "snail_oracle.h")]
    if (not os.path.exists(LIB_PATH)) or os.path.getmtime(LIB_PATH) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "liboracle.so"], stdout=subprocess.DEVNULL)


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB_PATH)
        vp, i32, u64p = C.c_void_p, C.c_int, C.c_void_p
        L.orc_tris_from_verts.argtypes = [vp, i32, vp]
        L.orc_bvh_build.argtypes = [vp, i32, vp, vp, vp]
        L.orc_bvh_build.restype = i32
        L.orc_fnv_nodes.argtypes = [vp, i32]
        L.orc_fnv_nodes.restype = C.c_uint64
        L.orc_fnv_tris.argtypes = [vp, i32]
        L.orc_fnv_tris.restype = C.c_uint64
        L.orc_gen_packet.argtypes = [vp, i32, i32, i32, i32, i32, vp, vp]
        L.orc_trace_rays.argtypes = [vp, vp, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, u64p, i32]
        L.orc_trace_shadow.argtypes = [vp, vp, i32, i32, vp, vp, vp, vp, u64p, i32]
        L.orc_render_primary.argtypes = [vp, vp, vp, i32, i32, i32, i32, i32, i32, vp, vp, vp, vp, u64p, i32, i32]
        L.orc_render_primary_sse4.argtypes = [vp, vp, vp, i32, i32, i32, i32, i32, i32, vp, vp, vp, vp, u64p, i32]
        L.orc_account_primary.argtypes = [vp, vp, vp, i32, i32, i32, i32, i32, i32, u64p, i32, i32]
        L.orc_shade_depth.argtypes = [vp, i32, vp, i32]
        L.orc_planar_encode_tile.argtypes = [vp, i32, i32, i32, i32, i32, vp]
        L.orc_planar_decode_tile.argtypes = [vp, i32, i32, i32, i32, vp, i32]
        L.orc_render_whitted.argtypes = [vp, vp, vp, i32, i32, vp, i32, vp, vp, i32, vp, i32, u64p, i32, i32]
        L.orc_veclib_exprs.argtypes = [vp, vp, i32]
        L.orc_trace_transparency.argtypes = [vp, vp, vp, i32, i32, vp, i32, vp, vp, vp, i32, vp, vp, vp, u64p, i32]
        for f in (L.orc_inv, L.orc_rsqrt):
            f.argtypes = [C.c_float, i32]
            f.restype = C.c_float
        L.orc_set_tables.argtypes = [vp]
        L.orc_tables_of_this_cpu.argtypes = [vp]
        L.orc_raw_approx.argtypes = [i32, i32, vp, vp, i32]
        for f in (L.orc_min, L.orc_max):
            f.argtypes = [C.c_float, C.c_float]
            f.restype = C.c_float
        _lib = L
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def set_tables(tables) -> None:
    """MODE_TABLE's rcpps / rsqrtps tables (uint32 [3, 4096]: tests/golden/rcp_tables.npz entries, or tables_of_this_cpu()); process-wide."""
    tab = np.ascontiguousarray(tables, dtype=np.uint32).reshape(3 * 4096)
    lib().orc_set_tables(_p(tab))


def tables_of_this_cpu() -> np.ndarray:
    tab = np.zeros((3, 4096), dtype=np.uint32)
    lib().orc_tables_of_this_cpu(_p(tab))
    return tab


def raw_approx(fn: int, bits: np.ndarray, table: bool) -> np.ndarray:
    """rcpps (fn 0) / rsqrtps (fn 1) of float bit patterns: by MODE_TABLE's rule over the tables in force, or by this host's instruction."""
    x = np.ascontiguousarray(bits, dtype=np.uint32)
    out = np.zeros_like(x)
    lib().orc_raw_approx(fn, 1 if table else 0, _p(x), _p(out), len(x))
    return out


def tris_from_verts(tri_verts: np.ndarray) -> np.ndarray:
    tv = np.ascontiguousarray(tri_verts, dtype=np.float32).reshape(-1, 9)
    out = np.zeros(len(tv), dtype=TRI_DTYPE)
    lib().orc_tris_from_verts(_p(tv), len(tv), _p(out))
    return out


class OracleScene:
    """tris (BVH order), nodes, depth, perm (BVH index -> input index)."""

    def __init__(self, tri_verts: np.ndarray):
        self.tris = tris_from_verts(tri_verts)
        n = len(self.tris)
        nodes = np.zeros(2 * n + 2, dtype=NODE_DTYPE)
        self.perm = np.zeros(n, dtype=np.int32)
        depth = C.c_int(0)
        nn = lib().orc_bvh_build(_p(self.tris), n, _p(nodes), C.byref(depth), _p(self.perm))
        self.nodes = np.ascontiguousarray(nodes[:nn])
        self.depth = depth.value

    @classmethod
    def from_arrays(cls, tris, nodes, depth, perm=None):
        """An oracle scene over a CALLER's tree (hand-made test trees): the same walk, no build."""
        self = cls.__new__(cls)
        self.tris = np.ascontiguousarray(np.asarray(tris).view(TRI_DTYPE))
        self.nodes = np.ascontiguousarray(np.asarray(nodes).view(NODE_DTYPE))
        self.depth, self.perm = int(depth), perm
        return self

    def fnv_nodes(self) -> int:
        return lib().orc_fnv_nodes(_p(self.nodes), len(self.nodes))

    def fnv_tris(self) -> int:
        return lib().orc_fnv_tris(_p(self.tris), len(self.tris))

    def render_primary(self, cam13: np.ndarray, resx, resy, rect=None, mode=MODE_IEEE, threads=8):
        x0, y0, w, h = rect if rect else (0, 0, resx, resy)
        cam = np.ascontiguousarray(cam13, dtype=np.float32)
        t = np.full((resy, resx), np.nan, dtype=np.float32)
        u = np.zeros((resy, resx), dtype=np.float32)
        v = np.zeros((resy, resx), dtype=np.float32)
        tid = np.zeros((resy, resx), dtype=np.int32)
        stats = np.zeros(4, dtype=np.uint64)
        lib().orc_render_primary(_p(self.nodes), _p(self.tris), _p(cam), resx, resy, x0, y0, w, h,
                                 _p(t), _p(u), _p(v), _p(tid), _p(stats), mode, threads)
        return t, u, v, tid, stats

    def render_primary_sse4(self, cam13: np.ndarray, resx, resy, rect=None, threads=8):
        """render_primary(mode=MODE_SSE) through the 4-wide SSE-intrinsics port (orc_render_primary_sse4): same outputs, same TreeStats"""
        x0, y0, w, h = rect if rect else (0, 0, resx, resy)
        cam = np.ascontiguousarray(cam13, dtype=np.float32)
        t = np.full((resy, resx), np.nan, dtype=np.float32)
        u = np.zeros((resy, resx), dtype=np.float32)
        v = np.zeros((resy, resx), dtype=np.float32)
        tid = np.zeros((resy, resx), dtype=np.int32)
        stats = np.zeros(4, dtype=np.uint64)
        lib().orc_render_primary_sse4(_p(self.nodes), _p(self.tris), _p(cam), resx, resy, x0, y0, w, h, _p(t), _p(u), _p(v), _p(tid), _p(stats), threads)
        return t, u, v, tid, stats

    def account_primary(self, cam13, resx, resy, rect=None, mode=MODE_IEEE, threads=8):
        x0, y0, w, h = rect if rect else (0, 0, resx, resy)
        cam = np.ascontiguousarray(cam13, dtype=np.float32)
        out = np.zeros(4, dtype=np.uint64)
        lib().orc_account_primary(_p(self.nodes), _p(self.tris), _p(cam), resx, resy, x0, y0, w, h, _p(out), mode, threads)
        return out

    def render_whitted(self, cam13, resx, resy, lights7, ambient=(0.1, 0.1, 0.1), color=(1.0, 1.0, 1.0), mode=MODE_IEEE, threads=8, reflections=False,
                       antialias=False, depth=False):
        cam = np.ascontiguousarray(cam13, dtype=np.float32)
        lights = np.ascontiguousarray(lights7, dtype=np.float32).reshape(-1, 7)
        amb = np.asarray(ambient, dtype=np.float32); col = np.asarray(color, dtype=np.float32)
        frame = np.zeros((resy, resx, 3), dtype=np.uint8)
        stats = np.zeros(4, dtype=np.uint64)
        lib().orc_render_whitted(_p(self.nodes), _p(self.tris), _p(cam), resx, resy, _p(lights), len(lights), _p(amb), _p(col), (1 if reflections else 0) | (2 if antialias else 0) | (4 if depth else 0), _p(frame), resx * 3,
                                 _p(stats), mode, threads)
        return frame, stats

    def trace_transparency(self, cam13, resx, resy, packet_xy, t_packets, sel, lights7, ambient=(0.1, 0.1, 0.1), color=(1.0, 1.0, 1.0), mode=MODE_IEEE):
        cam = np.ascontiguousarray(cam13, dtype=np.float32)
        xy = np.ascontiguousarray(packet_xy, dtype=np.int32).reshape(-1, 2)
        lights = np.ascontiguousarray(lights7, dtype=np.float32).reshape(-1, 7)
        amb = np.asarray(ambient, dtype=np.float32); col = np.asarray(color, dtype=np.float32)
        t = np.ascontiguousarray(t_packets, dtype=np.float32); s = np.ascontiguousarray(sel, dtype=np.uint8)
        out = np.zeros((len(xy), 256, 3), dtype=np.float32)
        stats = np.zeros(4, dtype=np.uint64)
        lib().orc_trace_transparency(_p(self.nodes), _p(self.tris), _p(cam), resx, resy, _p(xy), len(xy), _p(t), _p(s), _p(lights), len(lights), _p(amb), _p(col),
                                     _p(out), _p(stats), mode)
        return out, stats

    def trace_rays(self, origin, dir, idir, mask, distance, obj, bary, npackets, size, shared, mode=MODE_IEEE):
        stats = np.zeros(4, dtype=np.uint64)
        lib().orc_trace_rays(_p(self.nodes), _p(self.tris), npackets, size, int(shared), _p(origin), _p(dir), _p(idir),
                             _p(mask), _p(distance), _p(obj), _p(bary), _p(stats), mode)
        return stats

    def trace_shadow(self, origin, dir, idir, distance, npackets, size, mode=MODE_IEEE):
        stats = np.zeros(4, dtype=np.uint64)
        lib().orc_trace_shadow(_p(self.nodes), _p(self.tris), npackets, size, _p(origin), _p(dir), _p(idir), _p(distance),
                               _p(stats), mode)
        return stats


def gen_packet(cam13, resx, resy, px, py, mode=MODE_IEEE):
    cam = np.ascontiguousarray(cam13, dtype=np.float32)
    d = np.zeros(768, dtype=np.float32)
    i = np.zeros(768, dtype=np.float32)
    lib().orc_gen_packet(_p(cam), resx, resy, px, py, mode, _p(d), _p(i))
    return d, i


def shade_depth(t: np.ndarray, mode=MODE_IEEE) -> np.ndarray:
    tt = np.ascontiguousarray(t, dtype=np.float32).reshape(-1)
    out = np.zeros((len(tt), 3), dtype=np.uint8)
    lib().orc_shade_depth(_p(tt), len(tt), _p(out), mode)
    return out


def planar_encode(frame_bgr: np.ndarray, tiles) -> list:
    """per tile (x, y, w, h) the 3*w*h bytes of src/render.cpp:140-163"""
    f = np.ascontiguousarray(frame_bgr, dtype=np.uint8)
    out = []
    for x, y, w, h in np.asarray(tiles).reshape(-1, 4).tolist():
        o = np.zeros(3 * w * h, dtype=np.uint8)
        lib().orc_planar_encode_tile(_p(f), f.shape[1] * 3, x, y, w, h, _p(o))
        out.append(o)
    return out


def planar_decode(planes: list, tiles, resx: int, resy: int) -> np.ndarray:
    f = np.zeros((resy, resx, 3), dtype=np.uint8)
    for o, (x, y, w, h) in zip(planes, np.asarray(tiles).reshape(-1, 4).tolist()):
        lib().orc_planar_decode_tile(_p(np.ascontiguousarray(o)), x, y, w, h, _p(f), resx * 3)
    return f
