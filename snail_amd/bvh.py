"""Host-side BVH: the mirror of `class BVH` construction (src/bvh/tree.h:27-31, src/bvh/tree.cpp:293-328).

`HostBVH.build(tri_verts)` runs the C++ builder inside libsnailhip.so (snail_amd/csrc/bvh_build.cpp):
Triangle precompute + SAH full-sweep build with the reference's exact arithmetic, so that the
permuted triangle order -- which DEFINES triId (src/bvh/tree.cpp:118-120) -- is the reference's."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _lib

# record layouts of the reference (sizes verified in SURVEY.md section 8): Triangle = 64 B, BVH::Node = 32 B
TRI_DTYPE = np.dtype([("a", "<f4", 3), ("ba", "<f4", 3), ("ca", "<f4", 3), ("t0", "<f4"), ("it0", "<f4"),
                      ("pad", "<i4"), ("plane", "<f4", 4)])
NODE_DTYPE = np.dtype([("bmin", "<f4", 3), ("bmax", "<f4", 3), ("sub", "<u4"), ("aux", "<i4")])
assert TRI_DTYPE.itemsize == 64 and NODE_DTYPE.itemsize == 32


@dataclass
class HostBVH:
    tris: np.ndarray    # TRI_DTYPE [nTris], BVH order
    nodes: np.ndarray   # NODE_DTYPE [nNodes]
    depth: int
    perm: np.ndarray    # int32 [nTris]: BVH slot -> index in the input triangle list

    @staticmethod
    def triangles(tri_verts: np.ndarray) -> np.ndarray:
        tv = np.ascontiguousarray(tri_verts, dtype=np.float32).reshape(-1, 9)
        out = np.zeros(len(tv), dtype=TRI_DTYPE)
        _lib.check(_lib.lib().snail_tris_from_verts(_lib.ptr(tv), len(tv), _lib.ptr(out)), "snail_tris_from_verts")
        return out

    @staticmethod
    def build(tri_verts: np.ndarray) -> "HostBVH":
        tris = HostBVH.triangles(tri_verts)
        n = len(tris)
        if n == 0:
            raise _lib.SnailError("cannot build a BVH over zero triangles")
        nodes = np.zeros(2 * n + 2, dtype=NODE_DTYPE)
        perm = np.zeros(n, dtype=np.int32)
        nn, depth = C.c_int(0), C.c_int(0)
        _lib.check(_lib.lib().snail_bvh_build(_lib.ptr(tris), n, _lib.ptr(nodes), C.addressof(nn), C.addressof(depth),
                                              _lib.ptr(perm)), "snail_bvh_build")
        return HostBVH(tris, np.ascontiguousarray(nodes[:nn.value]), depth.value, perm)

    @property
    def n_tris(self) -> int:
        return len(self.tris)

    @property
    def n_nodes(self) -> int:
        return len(self.nodes)

    def bbox(self):
        """BVH::GetBBox (src/bvh/tree.h:32)."""
        return self.nodes[0]["bmin"].copy(), self.nodes[0]["bmax"].copy()

    def normal(self, elem: int) -> np.ndarray:
        """BVH::GetNormal (src/bvh/tree.h:40-42) = plane.xyz."""
        return self.tris[elem]["plane"][:3].copy()
