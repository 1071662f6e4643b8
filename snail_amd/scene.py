"""Device scene + the traversal entry points: the host-side mirror of the `AccStruct` concept that
`template <class AccStruct> class Scene` requires (src/scene.h:26-58, models src/bvh/tree.h:27-50).

  Scene.traverse_primary(ctx)    <-> BVH::TraversePrimary<sharedOrigin,hasMask>(Context&)  src/bvh/traverse.cpp:14-80
  Scene.traverse_shadow(ctx)     <-> BVH::TraverseShadow(ShadowContext&)                   src/bvh/traverse.cpp:82-149
  Scene.trace_primary(cam, ...)  <-> RayGenerator::Generate + SafeInv + TraversePrimary over a rect of 16x16 packets
                                     (the per-packet body of RenderTask::Work, src/render.cpp:58-62,112-115)

torch is used for device memory, streams and (in render.py) torch.distributed -- plumbing only; all
compute is in libsnailhip.so.  Tensors follow the reference's Context memory layouts (see
include/snail_hip.h)."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _lib
from .bvh import HostBVH
from .camera import Camera

PACKET_QUADS = 64       # 16x16 px, src/render.cpp:50-53
PACKET_DIM = 16


def _torch():
    import torch
    return torch


def _stream_ptr(stream=None):
    torch = _torch()
    if stream is not None:
        return C.c_void_p(stream.cuda_stream)
    # torch.cuda.current_stream() without a device walks through is_available() (~150 us per call on this stack); the raw
    # accessors are two C calls
    try:
        return C.c_void_p(torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice()))
    except AttributeError:      # another torch build: the documented path
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)


@dataclass
class HitFrame:
    """Row-major [resy, resx] hit records: miss = (+inf, 0, 0, 0) (src/scene_trace.cpp:112-115)."""
    t: "object"
    u: "object"
    v: "object"
    tri_id: "object"


@dataclass
class Context:
    """`struct Context<sharedOrigin,hasMask>` (src/ray_group.h:351-380) as device tensors.
    origin: [npackets,12] if shared else [npackets*size,12]; dir/idir: [npackets*size,12];
    mask: uint8 [npackets*size] or None; distance/object: [npackets*size,4]; barycentric: [npackets*size,8]."""
    origin: "object"
    dir: "object"
    idir: "object"
    distance: "object"
    object: "object"
    barycentric: "object"
    size: int = PACKET_QUADS
    shared_origin: bool = True
    mask: "object" = None

    @property
    def n_packets(self) -> int:
        return self.dir.shape[0] // self.size


@dataclass
class ShadowContext:
    """`struct ShadowContext` (src/ray_group.h:383-403): origin [npackets,3] (light), distance in/out."""
    origin: "object"
    dir: "object"
    idir: "object"
    distance: "object"
    size: int = PACKET_QUADS

    @property
    def n_packets(self) -> int:
        return self.dir.shape[0] // self.size


def host_sse_tables() -> np.ndarray:
    """The rcpps / rsqrtps tables in force for SNAIL_ARITH_HOST_SSE: uint32[3, 4096] (snail_host_sse_tables) -- this host CPU's unless
    set_arith_tables() gave others."""
    tab = np.zeros((3, 4096), dtype=np.uint32)
    _lib.check(_lib.lib().snail_host_sse_tables(_lib.ptr(tab)), "snail_host_sse_tables")
    return tab


def set_arith_tables(tables) -> None:
    """Tables of another CPU for SNAIL_ARITH_HOST_SSE (uint32[3, 4096]; None = this host's again): process-wide, in force for a scene from its next
    set_arith("host_sse") on (snail_arith_set_tables)."""
    if tables is None:
        _lib.check(_lib.lib().snail_arith_set_tables(None), "snail_arith_set_tables")
        return
    tab = np.ascontiguousarray(tables, dtype=np.uint32).reshape(3 * 4096)
    _lib.check(_lib.lib().snail_arith_set_tables(_lib.ptr(tab)), "snail_arith_set_tables")


class Scene:
    """A BVH resident in one GPU's HBM (the reference ships the same arrays to a render node in
    SendBVH, src/server.cpp:144-164)."""

    def __init__(self, bvh: HostBVH, device: int | None = None):
        torch = _torch()
        if not torch.cuda.is_available():
            raise _lib.SnailError("no HIP device available: the traversal path has no CPU fallback")
        self.device = torch.cuda.current_device() if device is None else int(device)
        self.bvh = bvh
        h = _lib.lib().snail_scene_create(_lib.ptr(bvh.nodes), bvh.n_nodes, _lib.ptr(bvh.tris), bvh.n_tris, bvh.depth, self.device)
        if not h:
            raise _lib.SnailError("snail_scene_create: %s" % _lib.lib().snail_last_error().decode())
        self._h = C.c_void_p(h)

    @classmethod
    def from_lbvh(cls, tri_verts: np.ndarray, device: int | None = None, max_leaf_tris: int = 4):
        """GPU builder option (NOT the parity tree): a linear BVH built and kept on the device
        (snail_scene_create_lbvh).  The returned scene carries .perm (triId -> input triangle), .build_ms (device time of the
        build) and a HostBVH downloaded from the device in .bvh (nodes incl. unused slots, tris in tree order)."""
        torch = _torch()
        if not torch.cuda.is_available():
            raise _lib.SnailError("no HIP device available: the traversal path has no CPU fallback")
        self = cls.__new__(cls)
        self.device = torch.cuda.current_device() if device is None else int(device)
        tv = np.ascontiguousarray(tri_verts, dtype=np.float32).reshape(-1, 9)
        perm = np.zeros(len(tv), dtype=np.int32)
        ms = C.c_float(0.0)
        h = _lib.lib().snail_scene_create_lbvh(_lib.ptr(tv), len(tv), self.device, int(max_leaf_tris), _lib.ptr(perm), C.addressof(ms))
        if not h:
            raise _lib.SnailError("snail_scene_create_lbvh: %s" % _lib.lib().snail_last_error().decode())
        self._h = C.c_void_p(h)
        self.perm, self.build_ms = perm, float(ms.value)
        nn, nt, depth, dev = C.c_int(0), C.c_int(0), C.c_int(0), C.c_int(0)
        _lib.check(_lib.lib().snail_scene_info(self._h, C.addressof(nn), C.addressof(nt), C.addressof(depth), C.addressof(dev)), "snail_scene_info")
        from .bvh import NODE_DTYPE, TRI_DTYPE
        nodes = np.zeros(nn.value, dtype=NODE_DTYPE)
        tris = np.zeros(nt.value, dtype=TRI_DTYPE)
        _lib.check(_lib.lib().snail_scene_download(self._h, _lib.ptr(nodes), _lib.ptr(tris)), "snail_scene_download")
        self.bvh = HostBVH(tris, nodes, depth.value, perm)
        return self

    def close(self):
        if getattr(self, "_h", None):
            _lib.lib().snail_scene_destroy(self._h)
            self._h = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    # ---- AccStruct-shaped accessors -----------------------------------------------------------
    def get_bbox(self):
        return self.bvh.bbox()

    def get_normal(self, elem: int, sub: int = 0):
        return self.bvh.normal(elem)

    def has_shading_data(self) -> bool:
        return False    # BVH::noShadingData: ShTriangle records stay with the host shader

    # ---- helpers ------------------------------------------------------------------------------
    def _dev(self):
        return _torch().device("cuda", self.device)

    def new_stats(self):
        """uint64[4] device accumulator {intersects, iters, rays, skips} (TreeStats, src/tree_stats.h:86-113)."""
        torch = _torch()
        return torch.zeros(4, dtype=torch.int64, device=self._dev())

    def alloc_frame(self, resx: int, resy: int) -> HitFrame:
        torch = _torch()
        d = self._dev()
        return HitFrame(torch.full((resy, resx), float("inf"), dtype=torch.float32, device=d),
                        torch.zeros((resy, resx), dtype=torch.float32, device=d),
                        torch.zeros((resy, resx), dtype=torch.float32, device=d),
                        torch.zeros((resy, resx), dtype=torch.int32, device=d))

    # ---- primary packets ----------------------------------------------------------------------
    def trace_primary(self, cam: Camera, resx: int, resy: int, rect=None, out: HitFrame | None = None, stats=None, stream=None, order=None,
                      slot_cost=None) -> HitFrame:
        """Trace every 16x16 packet of `rect` (x0,y0,w,h; default the whole image).  `order` / `slot_cost`: optional int32 device
        tensors of primary_slots(w, h) entries -- the dispatch-order feedback of snail_trace_primary_ordered_dev."""
        x0, y0, w, h = rect if rect is not None else (0, 0, resx, resy)
        out = out if out is not None else self.alloc_frame(resx, resy)
        cam13 = np.ascontiguousarray(cam.as_array13(), dtype=np.float32)
        if order is not None or slot_cost is not None:
            n = self.primary_slots(w, h)
            for a in (order, slot_cost):
                if a is not None and (a.numel() != n or a.dtype != _torch().int32 or not a.is_contiguous()):
                    raise ValueError(f"order / slot_cost must be contiguous int32 tensors of {n} entries")
            rc = _lib.lib().snail_trace_primary_ordered_dev(self._h, _lib.ptr(cam13), resx, resy, x0, y0, w, h, _lib.ptr(out.t), _lib.ptr(out.u),
                                                            _lib.ptr(out.v), _lib.ptr(out.tri_id), _lib.ptr(stats), _lib.ptr(order), _lib.ptr(slot_cost),
                                                            _stream_ptr(stream))
            _lib.check(rc, "snail_trace_primary_ordered_dev")
            return out
        rc = _lib.lib().snail_trace_primary_dev(self._h, _lib.ptr(cam13), resx, resy, x0, y0, w, h, _lib.ptr(out.t), _lib.ptr(out.u),
                                                _lib.ptr(out.v), _lib.ptr(out.tri_id), _lib.ptr(stats), _stream_ptr(stream))
        _lib.check(rc, "snail_trace_primary_dev")
        return out

    def trace_primary_batch(self, cams, resx: int, resy: int, outs, stats=None, stream=None, order=None, slot_cost=None, next_order=None, order_exact=False):
        """ONE launch for len(cams) frames (<= 8) of the whole image, each with its own camera and HitFrame (snail_trace_primary_batch_dev):
        the heaviest packets of all the frames first, one tail and one set of launch overheads for all of them.  next_order (int32 device tensor of
        primary_slots entries; may be `order` itself; needs slot_cost): the order the NEXT launch should use, derived from this launch's costs inside the
        launch (snail_trace_primary_batch_reorder_dev: no kernel launch of its own); order_exact = the caller knows the costs to be exact for the launches
        that will use the order (a still camera): SNAIL_ORDER_SORTED."""
        n = len(cams)
        cam13 = np.ascontiguousarray(np.stack([c.as_array13() for c in cams]), dtype=np.float32)
        arr = lambda xs: (C.c_void_p * n)(*[x.data_ptr() for x in xs])
        if next_order is not None:
            rc = _lib.lib().snail_trace_primary_batch_reorder_dev(self._h, n, _lib.ptr(cam13), resx, resy, arr([o.t for o in outs]), arr([o.u for o in outs]),
                                                                  arr([o.v for o in outs]), arr([o.tri_id for o in outs]), _lib.ptr(stats), _lib.ptr(order),
                                                                  _lib.ptr(slot_cost), _lib.ptr(next_order), self.ORDER_SORTED if order_exact else self.ORDER_AUTO, _stream_ptr(stream))
            _lib.check(rc, "snail_trace_primary_batch_reorder_dev")
            return outs
        rc = _lib.lib().snail_trace_primary_batch_dev(self._h, n, _lib.ptr(cam13), resx, resy, arr([o.t for o in outs]), arr([o.u for o in outs]),
                                                      arr([o.v for o in outs]), arr([o.tri_id for o in outs]), _lib.ptr(stats), _lib.ptr(order),
                                                      _lib.ptr(slot_cost), _stream_ptr(stream))
        _lib.check(rc, "snail_trace_primary_batch_dev")
        return outs

    def trace_packets_shaded_batch(self, cams, resx: int, resy: int, packet_xy, outs, stats=None, stream=None):
        """ONE launch for len(cams) frames of a packet list with the depth shading fused in; outs = one [n,256,3] uint8 tensor per frame
        (snail_trace_packets_shaded_batch_dev)."""
        n = len(cams)
        cam13 = np.ascontiguousarray(np.stack([c.as_array13() for c in cams]), dtype=np.float32)
        ptrs = (C.c_void_p * n)(*[x.data_ptr() for x in outs])
        rc = _lib.lib().snail_trace_packets_shaded_batch_dev(self._h, n, _lib.ptr(cam13), resx, resy, _lib.ptr(packet_xy), int(packet_xy.shape[0]), ptrs,
                                                             _lib.ptr(stats), _stream_ptr(stream))
        _lib.check(rc, "snail_trace_packets_shaded_batch_dev")
        return outs

    @staticmethod
    def primary_slots(w: int, h: int) -> int:
        """Dispatch slots of a w x h rect (>= its packets): the length of the order / slot_cost arrays of trace_primary."""
        return int(_lib.lib().snail_primary_slots(w, h))

    @staticmethod
    def order_from_cost(slot_cost, order=None, stream=None, exact=False):
        """Dispatch order (a permutation, int32) from the per-slot costs of a previous launch; stream-ordered.  Heaviest first when the costs are heavy-tailed
        or the caller knows them to be exact (exact=True: SNAIL_ORDER_SORTED), the built-in order otherwise (include/snail_hip.h)."""
        torch = _torch()
        order = order if order is not None else torch.empty_like(slot_cost)
        with torch.cuda.device(slot_cost.device):
            rc = _lib.lib().snail_order_from_cost_hint_dev(_lib.ptr(slot_cost), int(slot_cost.numel()), _lib.ptr(order), Scene.ORDER_SORTED if exact else Scene.ORDER_AUTO, _stream_ptr(stream))
        _lib.check(rc, "snail_order_from_cost_dev")
        return order

    def trace_packets(self, cam: Camera, resx: int, resy: int, packet_xy, out=None, stats=None, stream=None, order=None, slot_cost=None):
        """Trace an explicit packet list (int32 device tensor [n,2] of top-left pixels); results are
        packet-major [n,256] in the reference's quad order (t, u, v, tri_id).  `order` / `slot_cost`: optional int32 device tensors
        of n entries (snail_trace_packets_ordered_dev)."""
        torch = _torch()
        n = int(packet_xy.shape[0])
        if out is None:
            d = self._dev()
            out = (torch.empty((n, 256), dtype=torch.float32, device=d), torch.empty((n, 256), dtype=torch.float32, device=d),
                   torch.empty((n, 256), dtype=torch.float32, device=d), torch.empty((n, 256), dtype=torch.int32, device=d))
        cam13 = np.ascontiguousarray(cam.as_array13(), dtype=np.float32)
        if order is not None or slot_cost is not None:
            for a in (order, slot_cost):
                if a is not None and (a.numel() != n or a.dtype != torch.int32 or not a.is_contiguous()):
                    raise ValueError(f"order / slot_cost must be contiguous int32 tensors of {n} entries")
            rc = _lib.lib().snail_trace_packets_ordered_dev(self._h, _lib.ptr(cam13), resx, resy, _lib.ptr(packet_xy), n, _lib.ptr(out[0]), _lib.ptr(out[1]),
                                                            _lib.ptr(out[2]), _lib.ptr(out[3]), _lib.ptr(stats), _lib.ptr(order), _lib.ptr(slot_cost),
                                                            _stream_ptr(stream))
            _lib.check(rc, "snail_trace_packets_ordered_dev")
            return out
        rc = _lib.lib().snail_trace_packets_dev(self._h, _lib.ptr(cam13), resx, resy, _lib.ptr(packet_xy), n, _lib.ptr(out[0]), _lib.ptr(out[1]),
                                                _lib.ptr(out[2]), _lib.ptr(out[3]), _lib.ptr(stats), _stream_ptr(stream))
        _lib.check(rc, "snail_trace_packets_dev")
        return out

    def trace_packets_shaded(self, cam: Camera, resx: int, resy: int, packet_xy, out=None, stats=None, stream=None):
        """trace_packets with the gVals[1] depth shading fused into the kernel: returns packet-major [n,256,3] uint8 (B,G,R), equal
        to shade_depth(trace_packets(...)[0]) byte for byte."""
        torch = _torch()
        n = int(packet_xy.shape[0])
        if out is None:
            out = torch.empty((n, 256, 3), dtype=torch.uint8, device=self._dev())
        cam13 = np.ascontiguousarray(cam.as_array13(), dtype=np.float32)
        rc = _lib.lib().snail_trace_packets_shaded_dev(self._h, _lib.ptr(cam13), resx, resy, _lib.ptr(packet_xy), n, _lib.ptr(out), _lib.ptr(stats),
                                                       _stream_ptr(stream))
        _lib.check(rc, "snail_trace_packets_shaded_dev")
        return out

    @staticmethod
    def packets_to_frame(packet_xy, planes, frame: HitFrame, stream=None):
        resy, resx = frame.t.shape
        rc = _lib.lib().snail_packets_to_frame_dev(_lib.ptr(packet_xy), int(packet_xy.shape[0]), resx, resy, _lib.ptr(planes[0]), _lib.ptr(planes[1]),
                                                   _lib.ptr(planes[2]), _lib.ptr(planes[3]), _lib.ptr(frame.t), _lib.ptr(frame.u), _lib.ptr(frame.v),
                                                   _lib.ptr(frame.tri_id), _stream_ptr(stream))
        _lib.check(rc, "snail_packets_to_frame_dev")
        return frame

    # ---- framebuffer: gVals[1] depth shading + RGB8 store (src/scene_trace.cpp:128-137, src/render.cpp:11-17,171-198)
    @staticmethod
    def shade_depth(t_packets, out=None, stream=None, arith="ieee"):
        """packet-major distances [n,256] -> packet-major bytes [n,256,3] (B,G,R per pixel)."""
        torch = _torch()
        n = int(t_packets.shape[0])
        if out is None:
            out = torch.empty((n, 256, 3), dtype=torch.uint8, device=t_packets.device)
        code = Scene.ARITH[arith] if isinstance(arith, str) else int(arith)
        _lib.check(_lib.lib().snail_shade_depth_arith_dev(_lib.ptr(t_packets), n, _lib.ptr(out), code, _stream_ptr(stream)), "snail_shade_depth_arith_dev")
        return out

    @staticmethod
    def packets_bgr_to_frame(packet_xy, bgr_packets, frame_rgb8, stream=None):
        """scatter [n,256,3] bytes into an interleaved [resy,resx,3] uint8 frame."""
        resy, resx = int(frame_rgb8.shape[0]), int(frame_rgb8.shape[1])
        rc = _lib.lib().snail_packets_bgr_to_frame_dev(_lib.ptr(packet_xy), int(packet_xy.shape[0]), resx, resy, _lib.ptr(bgr_packets),
                                                       _lib.ptr(frame_rgb8), resx * 3, _stream_ptr(stream))
        _lib.check(rc, "snail_packets_bgr_to_frame_dev")
        return frame_rgb8

    @staticmethod
    def packets_bgr_to_frame_chunked(packet_xy, n_per_chunk: int, chunk_stride_bytes: int, bgr_base, frame_rgb8, stream=None):
        """packets_bgr_to_frame from a source cut into chunks of n_per_chunk packets, chunk_stride_bytes apart (bgr_base = a tensor whose
        data pointer is the first chunk's first packet): one launch scatters one frame's tiles of all ranks out of a multi-frame gather."""
        resy, resx = int(frame_rgb8.shape[0]), int(frame_rgb8.shape[1])
        rc = _lib.lib().snail_packets_bgr_to_frame_chunked_dev(_lib.ptr(packet_xy), int(packet_xy.shape[0]), int(n_per_chunk), int(chunk_stride_bytes), resx, resy,
                                                               _lib.ptr(bgr_base), _lib.ptr(frame_rgb8), resx * 3, _stream_ptr(stream))
        _lib.check(rc, "snail_packets_bgr_to_frame_chunked_dev")
        return frame_rgb8

    @staticmethod
    def tile_layout(tiles: np.ndarray):
        """For a list of tiles (x, y, w, h) whose packets are stored tile after tile (render.tile_packets order): the index of
        each tile's first packet and the byte offset of each tile's planes in a buffer that holds the tiles back to back
        (3*w*h bytes each -- the per-tile buffers of src/node.cpp:245-258).  Returns (first_packet int32[n], offsets int64[n], total)."""
        t = np.asarray(tiles, dtype=np.int64).reshape(-1, 4)
        npk = ((t[:, 2] + 15) // 16) * ((t[:, 3] + 15) // 16)
        first = np.concatenate([[0], np.cumsum(npk)[:-1]]).astype(np.int32)
        size = 3 * t[:, 2] * t[:, 3]
        off = np.concatenate([[0], np.cumsum(size)[:-1]]).astype(np.int64)
        return first, off, int(size.sum())

    @staticmethod
    def packets_bgr_to_planar(tiles_dev, first_packet_dev, offsets_dev, bgr_packets, out, stream=None):
        """The render node's tile wire format (src/render.cpp:140-163): per tile the planes R, G-R, B-R."""
        rc = _lib.lib().snail_packets_bgr_to_planar_dev(_lib.ptr(tiles_dev), _lib.ptr(first_packet_dev), _lib.ptr(offsets_dev), int(tiles_dev.shape[0]),
                                                        _lib.ptr(bgr_packets), _lib.ptr(out), _stream_ptr(stream))
        _lib.check(rc, "snail_packets_bgr_to_planar_dev")
        return out

    @staticmethod
    def planar_to_frame(tiles_dev, offsets_dev, planar, frame_rgb8, stream=None):
        """Inverse of packets_bgr_to_planar (src/compression.cpp:112-141) into an interleaved [resy,resx,3] uint8 frame."""
        resy, resx = int(frame_rgb8.shape[0]), int(frame_rgb8.shape[1])
        rc = _lib.lib().snail_planar_to_frame_dev(_lib.ptr(tiles_dev), _lib.ptr(offsets_dev), int(tiles_dev.shape[0]), _lib.ptr(planar),
                                                  _lib.ptr(frame_rgb8), resx * 3, resx, resy, _stream_ptr(stream))
        _lib.check(rc, "snail_planar_to_frame_dev")
        return frame_rgb8

    WHITTED_STAGES = 4
    ORDER_AUTO, ORDER_SORTED = 0, 1      # SNAIL_ORDER_* (include/snail_hip.h)

    def render_whitted(self, cam: Camera, resx: int, resy: int, lights7, ambient=(0.1, 0.1, 0.1), color=(1.0, 1.0, 1.0), out=None, stats=None,
                       stream=None, reflections: bool = False, order=None, slot_cost=None, next_order=None, order_exact=False):
        """Scene::RayTrace in the reference's simple-shading configuration (primary + one shadow packet per point light;
        reflections=True = gVals[7], one mirrored bounce shaded the same way), staged on the device; returns the interleaved
        [resy,resx,3] uint8 (B,G,R) frame.  lights7 = n x {pos, color, radius} (class Light, src/light.h:5-16); defaults =
        Scene::ambientLight / defaultMat (src/scene.cpp:6-10)."""
        torch = _torch()
        if out is None:
            out = torch.zeros((resy, resx, 3), dtype=torch.uint8, device=self._dev())
        lights = np.ascontiguousarray(lights7, dtype=np.float32).reshape(-1, 7)
        cam13 = np.ascontiguousarray(cam.as_array13(), dtype=np.float32)
        amb = np.ascontiguousarray(ambient, dtype=np.float32); col = np.ascontiguousarray(color, dtype=np.float32)
        if order is not None or slot_cost is not None:
            # dispatch-order feedback of every walking stage: int32 device tensors [WHITTED_STAGES, primary_slots(resx, resy)] (snail_render_whitted_ordered_dev)
            n = self.primary_slots(resx, resy)
            for a in (order, slot_cost):
                if a is not None and (tuple(a.shape) != (self.WHITTED_STAGES, n) or a.dtype != torch.int32 or not a.is_contiguous()):
                    raise ValueError(f"order / slot_cost must be contiguous int32 tensors of shape ({self.WHITTED_STAGES}, {n})")
            if next_order is not None:      # the next orders of all four stages derived inside this launch (may be `order` itself)
                if tuple(next_order.shape) != (self.WHITTED_STAGES, n) or next_order.dtype != torch.int32 or not next_order.is_contiguous():
                    raise ValueError(f"next_order must be a contiguous int32 tensor of shape ({self.WHITTED_STAGES}, {n})")
                rc = _lib.lib().snail_render_whitted_reorder_dev(self._h, _lib.ptr(cam13), resx, resy, _lib.ptr(lights), len(lights), _lib.ptr(amb), _lib.ptr(col),
                                                                 1 if reflections else 0, _lib.ptr(out), resx * 3, _lib.ptr(stats), _lib.ptr(order), _lib.ptr(slot_cost),
                                                                 _lib.ptr(next_order), self.ORDER_SORTED if order_exact else self.ORDER_AUTO, _stream_ptr(stream))
                _lib.check(rc, "snail_render_whitted_reorder_dev")
                return out
            rc = _lib.lib().snail_render_whitted_ordered_dev(self._h, _lib.ptr(cam13), resx, resy, _lib.ptr(lights), len(lights), _lib.ptr(amb), _lib.ptr(col),
                                                             1 if reflections else 0, _lib.ptr(out), resx * 3, _lib.ptr(stats), _lib.ptr(order), _lib.ptr(slot_cost),
                                                             _stream_ptr(stream))
            _lib.check(rc, "snail_render_whitted_ordered_dev")
            return out
        rc = _lib.lib().snail_render_whitted_dev(self._h, _lib.ptr(cam13), resx, resy, _lib.ptr(lights), len(lights), _lib.ptr(amb), _lib.ptr(col),
                                                 1 if reflections else 0, _lib.ptr(out), resx * 3, _lib.ptr(stats), _stream_ptr(stream))
        _lib.check(rc, "snail_render_whitted_dev")
        return out

    def render_whitted_packets(self, cam: Camera, resx: int, resy: int, packet_xy, lights7, ambient=(0.1, 0.1, 0.1), color=(1.0, 1.0, 1.0), out=None,
                               stats=None, stream=None, reflections: bool = False):
        """render_whitted for an explicit packet list (a rank's tiles): packet-major [n,256,3] uint8 (B,G,R)."""
        torch = _torch()
        n = int(packet_xy.shape[0])
        if out is None:
            out = torch.empty((n, 256, 3), dtype=torch.uint8, device=self._dev())
        lights = np.ascontiguousarray(lights7, dtype=np.float32).reshape(-1, 7)
        cam13 = np.ascontiguousarray(cam.as_array13(), dtype=np.float32)
        amb = np.ascontiguousarray(ambient, dtype=np.float32); col = np.ascontiguousarray(color, dtype=np.float32)
        rc = _lib.lib().snail_render_whitted_packets_dev(self._h, _lib.ptr(cam13), resx, resy, _lib.ptr(packet_xy), n, _lib.ptr(lights), len(lights),
                                                         _lib.ptr(amb), _lib.ptr(col), 1 if reflections else 0, _lib.ptr(out), _lib.ptr(stats), _stream_ptr(stream))
        _lib.check(rc, "snail_render_whitted_packets_dev")
        return out

    def trace_transparency(self, cam: Camera, resx: int, resy: int, packet_xy, t_packets, tri_id_packets, sel, lights7, ambient=(0.1, 0.1, 0.1),
                           color=(1.0, 1.0, 1.0), out=None, stats=None, stream=None):
        """Scene::TraceTransparency (src/scene_trace.cpp:620-634) for the listed primary packets: their rays continued behind the hits for
        the lanes of `sel` (uint8 [n, 64], the reference's transSel) and shaded by the nested RayTrace; returns the colours
        [n, 256, 3] float32 (`transColor`).  snail_trace_transparency_dev."""
        torch = _torch()
        n = int(packet_xy.shape[0])
        if out is None:
            out = torch.zeros((n, 256, 3), dtype=torch.float32, device=self._dev())
        lights = np.ascontiguousarray(lights7, dtype=np.float32).reshape(-1, 7)
        cam13 = np.ascontiguousarray(cam.as_array13(), dtype=np.float32)
        amb = np.ascontiguousarray(ambient, dtype=np.float32); col = np.ascontiguousarray(color, dtype=np.float32)
        rc = _lib.lib().snail_trace_transparency_dev(self._h, _lib.ptr(cam13), resx, resy, _lib.ptr(packet_xy), n, _lib.ptr(t_packets), _lib.ptr(tri_id_packets),
                                                     _lib.ptr(sel), _lib.ptr(lights), len(lights), _lib.ptr(amb), _lib.ptr(col), _lib.ptr(out), _lib.ptr(stats),
                                                     _stream_ptr(stream))
        _lib.check(rc, "snail_trace_transparency_dev")
        return out

    RENDER_REFLECTIONS, RENDER_DEPTH, RENDER_AA4 = 1, 2, 4     # include/snail_hip.h: flags of the host-pointer tile API

    def render_image_host(self, cam: Camera, resx: int, resy: int, lights7=None, flags: int = 0, ambient=(0.1, 0.1, 0.1), color=(1.0, 1.0, 1.0)):
        """snail_render_image = Render(scene, camera, image, options, threads) of src/render.h:21-23 (what a C++ host calls): the rgb8 frame
        [resy, resx, 3] (B,G,R) in host memory and the call's TreeStats.  flags: RENDER_REFLECTIONS (gVals[7]), RENDER_DEPTH (gVals[1]),
        RENDER_AA4 (gVals[9], 4x antialiasing)."""
        lights = np.ascontiguousarray(lights7 if lights7 is not None else np.zeros((0, 7)), dtype=np.float32).reshape(-1, 7)
        cam13 = np.ascontiguousarray(cam.as_array13(), dtype=np.float32)
        amb = np.ascontiguousarray(ambient, dtype=np.float32); col = np.ascontiguousarray(color, dtype=np.float32)
        img = np.zeros((resy, resx, 3), dtype=np.uint8)
        stats = np.zeros(4, dtype=np.uint64)
        rc = _lib.lib().snail_render_image(self._h, _lib.ptr(cam13), resx, resy, _lib.ptr(lights) if len(lights) else None, len(lights), _lib.ptr(amb), _lib.ptr(col),
                                           int(flags), _lib.ptr(img), resx * 3, _lib.ptr(stats))
        _lib.check(rc, "snail_render_image")
        return img, stats

    def render_tiles_host(self, cam: Camera, resx: int, resy: int, tiles, lights7=None, flags: int = 0, ambient=(0.1, 0.1, 0.1), color=(1.0, 1.0, 1.0), scenes=None):
        """snail_render_tiles (or, with `scenes` = a list of Scene objects holding the same tree, snail_render_tiles_multi): the tile-list
        renderer of src/render.h:16-19.  tiles = [n, 4] (x, y, w, h); returns (data, offsets, stats) with tile k's planes R, G-R, B-R at
        data[offsets[k]:offsets[k] + 3 w h]."""
        import ctypes as C_
        t = np.ascontiguousarray(tiles, dtype=np.int32).reshape(-1, 4)
        size = 3 * t[:, 2].astype(np.int64) * t[:, 3]
        offsets = np.concatenate([[0], np.cumsum(size)[:-1]]).astype(np.int64)
        data = np.zeros(int(size.sum()), dtype=np.uint8)
        lights = np.ascontiguousarray(lights7 if lights7 is not None else np.zeros((0, 7)), dtype=np.float32).reshape(-1, 7)
        cam13 = np.ascontiguousarray(cam.as_array13(), dtype=np.float32)
        amb = np.ascontiguousarray(ambient, dtype=np.float32); col = np.ascontiguousarray(color, dtype=np.float32)
        stats = np.zeros(4, dtype=np.uint64)
        lp = _lib.ptr(lights) if len(lights) else None
        if scenes:
            hs = (C_.c_void_p * len(scenes))(*[sc._h.value for sc in scenes])
            rc = _lib.lib().snail_render_tiles_multi(hs, len(scenes), _lib.ptr(cam13), resx, resy, _lib.ptr(t), _lib.ptr(offsets), len(t), lp, len(lights), _lib.ptr(amb),
                                                     _lib.ptr(col), int(flags), _lib.ptr(data), _lib.ptr(stats))
            _lib.check(rc, "snail_render_tiles_multi")
        else:
            rc = _lib.lib().snail_render_tiles(self._h, _lib.ptr(cam13), resx, resy, _lib.ptr(t), _lib.ptr(offsets), len(t), lp, len(lights), _lib.ptr(amb), _lib.ptr(col),
                                               int(flags), _lib.ptr(data), _lib.ptr(stats))
            _lib.check(rc, "snail_render_tiles")
        return data, offsets, stats

    def trace_primary_host(self, cam: Camera, resx: int, resy: int, rect=None):
        """Host-buffer entry point (what a C++ host would call): numpy planes in, numpy planes out."""
        x0, y0, w, h = rect if rect is not None else (0, 0, resx, resy)
        t = np.full((resy, resx), np.inf, dtype=np.float32)
        u = np.zeros((resy, resx), dtype=np.float32)
        v = np.zeros((resy, resx), dtype=np.float32)
        tid = np.zeros((resy, resx), dtype=np.int32)
        stats = np.zeros(4, dtype=np.uint64)
        cam13 = np.ascontiguousarray(cam.as_array13(), dtype=np.float32)
        rc = _lib.lib().snail_trace_primary(self._h, _lib.ptr(cam13), resx, resy, x0, y0, w, h, _lib.ptr(t), _lib.ptr(u), _lib.ptr(v), _lib.ptr(tid),
                                            _lib.ptr(stats))
        _lib.check(rc, "snail_trace_primary")
        return t, u, v, tid, stats

    def trace_frame_packets_host(self, cam: Camera, resx: int, resy: int):
        """snail_trace_frame_packets: the whole frame's primary packets with host outputs in PACKET-MAJOR order ([packets, 256], the reference's quad
        order) -- the Context arrays a host's per-packet loop reads; rays of edge packets outside the image included.  Returns (t, u, v, triId, stats)."""
        n = ((resx + 15) // 16) * ((resy + 15) // 16)
        t = np.zeros((n, 256), dtype=np.float32); u = np.zeros((n, 256), dtype=np.float32); v = np.zeros((n, 256), dtype=np.float32)
        tid = np.zeros((n, 256), dtype=np.int32)
        stats = np.zeros(4, dtype=np.uint64)
        cam13 = np.ascontiguousarray(cam.as_array13(), dtype=np.float32)
        rc = _lib.lib().snail_trace_frame_packets(self._h, _lib.ptr(cam13), resx, resy, _lib.ptr(t), _lib.ptr(u), _lib.ptr(v), _lib.ptr(tid), _lib.ptr(stats))
        _lib.check(rc, "snail_trace_frame_packets")
        return t, u, v, tid, stats

    # ---- generic / shadow packets -------------------------------------------------------------
    def traverse_primary(self, ctx: Context, stats=None, stream=None) -> Context:
        rc = _lib.lib().snail_trace_rays_dev(self._h, ctx.n_packets, ctx.size, int(ctx.shared_origin), _lib.ptr(ctx.origin), _lib.ptr(ctx.dir),
                                             _lib.ptr(ctx.idir), _lib.ptr(ctx.mask), _lib.ptr(ctx.distance), _lib.ptr(ctx.object),
                                             _lib.ptr(ctx.barycentric), _lib.ptr(stats), _stream_ptr(stream))
        _lib.check(rc, "snail_trace_rays_dev")
        return ctx

    def traverse_shadow(self, ctx: ShadowContext, stats=None, stream=None) -> ShadowContext:
        rc = _lib.lib().snail_trace_shadow_dev(self._h, ctx.n_packets, ctx.size, _lib.ptr(ctx.origin), _lib.ptr(ctx.dir), _lib.ptr(ctx.idir),
                                               _lib.ptr(ctx.distance), _lib.ptr(stats), _stream_ptr(stream))
        _lib.check(rc, "snail_trace_shadow_dev")
        return ctx

    def trace_rays_host(self, origin, dir, idir, mask, distance, obj, bary, n_packets, size, shared):
        stats = np.zeros(4, dtype=np.uint64)
        rc = _lib.lib().snail_trace_rays(self._h, n_packets, size, int(shared), _lib.ptr(origin), _lib.ptr(dir), _lib.ptr(idir), _lib.ptr(mask),
                                         _lib.ptr(distance), _lib.ptr(obj), _lib.ptr(bary), _lib.ptr(stats))
        _lib.check(rc, "snail_trace_rays")
        return stats

    def trace_shadow_host(self, origin3, dir, idir, distance, n_packets, size):
        stats = np.zeros(4, dtype=np.uint64)
        rc = _lib.lib().snail_trace_shadow(self._h, n_packets, size, _lib.ptr(origin3), _lib.ptr(dir), _lib.ptr(idir), _lib.ptr(distance),
                                           _lib.ptr(stats))
        _lib.check(rc, "snail_trace_shadow")
        return stats

    # ---- measurement --------------------------------------------------------------------------
    def account_primary(self, cam: Camera, resx: int, resy: int, rect=None) -> np.ndarray:
        """{rays, sum V_n, sum V_t, hits} of the single-ray accounting walk."""
        x0, y0, w, h = rect if rect is not None else (0, 0, resx, resy)
        out = np.zeros(4, dtype=np.uint64)
        cam13 = np.ascontiguousarray(cam.as_array13(), dtype=np.float32)
        _lib.check(_lib.lib().snail_account_primary(self._h, _lib.ptr(cam13), resx, resy, x0, y0, w, h, _lib.ptr(out)), "snail_account_primary")
        return out

    def packet_costs(self, cam: Camera, resx: int, resy: int) -> np.ndarray:
        """Per-packet diagnostics of one full-frame primary launch (snail_account_packets; the packet code built with counters,
        a measurement pass, never a product path): uint32 [packets, 8] = {node visits, quad x triangle tests, shader cycles, start >> 6, triangle records
        fetched, leaf bodies, 0, 0}, row-major over the packet grid."""
        np_ = ((resx + 15) // 16) * ((resy + 15) // 16)
        out = np.zeros((np_, 8), dtype=np.uint32)
        cam13 = np.ascontiguousarray(cam.as_array13(), dtype=np.float32)
        _lib.check(_lib.lib().snail_account_packets(self._h, _lib.ptr(cam13), resx, resy, _lib.ptr(out)), "snail_account_packets")
        return out

    # ---- arithmetic of Inv / RSqrt / FastInv (include/snail_hip.h, "arithmetic") ------------------------
    ARITH = {"ieee": 0, "host_sse": 1}

    def set_arith(self, arith) -> None:
        """"ieee" (default: veclib's scalar definitions) or "host_sse" (veclib's SSE definitions as THIS host's CPU executes them:
        rcpps / rsqrtps reproduced on the device from the CPU's tables + the Newton steps).  Raises SnailError when the host's
        instructions cannot be reproduced from tables."""
        code = self.ARITH[arith] if isinstance(arith, str) else int(arith)
        _lib.check(_lib.lib().snail_scene_set_arith(self._h, code), "snail_scene_set_arith")

    def arith(self) -> str:
        a = C.c_int(0)
        _lib.check(_lib.lib().snail_scene_arith(self._h, C.addressof(a)), "snail_scene_arith")
        return {v: k for k, v in self.ARITH.items()}[a.value]

    def flags(self):
        """(fastOK, nestedOK) as snail_scene_create found them (snail_scene_flags)."""
        f, n = C.c_int(0), C.c_int(0)
        _lib.check(_lib.lib().snail_scene_flags(self._h, C.addressof(f), C.addressof(n)), "snail_scene_flags")
        return bool(f.value), bool(n.value)

    def last_launch(self):
        b, t = C.c_int(0), C.c_int(0)
        _lib.check(_lib.lib().snail_last_launch(self._h, C.addressof(b), C.addressof(t)), "snail_last_launch")
        return b.value, t.value
