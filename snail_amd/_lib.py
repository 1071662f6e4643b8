"""ctypes binding of libsnailhip.so (the C-ABI of include/snail_hip.h).

There is NO fallback: if the HIP library is missing or a call fails, SnailError is raised.  Nothing in
this package imports or calls the CPU oracle under oracle/ (that is test infrastructure)."""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# SNAIL_LIB_PATH: load another build of the same C-ABI (an experiment of tools/variant.sh, or the workbench build libsnailhip_debug.so
# when a whole process should run under its SNAIL_DEBUG_* environment switches) without touching the product library
LIB_PATH = os.environ.get("SNAIL_LIB_PATH") or os.path.join(HERE, "libsnailhip.so")


class SnailError(RuntimeError):
    pass


_lib = None

# name -> (restype, argtypes); kept in one table so tests can check every symbol of include/snail_hip.h
_VP, _I, _F13 = C.c_void_p, C.c_int, C.c_void_p
SIGNATURES = {
    "snail_last_error": (C.c_char_p, []),
    "snail_device_count": (_I, []),
    "snail_tris_from_verts": (_I, [_VP, _I, _VP]),
    "snail_bvh_build": (_I, [_VP, _I, _VP, _VP, _VP, _VP]),
    "snail_scene_create": (_VP, [_VP, _I, _VP, _I, _I, _I]),
    "snail_scene_destroy": (None, [_VP]),
    "snail_scene_info": (_I, [_VP, _VP, _VP, _VP, _VP]),
    "snail_scene_flags": (_I, [_VP, _VP, _VP]),
    "snail_scene_set_arith": (_I, [_VP, _I]),
    "snail_scene_arith": (_I, [_VP, _VP]),
    "snail_host_sse_tables": (_I, [_VP]),
    "snail_arith_set_tables": (_I, [_VP]),
    "snail_arith_prepare_device": (_I, []),
    "snail_host_sse_check": (_I, [_I, C.c_uint64, C.c_uint64, _I, _VP, _VP]),
    "snail_scene_create_lbvh": (_VP, [_VP, _I, _I, _I, _VP, _VP]),
    "snail_scene_download": (_I, [_VP, _VP, _VP]),
    "snail_trace_primary": (_I, [_VP, _F13, _I, _I, _I, _I, _I, _I, _VP, _VP, _VP, _VP, _VP]),
    "snail_trace_frame_packets": (_I, [_VP, _F13, _I, _I, _VP, _VP, _VP, _VP, _VP]),
    "snail_trace_primary_dev": (_I, [_VP, _F13, _I, _I, _I, _I, _I, _I, _VP, _VP, _VP, _VP, _VP, _VP]),
    "snail_trace_packets_dev": (_I, [_VP, _F13, _I, _I, _VP, _I, _VP, _VP, _VP, _VP, _VP, _VP]),
    "snail_trace_packets_shaded_dev": (_I, [_VP, _F13, _I, _I, _VP, _I, _VP, _VP, _VP]),
    "snail_trace_primary_batch_dev": (_I, [_VP, _I, _VP, _I, _I, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP]),
    "snail_trace_packets_shaded_batch_dev": (_I, [_VP, _I, _VP, _I, _I, _VP, _I, _VP, _VP, _VP]),
    "snail_trace_primary_batch_reorder_dev": (_I, [_VP, _I, _VP, _I, _I, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _I, _VP]),
    "snail_render_whitted_reorder_dev": (_I, [_VP, _F13, _I, _I, _VP, _I, _VP, _VP, _I, _VP, _I, _VP, _VP, _VP, _VP, _I, _VP]),
    "snail_primary_slots": (_I, [_I, _I]),
    "snail_trace_primary_ordered_dev": (_I, [_VP, _F13, _I, _I, _I, _I, _I, _I, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP]),
    "snail_trace_packets_ordered_dev": (_I, [_VP, _F13, _I, _I, _VP, _I, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP]),
    "snail_order_from_cost_dev": (_I, [_VP, _I, _VP, _VP]),
    "snail_order_from_cost_hint_dev": (_I, [_VP, _I, _VP, _I, _VP]),
    "snail_packets_to_frame_dev": (_I, [_VP, _I, _I, _I, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP]),
    "snail_trace_rays": (_I, [_VP, _I, _I, _I, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP]),
    "snail_trace_rays_dev": (_I, [_VP, _I, _I, _I, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP]),
    "snail_trace_shadow": (_I, [_VP, _I, _I, _VP, _VP, _VP, _VP, _VP]),
    "snail_trace_shadow_dev": (_I, [_VP, _I, _I, _VP, _VP, _VP, _VP, _VP, _VP]),
    "snail_shade_depth_dev": (_I, [_VP, _I, _VP, _VP]),
    "snail_shade_depth_arith_dev": (_I, [_VP, _I, _VP, _I, _VP]),
    "snail_packets_bgr_to_frame_dev": (_I, [_VP, _I, _I, _I, _VP, _VP, _I, _VP]),
    "snail_packets_bgr_to_frame_chunked_dev": (_I, [_VP, _I, _I, C.c_int64, _I, _I, _VP, _VP, _I, _VP]),
    "snail_packets_bgr_to_planar_dev": (_I, [_VP, _VP, _VP, _I, _VP, _VP, _VP]),
    "snail_planar_to_frame_dev": (_I, [_VP, _VP, _I, _VP, _VP, _I, _I, _I, _VP]),
    "snail_render_whitted_dev": (_I, [_VP, _F13, _I, _I, _VP, _I, _VP, _VP, _I, _VP, _I, _VP, _VP]),
    "snail_render_whitted_ordered_dev": (_I, [_VP, _F13, _I, _I, _VP, _I, _VP, _VP, _I, _VP, _I, _VP, _VP, _VP, _VP]),
    "snail_render_whitted_packets_dev": (_I, [_VP, _F13, _I, _I, _VP, _I, _VP, _I, _VP, _VP, _I, _VP, _VP, _VP]),
    "snail_trace_transparency_dev": (_I, [_VP, _F13, _I, _I, _VP, _I, _VP, _VP, _VP, _VP, _I, _VP, _VP, _VP, _VP, _VP]),
    "snail_render_tiles": (_I, [_VP, _F13, _I, _I, _VP, _VP, _I, _VP, _I, _VP, _VP, _I, _VP, _VP]),
    "snail_render_tiles_multi": (_I, [_VP, _I, _F13, _I, _I, _VP, _VP, _I, _VP, _I, _VP, _VP, _I, _VP, _VP]),
    "snail_render_image": (_I, [_VP, _F13, _I, _I, _VP, _I, _VP, _VP, _I, _VP, _I, _VP]),
    "snail_account_primary": (_I, [_VP, _F13, _I, _I, _I, _I, _I, _I, _VP]),
    "snail_account_packets": (_I, [_VP, _F13, _I, _I, _VP]),
    "snail_last_launch": (_I, [_VP, _VP, _VP]),
}

# include/snail_hip_debug.h: the workbench build only (libsnailhip_debug.so, -DSNAIL_DEBUG_API)
DEBUG_SIGNATURES = {
    "snail_debug_delay_dev": (_I, [C.c_float, _VP]),
    "snail_debug_clock_dev": (_I, [C.c_float, _VP, _VP]),
    "snail_debug_recip_check": (_I, [_VP]),
    "snail_debug_dispatch_rate": (_I, [_I, _I, _I, _VP]),
    "snail_debug_hostsse_device_check": (_I, [_I, _I, _VP, _VP]),
    "snail_debug_occupancy": (_I, [_VP]),
    "snail_debug_anyorder": (_I, [_VP, _F13, _I, _I, _I, _I, _VP]),
}
DEBUG_LIB_PATH = os.path.join(HERE, "libsnailhip_debug.so")
_dbg = None


def debug_lib():
    """The workbench build of the same sources: the product C-ABI + include/snail_hip_debug.h + the SNAIL_DEBUG_* environment switches.
    For tests and tools only; nothing on a product path loads it."""
    global _dbg
    if _dbg is None:
        if not os.path.exists(DEBUG_LIB_PATH):
            raise SnailError("workbench library %s is missing: `make -C snail_amd/csrc debug`" % DEBUG_LIB_PATH)
        L = C.CDLL(DEBUG_LIB_PATH)
        for table in (SIGNATURES, DEBUG_SIGNATURES):
            for name, (res, args) in table.items():
                fn = getattr(L, name)
                fn.restype = res
                fn.argtypes = args
        _dbg = L
    return _dbg


def lib():
    """Load libsnailhip.so or fail loudly."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SnailError(
                "HIP extension %s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C snail_amd/csrc`). There is no CPU fallback." % LIB_PATH)
        try:
            L = C.CDLL(LIB_PATH)
        except OSError as e:  # pragma: no cover - depends on the box
            raise SnailError("cannot load %s: %s" % (LIB_PATH, e)) from e
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        for name, (res, args) in DEBUG_SIGNATURES.items():     # present only when SNAIL_LIB_PATH names the workbench build
            fn = getattr(L, name, None)
            if fn is not None:
                fn.restype = res
                fn.argtypes = args
        _lib = L
    return _lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = lib().snail_last_error()
        raise SnailError("%s failed (status %d): %s" % (what, rc, msg.decode() if msg else "?"))


def ptr(a):
    """Device/host address of a numpy array, torch tensor, or None."""
    if a is None:
        return None
    if hasattr(a, "data_ptr"):
        return C.c_void_p(a.data_ptr())
    return a.ctypes.data_as(C.c_void_p)
