"""snail_amd -- MI355X-native drop-in for the hot path of nadult/Snail: packetised SAH-BVH traversal +
ray/triangle intersection.  The product path is libsnailhip.so (hand-written gfx950 HIP
behind the C-ABI of include/snail_hip.h); this package is the host-side mirror of the reference's
operator interface for that path."""
from ._lib import SnailError, lib, LIB_PATH  # noqa: F401
from .bvh import HostBVH, TRI_DTYPE, NODE_DTYPE  # noqa: F401
from .camera import Camera, FPSCamera, survey_camera  # noqa: F401

__all__ = ["SnailError", "lib", "LIB_PATH", "HostBVH", "TRI_DTYPE", "NODE_DTYPE", "Camera", "FPSCamera", "survey_camera"]
