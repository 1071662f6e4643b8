"""Camera types mirroring src/camera.h:7-14 (Camera) and src/camera.cpp:3-45 (FPSCamera -> Camera).

All arithmetic is float32, one rounding per operation, in the reference's order, because the camera
vectors feed the primary-ray generator whose output must be bit-identical between oracle and GPU."""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np

F32 = np.float32


def _f(x) -> np.float32:
    return np.float32(x)


@dataclass
class Camera:
    """`struct Camera { float plane_dist; Vec3f pos, right, up, front; }` (src/camera.h:7-14)."""
    pos: np.ndarray
    right: np.ndarray
    up: np.ndarray
    front: np.ndarray
    plane_dist: float = 1.0

    def __setattr__(self, name, value):
        if name in ("pos", "right", "up", "front"):
            # own read-only copies: the flat form below is cached, so an in-place edit of a field array (cam.pos[0] = x, or of an array the
            # caller still holds) must fail loudly instead of leaving the cache stale; assign a new array to move the camera
            value = np.array(value, dtype=F32).reshape(3)
            value.setflags(write=False)
        object.__setattr__(self, name, value)
        if name != "_a13":
            object.__setattr__(self, "_a13", None)     # a changed field invalidates the cached flat form

    def as_array13(self) -> np.ndarray:
        """pos, right, up, front, plane_dist -- the flat layout the C-ABI takes (cached, read-only: a frame loop asks for it several times per frame)."""
        a = self.__dict__.get("_a13")
        if a is None:
            a = np.ascontiguousarray(np.concatenate([
                np.asarray(self.pos, dtype=F32), np.asarray(self.right, dtype=F32),
                np.asarray(self.up, dtype=F32), np.asarray(self.front, dtype=F32),
                np.asarray([self.plane_dist], dtype=F32)]).astype(F32))
            a.setflags(write=False)
            object.__setattr__(self, "_a13", a)
        return a


def _rotate(axis, radians):
    """static Rotate() of src/camera.cpp:3-21."""
    c = _f(math.cos(float(_f(radians))))
    s = _f(math.sin(float(_f(radians))))
    omc = _f(_f(1.0) - c)
    ax = [_f(a) for a in axis]
    xx, yy, zz = _f(ax[0] * ax[0]), _f(ax[1] * ax[1]), _f(ax[2] * ax[2])
    xym = _f(_f(ax[0] * ax[1]) * omc)
    xzm = _f(_f(ax[0] * ax[2]) * omc)
    yzm = _f(_f(ax[1] * ax[2]) * omc)
    xs, ys, zs = _f(ax[0] * s), _f(ax[1] * s), _f(ax[2] * s)
    v1 = np.array([_f(_f(xx * omc) + c), _f(xym + zs), _f(xzm - ys)], dtype=F32)
    v2 = np.array([_f(xym - zs), _f(_f(yy * omc) + c), _f(yzm + xs)], dtype=F32)
    v3 = np.array([_f(xzm + ys), _f(yzm - xs), _f(_f(zz * omc) + c)], dtype=F32)
    return v1, v2, v3


def _dot(a, b) -> np.float32:
    return _f(_f(_f(a[0] * b[0]) + _f(a[1] * b[1])) + _f(a[2] * b[2]))


@dataclass
class FPSCamera:
    """src/camera.h:30-51; conversion operator src/camera.cpp:31-45."""
    pos: np.ndarray = field(default_factory=lambda: np.zeros(3, dtype=F32))
    ang: float = 0.0
    pitch: float = 0.0
    plane_dist: float = 1.0

    def camera(self) -> Camera:
        x0, y0, z0 = _rotate((0, 1, 0), self.ang)
        x1, y1, z1 = _rotate((1, 0, 0), self.pitch)
        right = np.array([_dot(x1, x0), _dot(x1, y0), _dot(x1, z0)], dtype=F32)
        up = np.array([_dot(y1, x0), _dot(y1, y0), _dot(y1, z0)], dtype=F32)
        front = np.array([_dot(z1, x0), _dot(z1, y0), _dot(z1, z0)], dtype=F32)
        return Camera(np.asarray(self.pos, dtype=F32), right, up, front, self.plane_dist)


def survey_camera(tri_verts: np.ndarray) -> Camera:
    """The far camera SURVEY.md section 8(c) used for its recorded digests: FPSCamera with ang = pitch = 0
    at  centre - (0, 0, 1.2*size.z + 0.5*max(size.x, size.y))  of the scene bbox, looking +z."""
    tv = np.asarray(tri_verts, dtype=F32).reshape(-1, 3)
    mn, mx = tv.min(axis=0), tv.max(axis=0)
    size = (mx - mn).astype(F32)
    centre = ((mx + mn).astype(F32) * _f(0.5)).astype(F32)
    back = _f(_f(_f(1.2) * size[2]) + _f(_f(0.5) * max(size[0], size[1])))
    pos = np.array([centre[0], centre[1], _f(centre[2] - back)], dtype=F32)
    return FPSCamera(pos, 0.0, 0.0).camera()
