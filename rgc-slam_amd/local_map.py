"""f2: the odometer's local map kept resident on the device -- host-side mirror of the rgc_map_* entry points of
include/rgc_hip.h.  Replaces the keyframe deque and its per-frame re-framing / re-filtering / re-upload
(/root/reference/rgc_slam/src/RGC_odometer.cpp:1218-1256, 985-991, 1007).  Nothing is computed on the CPU."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import RgcError

_dp = C.POINTER(C.c_double)


class RollingLocalMap:
    """Lives in the context of a registration object (``FastVGICP``): ``commit`` makes the map that object's target."""

    def __init__(self, registration):
        self._reg = registration
        self._L = registration._L
        self._h = registration._h

    def _chk(self, rc):
        if rc != 0:
            raise RgcError(rc, self._L.rgc_last_error(self._h).decode() or self._L.rgc_status_string(rc).decode())

    def reset(self, origin=None):
        o = None if origin is None else np.ascontiguousarray(origin, np.float64)
        self._chk(self._L.rgc_map_reset(self._h, o.ctypes.data_as(_dp) if o is not None else None))

    def insert(self, xyzi, q_w_xyzw, t_w) -> int:
        """surroundingCloud.push_back(transformPointCloud(cloud, q_w_curr, t_w_curr)) (:1237); returns the keyframe id"""
        a = np.ascontiguousarray(xyzi, dtype=np.float32)
        if a.ndim != 2 or a.shape[1] < 4:
            raise RgcError(_lib.ERR_INVALID, "a keyframe is (n, >=4) float32: x, y, z, intensity")
        q, t = np.ascontiguousarray(q_w_xyzw, np.float64), np.ascontiguousarray(t_w, np.float64)
        kid = C.c_int(-1)
        self._chk(self._L.rgc_map_insert(self._h, a.ctypes.data, a.shape[0], a.strides[0], q.ctypes.data_as(_dp), t.ctypes.data_as(_dp), 0, C.byref(kid)))
        return kid.value

    def evict(self, max_keyframes=0, center=None, radius=0.0) -> int:
        c = None if center is None else np.ascontiguousarray(center, np.float64)
        n = C.c_int(0)
        self._chk(self._L.rgc_map_evict(self._h, int(max_keyframes), c.ctypes.data_as(_dp) if c is not None else None, float(radius), C.byref(n)))
        return n.value

    def rebase(self, new_origin):
        o = np.ascontiguousarray(new_origin, np.float64)
        self._chk(self._L.rgc_map_rebase(self._h, o.ctypes.data_as(_dp)))

    def commit(self, leaf) -> int:
        """VoxelGrid(leaf) of the keyframes + setInputTarget, on the device; a no-op when nothing changed"""
        n = C.c_int(0)
        self._chk(self._L.rgc_map_commit(self._h, float(leaf), C.byref(n)))
        self._reg._n_tgt = n.value
        self._reg._fitness = None
        return n.value

    def info(self) -> dict:
        i = _lib.MapInfo()
        self._chk(self._L.rgc_map_get_info(self._h, C.byref(i)))
        return dict(n_keyframes=i.n_keyframes, n_points=i.n_points, n_target=i.n_target, revision=i.revision, oldest_id=i.oldest_id,
                    newest_id=i.newest_id, origin=np.array(list(i.origin)))

    def _download(self, which):
        n = C.c_int(0)
        self._chk(self._L.rgc_map_download(self._h, which, None, 0, C.byref(n)))
        out = np.empty((n.value, 4), np.float32)
        if n.value:
            self._chk(self._L.rgc_map_download(self._h, which, out.ctypes.data, n.value, C.byref(n)))
        return out

    def points(self):
        """the stored keyframe points (map frame), insertion order"""
        return self._download(0)

    def target(self):
        """the committed target cloud"""
        return self._download(1)
