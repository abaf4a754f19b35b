"""Host-side mirror of the per-point stages around the registration operator in the odometer's frame body
(/root/reference/rgc_slam/src/RGC_odometer.cpp, vg_ICP::ICP_thread), backed by the HIP library through the C-ABI.
Same names as the reference functions; nothing is computed on the CPU."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import RgcError


class Preprocessor:
    """adjustDistortion (B2), pcl::VoxelGrid (B3) and transformPointCloud (B9) on the GPU."""

    def __init__(self, device: int = 0):
        self._L = _lib.load()
        h = C.c_void_p()
        rc = self._L.rgc_create(device, None, C.byref(h))
        if rc != 0:
            raise RgcError(rc, self._L.rgc_status_string(rc).decode())
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self._L.rgc_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            raise RgcError(rc, self._L.rgc_last_error(self._h).decode())

    def adjustDistortion(self, xyzi, q_last_curr_xyzw, t_last_curr):
        """RGC_odometer.cpp:1441-1481; returns the de-skewed copy (n,4)."""
        a = np.array(xyzi, dtype=np.float32, order="C", copy=True)
        q = np.ascontiguousarray(q_last_curr_xyzw, dtype=np.float64)
        t = np.ascontiguousarray(t_last_curr, dtype=np.float64)
        dp = C.POINTER(C.c_double)
        self._chk(self._L.rgc_deskew(self._h, a.ctypes.data, a.shape[0], a.strides[0], q.ctypes.data_as(dp), t.ctypes.data_as(dp), 0))
        return a

    def voxelGridFilter(self, xyzi, leaf):
        """pcl::VoxelGrid<PointXYZI> setLeafSize(leaf,leaf,leaf) + filter (RGC_odometer.cpp:976-991)."""
        a = np.ascontiguousarray(xyzi, dtype=np.float32)
        out = np.empty((a.shape[0], 4), np.float32)
        n = C.c_int(0)
        self._chk(self._L.rgc_voxelgrid(self._h, a.ctypes.data, a.shape[0], a.strides[0], float(leaf), out.ctypes.data, C.byref(n), 0))
        return out[:n.value].copy()

    def transformPointCloud(self, xyzi, q_xyzw, t):
        """RGC_odometer.cpp:1495-1514."""
        a = np.ascontiguousarray(xyzi, dtype=np.float32)
        q = np.ascontiguousarray(q_xyzw, dtype=np.float64)
        tt = np.ascontiguousarray(t, dtype=np.float64)
        out = np.empty((a.shape[0], 4), np.float32)
        dp = C.POINTER(C.c_double)
        self._chk(self._L.rgc_transform_cloud(self._h, a.ctypes.data, a.shape[0], a.strides[0], q.ctypes.data_as(dp), tt.ctypes.data_as(dp),
                                              out.ctypes.data, 0))
        return out
