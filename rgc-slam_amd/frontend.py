"""Host-side mirror of the scan feature + ground front-end node (``ScanRegistration``,
/root/reference/rgc_slam/src/scanRegistration.cpp), backed by the HIP library through the C-ABI.  The method is named
after the ROS callback it replaces; its return value holds what the node publishes (:689-727)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import RgcError


class ScanRegistration:
    def __init__(self, scan_line: int = 16, minimum_range: float = 0.5, maxmum_range: float = 80.0, USE_intensity: int = 1, device: int = 0):
        self._L = _lib.load()
        h = C.c_void_p()
        rc = self._L.rgc_create(device, None, C.byref(h))
        if rc != 0:
            raise RgcError(rc, self._L.rgc_status_string(rc).decode())
        self._h = h
        self.params = _lib.FeParams(scan_line, minimum_range, maxmum_range, USE_intensity)

    def close(self):
        if getattr(self, "_h", None):
            self._L.rgc_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def laserCloudHandlerMsg(self, data: bytes, n_points: int, layout, diagnostics: bool = True) -> dict:
        """The callback on the raw sensor_msgs/PointCloud2 bytes (scanRegistration.cpp:89-730 including fromROSMsg at :107-108):
        the message is unpacked by a kernel and the sweep never exists on the host.  layout: rgc_slam_amd.wire.layout(...)."""
        buf = np.frombuffer(data, dtype=np.uint8)
        if buf.size < n_points * layout.point_step:
            raise RgcError(_lib.ERR_INVALID, "data shorter than n_points * point_step")
        d = C.c_void_p()
        rc = self._L.rgc_device_alloc(self._h, max(n_points, 1) * 16, C.byref(d))
        if rc != 0:
            raise RgcError(rc, self._L.rgc_last_error(self._h).decode())
        try:
            rc = self._L.rgc_pc2_unpack(self._h, buf.ctypes.data, n_points, C.byref(layout), d, None, None, 1)
            if rc != 0:
                raise RgcError(rc, self._L.rgc_last_error(self._h).decode())
            return self._run(d, n_points, 16, diagnostics, on_device=True)
        finally:
            self._L.rgc_device_free(self._h, d)

    def laserCloudHandler(self, xyzi, diagnostics: bool = True, cloud: bool = True) -> dict:
        """scanRegistration.cpp:89-730.  xyzi: raw cloud (n, >=4) float32 in firing order.  cloud=False: the ring-major sweep stays on the
        device (rgc_frontend_cloud_device), as in the chained frame body -- the library then sizes the sweep's launches from the previous one."""
        a = np.ascontiguousarray(xyzi, dtype=np.float32)
        if a.ndim != 2 or a.shape[1] < 4:
            raise RgcError(_lib.ERR_INVALID, "cloud must be (n, >=4) float32: x, y, z, intensity")
        return self._run(a.ctypes.data, a.shape[0], a.strides[0], diagnostics, on_device=False, keepalive=a, want_cloud=cloud)

    def _run(self, ptr, n, stride, diagnostics, on_device, keepalive=None, want_cloud=True) -> dict:
        ns = self.params.n_scans
        fcap, gcap = ns * 6 * 41, max(10 * n, 1)
        f32, i32 = C.POINTER(C.c_float), C.POINTER(C.c_int)
        bufs = dict(cloud=np.zeros((max(n, 1), 4), np.float32), sharp=np.zeros((fcap, 5), np.float32), flat=np.zeros((fcap, 5), np.float32),
                    inten=np.zeros((fcap, 5), np.float32), ground_pts=np.zeros((gcap, 4), np.float32))
        diag = {}
        if diagnostics:
            diag = dict(curvature=np.zeros(max(n, 1), np.float32), curvature2=np.zeros(max(n, 1), np.float32),
                        inten_curvature=np.zeros(max(n, 1), np.float32), label=np.zeros(max(n, 1), np.int32),
                        inten_label=np.zeros(max(n, 1), np.int32), picked=np.zeros(max(n, 1), np.int32), ground_marked=np.zeros(max(n, 1), np.int32))
        o = _lib.FeOut()
        for k, v in {**bufs, **diag}.items():
            setattr(o, k, v.ctypes.data_as(i32 if v.dtype == np.int32 else f32))
        o.cloud_cap, o.feat_cap, o.ground_cap = max(n, 1), fcap, gcap
        if not want_cloud:
            o.cloud = None
        fn = self._L.rgc_frontend_device if on_device else self._L.rgc_frontend
        rc = fn(self._h, ptr, n, stride, C.byref(self.params), C.byref(o))
        if rc != 0:
            raise RgcError(rc, self._L.rgc_last_error(self._h).decode())
        m = o.n_cloud
        out = dict(cloud=bufs["cloud"][:m].copy(), sharp=bufs["sharp"][:o.n_sharp].copy(), flat=bufs["flat"][:o.n_flat].copy(),
                   inten=bufs["inten"][:o.n_inten].copy(), ground_pts=bufs["ground_pts"][:min(o.n_ground, gcap)].copy(), n_cloud=m,
                   n_sharp_own=o.n_sharp_own, n_ground=o.n_ground, ring_count=np.array(o.ring_count[:ns]),
                   groundparam=np.array(o.groundparam[:]), ground_valid=bool(o.ground_valid))
        for k, v in diag.items():
            out[k] = v[:m].copy()
        return out
