"""Wire and disk formats at the edges of the path, behind include/rgc_hip.h:

* sensor_msgs/PointCloud2 bytes <-> arrays (pcl::fromROSMsg at scanRegistration.cpp:107-108, pcl::toROSMsg at :689-727) --
  unpacked / packed by a kernel on the MI355X (needs the device);
* the TUM trajectory line of RGC_odometer.cpp:1315-1316 and the key-frame .pcd file of :1353-1354 (host only)."""
import ctypes as C

import numpy as np

from . import _lib

INT8, UINT8, INT16, UINT16, INT32, UINT32, FLOAT32, FLOAT64 = 1, 2, 3, 4, 5, 6, 7, 8   # sensor_msgs/PointField
FIELDS = ("x", "y", "z", "intensity", "ring", "time")


def layout(point_step, fields, is_bigendian=False, strict=False):
    """fields: {name: (offset, datatype)} for any of x, y, z, intensity, ring, time (the message's PointField table)"""
    L = _lib.Pc2Layout()
    L.point_step = int(point_step)
    for i, name in enumerate(FIELDS):
        off, ty = fields.get(name, (-1, 0))
        L.offset[i], L.datatype[i] = int(off), int(ty)
    L.is_bigendian, L.strict = int(bool(is_bigendian)), int(bool(strict))
    return L


class Wire:
    """PointCloud2 (un)packing on the device."""

    def __init__(self, device: int = 0):
        self._L = _lib.load()
        h = C.c_void_p()
        rc = self._L.rgc_create(device, None, C.byref(h))
        if rc:
            raise _lib.RgcError(rc, self._L.rgc_status_string(rc).decode())
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
            raise _lib.RgcError(rc, self._L.rgc_last_error(self._h).decode() or self._L.rgc_status_string(rc).decode())

    def unpack(self, data: bytes, n_points: int, lay, want_ring=False, want_time=False):
        """-> (xyzi (n,4) float32, ring (n,) int32 or None, time (n,) float32 or None)"""
        buf = np.frombuffer(data, dtype=np.uint8)
        if buf.size < n_points * lay.point_step:
            raise ValueError("data shorter than n_points * point_step")
        xyzi = np.empty((n_points, 4), np.float32)
        ring = np.empty(n_points, np.int32) if want_ring else None
        tm = np.empty(n_points, np.float32) if want_time else None
        self._chk(self._L.rgc_pc2_unpack(self._h, buf.ctypes.data, n_points, C.byref(lay), xyzi.ctypes.data,
                                         ring.ctypes.data if want_ring else None, tm.ctypes.data if want_time else None, 0))
        return xyzi, ring, tm

    def pack(self, pts, kind="xyzi") -> bytes:
        """kind 'xyzi': pts (n,4) -> PointXYZI message bytes (32 per point); 'xyzinormal': pts (n,5) -> PointXYZINormal (48)"""
        k = 0 if kind == "xyzi" else 1
        a = np.ascontiguousarray(pts, dtype=np.float32)
        if a.ndim != 2 or a.shape[1] != (4 if k == 0 else 5):
            raise ValueError("expected (n,4) for xyzi, (n,5) for xyzinormal")
        out = np.empty(a.shape[0] * (32 if k == 0 else 48), np.uint8)
        self._chk(self._L.rgc_pc2_pack(self._h, k, a.ctypes.data, a.shape[0], 0, out.ctypes.data))
        return out.tobytes()


def point_fields(kind="xyzi"):
    """the PointField table pcl::toROSMsg emits: ([(name, offset, datatype, count)], point_step)"""
    L = _lib.load()
    arr = (_lib.Pc2Field * 8)()
    step = C.c_int(0)
    n = L.rgc_pc2_point_fields(0 if kind == "xyzi" else 1, arr, 8, C.byref(step))
    if n < 0:
        raise _lib.RgcError(n, "rgc_pc2_point_fields")
    return [(arr[i].name.decode(), arr[i].offset, arr[i].datatype, arr[i].count) for i in range(n)], step.value


def tum_line(stamp, t, q_xyzw) -> str:
    L = _lib.load()
    buf = C.create_string_buffer(256)
    tt = (C.c_double * 3)(*[float(v) for v in t])
    qq = (C.c_double * 4)(*[float(v) for v in q_xyzw])
    n = L.rgc_tum_line(float(stamp), tt, qq, buf, 256)
    if n < 0:
        raise _lib.RgcError(n, "rgc_tum_line")
    return buf.value.decode()


def pcd_write(path, xyzi, binary=True):
    L = _lib.load()
    a = np.ascontiguousarray(xyzi, dtype=np.float32)
    rc = L.rgc_pcd_write(str(path).encode(), a.ctypes.data_as(C.POINTER(C.c_float)), a.shape[0], 1 if binary else 0)
    if rc:
        raise _lib.RgcError(rc, "rgc_pcd_write")
