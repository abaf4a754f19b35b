"""Host-side mirror of the mapping node's scan-to-map FEATURE registration: the optimisation block of
laserMapping's frame loop, src/RGC_mapping.cpp:1069-1358, behind the C-ABI of include/rgc_hip.h (rgc_mapreg_*).

    reg = MapFeatureRegistration(device=0)
    reg.setInputMaps(laserCloudCornerFromMapDS, laserCloudSurfFromMapDS)         # :1073-1074
    q_w_curr, t_w_curr, q_w_last, t_w_last, report = reg.optimize(               # :1076-1341, :1375-1376
        laserCloudCornerDS, laserCloudSurfDS, laserCloudCornerLastDS, laserCloudSurfLastDS, q_w_curr, t_w_curr, q_w_last, t_w_last)

Features are (n,4) float32 {x, y, z, weight} -- the x, y, z, normal_x of the reference's PointXYZINormal -- and quaternions
are x, y, z, w like Eigen's coeffs().  No CPU fallback: without librgc_hip.so / an MI355X this raises."""
import ctypes as C

import numpy as np

from . import _lib


def _f32(a, cols=None):
    a = np.ascontiguousarray(a, dtype=np.float32)
    if cols is not None and (a.ndim != 2 or a.shape[1] != cols):
        raise ValueError(f"expected an (n,{cols}) array")
    return a, a.ctypes.data_as(C.POINTER(C.c_float))


class MapFeatureRegistration:
    def __init__(self, device: int = 0):
        self._L = _lib.load()
        h = C.c_void_p()
        rc = self._L.rgc_create(device, None, C.byref(h))
        if rc:
            raise _lib.RgcError(rc, self._L.rgc_status_string(rc).decode())
        self._h = h
        self._ready = False

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

    def setInputMaps(self, corner_map, surf_map):
        """kdtreeCornerFromMap / kdtreeSurfFromMap ->setInputCloud (:1073-1074); (n,3) or (n,4) float32"""
        cm, cp = _f32(corner_map)
        sm, sp = _f32(surf_map)
        if cm.shape[1] != sm.shape[1]:
            raise ValueError("both maps must have the same point layout")
        self._chk(self._L.rgc_mapreg_set_maps(self._h, cp, cm.shape[0], sp, sm.shape[0], 4 * cm.shape[1]))
        self._ready = True

    def associate(self, feat, q_xyzw, t, kind):
        """One association loop (:1092-1138 for kind='edge', :1191-1236 for 'plane'): dict of per-feature factor parameters."""
        f, fp = _f32(feat, 4)
        q = np.ascontiguousarray(q_xyzw, np.float64)
        tt = np.ascontiguousarray(t, np.float64)
        out = np.zeros((f.shape[0], 8))
        nv = C.c_int(0)
        dp = C.POINTER(C.c_double)
        self._chk(self._L.rgc_mapreg_associate(self._h, 0 if kind == "edge" else 1, fp, f.shape[0], q.ctypes.data_as(dp), tt.ctypes.data_as(dp),
                                               out.ctypes.data_as(dp), C.byref(nv)))
        valid = out[:, 7] != 0
        if kind == "edge":
            return dict(valid=valid, a=out[:, 0:3].copy(), b=out[:, 3:6].copy(), var=out[:, 6].copy(), n_valid=nv.value)
        return dict(valid=valid, n=out[:, 0:3].copy(), d=out[:, 3].copy(), var=out[:, 6].copy(), n_valid=nv.value)

    @staticmethod
    def _ground(d):
        if d is None:
            return None
        g = _lib.MapregGround()
        for k in ("last_v1", "last_v2", "last_norm", "cur_norm", "last_t"):
            setattr(g, k, (C.c_double * 3)(*[float(v) for v in d[k]]))
        for k in ("q_history", "last_q"):
            setattr(g, k, (C.c_double * 4)(*[float(v) for v in d[k]]))
        g.last_distance, g.cur_distance, g.p_var = float(d["last_distance"]), float(d["cur_distance"]), float(d.get("p_var", 0.2))
        return g

    @staticmethod
    def _imu(d):
        if d is None:
            return None
        m = _lib.MapregImu()
        m.delta_q = (C.c_double * 4)(*[float(v) for v in d["delta_q"]])
        m.imu_cov, m.pr_var = float(d["imu_cov"]), float(d.get("pr_var", 0.02))
        m.pitch_cur, m.roll_cur, m.pitch_last, m.roll_last = (float(d[k]) for k in ("pitch_cur", "roll_cur", "pitch_last", "roll_last"))
        return m

    def optimize(self, corner_cur, surf_cur, corner_last, surf_last, q_w_curr, t_w_curr, q_w_last, t_w_last, ground_cur=None, ground_last=None,
                 imu=None):
        """The two-pass associate + solve block; returns (q_w_curr, t_w_curr, q_w_last, t_w_last, report) with report = None when
        the size gate of :1069 is not met (poses returned unchanged).  ground_cur / ground_last: dicts with the fields of
        rgc_mapreg_ground (the Ground_DeltaFactor_goable blocks of :1314-1340), or None.  imu: dict with the fields of
        rgc_mapreg_imu (the RelativeRFactor / PitchRollFactor block of :1285-1312), or None."""
        cc, ccp = _f32(corner_cur, 4); sc, scp = _f32(surf_cur, 4); cl, clp = _f32(corner_last, 4); sl, slp = _f32(surf_last, 4)
        x = np.concatenate([np.asarray(q_w_curr, float), np.asarray(t_w_curr, float), np.asarray(q_w_last, float), np.asarray(t_w_last, float)])
        x = np.ascontiguousarray(x, np.float64)
        rep = (_lib.MapregReport * 2)()
        gate = C.c_int(0)
        gc, gl, im = self._ground(ground_cur), self._ground(ground_last), self._imu(imu)
        self._chk(self._L.rgc_mapreg_optimize(self._h, ccp, cc.shape[0], scp, sc.shape[0], clp, cl.shape[0], slp, sl.shape[0],
                                              C.byref(gc) if gc else None, C.byref(gl) if gl else None,
                                              C.byref(im) if im else None, x.ctypes.data_as(C.POINTER(C.c_double)), rep, C.byref(gate)))
        report = None if gate.value else [{k: getattr(r, k) for k, _ in _lib.MapregReport._fields_} for r in rep]
        return x[0:4].copy(), x[4:7].copy(), x[7:11].copy(), x[11:14].copy(), report
