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


# ======================================================================================================================
# vg_ICP::ICP_thread per-frame body (src/RGC_odometer.cpp:932-1322) as a host-side class.  The per-point work goes to the
# HIP library; this class only holds the scalar state machine of the node (poses, ground_last, the 3-keyframe sub-map).
# ======================================================================================================================
class HipBackend:
    """The operations the frame body needs, each one C-ABI call (or a few) into librgc_hip.so."""

    def __init__(self, device: int = 0, scan_line: int = 16):
        from . import frontend, local_map, registration
        self.fe = frontend.ScanRegistration(scan_line, device=device)
        self.pre = Preprocessor(device)
        self.reg = registration.odometer_vgicp(device)
        self.map = local_map.RollingLocalMap(self.reg)
        self._L = _lib.load()

    def close(self):
        self.fe.close(); self.pre.close(); self.reg.close()

    def frontend(self, raw):
        return self.fe.laserCloudHandler(raw, diagnostics=False)

    def deskew(self, xyzi, q, t):
        return self.pre.adjustDistortion(xyzi, q, t)

    def voxelgrid(self, xyzi, leaf):
        return self.pre.voxelGridFilter(xyzi, leaf)

    def transform(self, xyzi, q, t):
        return self.pre.transformPointCloud(xyzi, q, t)

    def register(self, source, target, guess):
        """-> (final 4x4 float32, fitness)  RGC_odometer.cpp:998-1011"""
        self.reg.setInputTarget(target)
        self.reg.setInputSource(source)
        self.reg.align(guess, want_output=False, want_fitness=True)
        return self.reg.getFinalTransformation(), self.reg.getFitnessScore()

    # f2: the local map resident on the device (rgc_map_*), registration in the map frame
    def map_reset(self, origin):
        self.map.reset(origin)

    def map_insert(self, xyzi, q, t):
        return self.map.insert(xyzi, q, t)

    def map_evict(self, max_keyframes, center=None, radius=0.0):
        return self.map.evict(max_keyframes, center, radius)

    def map_rebase(self, origin):
        self.map.rebase(origin)

    def map_register(self, source, guess, leaf):
        """-> (final 4x4 float32 in the map frame, fitness); the target is rebuilt only if a keyframe changed"""
        self.map.commit(leaf)
        self.reg.setInputSource(source)
        self.reg.align(guess, want_output=False, want_fitness=True)
        return self.reg.getFinalTransformation(), self.reg.getFitnessScore()

    def extract(self, T):
        q, t = np.empty(4), np.empty(3)
        dp = C.POINTER(C.c_double)
        Tf = np.ascontiguousarray(T, dtype=np.float32)
        rc = self._L.rgc_extract_pose(Tf.ctypes.data_as(C.POINTER(C.c_float)), q.ctypes.data_as(dp), t.ctypes.data_as(dp))
        if rc:
            raise RgcError(rc, "rgc_extract_pose")
        return q, t

    def fuse(self, q_l, t_l, fitness, use_ground, g_last, g_cur, q_wf, q_imu=None):
        fin = _lib.FuseIn()
        self._L.rgc_default_fuse_in(C.byref(fin))
        fin.q_lidar_xyzw[:] = list(q_l); fin.t_lidar[:] = list(t_l); fin.fitness = float(fitness)
        fin.use_ground = int(use_ground)
        if use_ground:
            fin.ground_last[:] = list(g_last); fin.ground_cur[:] = list(g_cur); fin.q_w_curr_f_xyzw[:] = list(q_wf)
        if q_imu is not None:                                                        # USE_IMU && imuflag == 1, :1104-1119
            fin.use_imu = 1
            fin.q_imu_xyzw[:] = list(q_imu)
        q, t = np.empty(4), np.empty(3)
        dp = C.POINTER(C.c_double)
        rc = self._L.rgc_fuse_pose(C.byref(fin), q.ctypes.data_as(dp), t.ctypes.data_as(dp), None)
        if rc:
            raise RgcError(rc, "rgc_fuse_pose")
        return q, t

    def compose(self, q_w, t_w, q_f, t_f, t_l, R_imu_wl=None):
        qo, to, tl = np.empty(4), np.empty(3), np.empty(3)
        dp = C.POINTER(C.c_double)
        a = [np.ascontiguousarray(x, dtype=np.float64) for x in (q_w, t_w, q_f, t_f, t_l)]
        R = None if R_imu_wl is None else np.ascontiguousarray(R_imu_wl, dtype=np.float64).reshape(9)
        rc = self._L.rgc_compose_pose(*[x.ctypes.data_as(dp) for x in a], 0 if R is None else 1, None if R is None else R.ctypes.data_as(dp),
                                      qo.ctypes.data_as(dp), to.ctypes.data_as(dp), tl.ctypes.data_as(dp))
        if rc:
            raise RgcError(rc, "rgc_compose_pose")
        return qo, to, tl

    # B1: the IMU side (rgc_imu_filter_*, rgc_imu_preintegrate) and the ground-change detector (rgc_ground_gate_*)
    def imu_filter(self):
        L, dp = self._L, C.POINTER(C.c_double)

        class _Filter:
            def __init__(self):
                self.s = _lib.ImuFilter()
                L.rgc_imu_filter_init(C.byref(self.s))

            def push(self, t, acc, gyr):
                a, g = np.ascontiguousarray(acc, np.float64), np.ascontiguousarray(gyr, np.float64)
                ao, go = np.empty(3), np.empty(3)
                rc = L.rgc_imu_filter_push(C.byref(self.s), float(t), a.ctypes.data_as(dp), g.ctypes.data_as(dp), ao.ctypes.data_as(dp), go.ctypes.data_as(dp))
                if rc < 0:
                    raise RgcError(rc, "rgc_imu_filter_push")
                return (ao, go) if rc == 1 else None

            @property
            def Rwi(self):
                return np.array(self.s.Rwi[:]).reshape(3, 3)
        return _Filter()

    def imu_preintegrate(self, stamps, gyr, acc, prev_time, cur_time):
        return imu_preintegrate(stamps, gyr, acc, prev_time, cur_time)

    def ground_gate(self):
        L, dp = self._L, C.POINTER(C.c_double)

        class _Gate:
            def __init__(self):
                self.s = _lib.GroundGate()
                L.rgc_ground_gate_init(C.byref(self.s))

            def remember(self):
                L.rgc_ground_gate_remember(C.byref(self.s))

            def step(self, g_last, g_cur, q_l, t_l, dq_imu, q_w):
                def p(x):
                    return None if x is None else np.ascontiguousarray(x, np.float64).ctypes.data_as(dp)
                keep = [None if x is None else np.ascontiguousarray(x, np.float64) for x in (g_last, g_cur, q_l, t_l, dq_imu, q_w)]
                qf = np.empty(4)
                have = g_last is not None and g_cur is not None
                rc = L.rgc_ground_gate_step(C.byref(self.s), *[None if (x is None or (i < 4 and not have)) else x.ctypes.data_as(dp) for i, x in enumerate(keep)],
                                            qf.ctypes.data_as(dp))
                if rc < 0:
                    raise RgcError(rc, "rgc_ground_gate_step")
                return rc, qf
        return _Gate()

    def ypr2R(self, ypr_deg):
        R, y = np.empty(9), np.ascontiguousarray(ypr_deg, np.float64)
        dp = C.POINTER(C.c_double)
        self._L.rgc_ypr2R(y.ctypes.data_as(dp), R.ctypes.data_as(dp))
        return R.reshape(3, 3)

    def R2ypr_m(self, R):
        ypr, Rc = np.empty(3), np.ascontiguousarray(R, np.float64).reshape(9)
        dp = C.POINTER(C.c_double)
        self._L.rgc_R2ypr(Rc.ctypes.data_as(dp), ypr.ctypes.data_as(dp))
        return ypr

    def R2ypr(self, q):
        x, y, z, w = q
        R = np.ascontiguousarray([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                                  [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                                  [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]], dtype=np.float64)
        ypr = np.empty(3)
        dp = C.POINTER(C.c_double)
        self._L.rgc_R2ypr(R.ctypes.data_as(dp), ypr.ctypes.data_as(dp))
        return ypr


def _qconj(q):
    return np.array([-q[0], -q[1], -q[2], q[3]])


def _qmul(a, b):
    ax, ay, az, aw = a
    bx, by, bz, bw = b
    return np.array([aw * bx + ax * bw + ay * bz - az * by, aw * by - ax * bz + ay * bw + az * bx,
                     aw * bz + ax * by - ay * bx + az * bw, aw * bw - ax * bx - ay * by - az * bz])


def _qrot(q, v):
    x, y, z, w = q
    u = 2 * np.cross([x, y, z], v)
    return v + w * u + np.cross([x, y, z], u)


class Odometer:
    """One sequence of the odometry node (vg_ICP: the callbacks' buffering + ICP_thread's frame body), USE_IMU = 0 or 1 like the
    launch file's switch (launch/run.launch:18).  With the IMU: imu_callback feeds the attitude filter and the sample buffer
    (:444-486), the gyro's pre-integrated rotation is the registration's initial guess (:883-931, 993-996) and a factor of the
    fusion (:1104-1119), pitch / roll are blended towards the filter's attitude (:1206-1214) and the first `first_frames` sweeps
    only initialise the pose from it (:857-882).  The ground-change detector (:1034-1087) runs in both modes (without an IMU its
    pitch-rate condition never fires).  Not mirrored: the two gravity-direction solves of the first frame (:1121-1186), whose
    results (g_init, q_body2world) do not enter the pose.  Constants: RGC_odometer.cpp:280-310, 387-393."""

    keyframeAddingDistance, keyframeAddingAngle = 0.3, 0.2      # :280-281 (the angle is compared against DEGREES)
    slipwide = 3                                                # :299
    planeResolution1, planeResolution2 = 0.2, 0.3               # :305-306
    R_il_ypr = (-1.29, -0.15, 0.65)                             # :387, degrees

    def __init__(self, backend, use_ground: bool = True, use_imu: bool = False, first_frames: int = 0, init_yaw: float = 0.0):
        self.b = backend
        self.USE_GROUND, self.USE_IMU = use_ground, use_imu
        self.first_frames, self.init_yaw = first_frames, init_yaw                    # firstflagnum (:303), init_yaw (:358)
        self.q_w_curr, self.t_w_curr = np.array([0, 0, 0, 1.0]), np.zeros(3)
        self.q_last_curr, self.t_last_curr = np.array([0, 0, 0, 1.0]), np.zeros(3)   # para_q / para_t, :26-34
        self.full_last = None
        self.ground_last = None
        self.surrounding, self.surrounding_q, self.surrounding_t = [], [], []
        self.submap = np.zeros((0, 4), np.float32)
        self.submapflag = 0
        self.fitness = 1.0
        self.frames = 0
        self.gate = backend.ground_gate()
        self.gflag = 0
        self.prev_time = 0.0
        self.delta_q_imu = None
        if use_imu:
            self.imu = backend.imu_filter()
            self.imu_buf = []                                                         # accBuf / gyrBuf: (t, acc, gyr), bias-free
            self.R_il = backend.ypr2R(self.R_il_ypr)

    # ---- callbacks' side ----
    def imu_callback(self, stamp, acc, gyr):
        """sensor_msgs/Imu -> attitude filter + sample buffer (:444-486)"""
        s = self.imu.push(stamp, acc, gyr)
        if s is not None:
            self.imu_buf.append((float(stamp), s[0], s[1]))

    def _imu_interval(self, t0, t1):
        """getIMUInterval, :1376-1416"""
        buf = self.imu_buf
        if not buf or (t0 <= buf[0][0] and t1 <= buf[0][0]) or not (t1 <= buf[-1][0]):
            return None
        while buf[0][0] <= t0:
            buf.pop(0)
        out = []
        while buf[0][0] < t1:
            out.append(buf.pop(0))
        out.append(buf[0])
        return out

    def _begin_frame(self, stamp):
        """:857-931, 955-956: pose initialisation of the first sweeps and the IMU guess; False = this sweep is dropped"""
        if self.frames < self.first_frames:
            self.frames += 1
            self.prev_time = stamp
            self.t_w_curr = np.zeros(3)
            if self.USE_IMU:
                y = self.b.R2ypr_m(self.imu.Rwi @ self.R_il) + np.array([self.init_yaw, 0.0, 0.0])
                self.q_w_curr = _R2q(self.b.ypr2R(y))
            return False
        if self.USE_IMU:
            iv = self._imu_interval(self.prev_time, stamp)
            if iv is None:
                return False
            self.delta_q_imu = self.b.imu_preintegrate([x[0] for x in iv], [x[2] for x in iv], [x[1] for x in iv], self.prev_time, stamp)
            self.q_last_curr = self.delta_q_imu.copy()                               # :929-930
        self.prev_time = stamp
        return True

    def _fuse_and_compose(self, q_l, t_l, ground_cur, ground_valid):
        """:1025-1214: ground-change detector, fusion, composition, gravity blend"""
        b = self.b
        have = ground_valid and self.ground_last is not None
        self.gflag, q_wf = self.gate.step(self.ground_last if have else None, ground_cur if have else None, q_l, t_l, self.delta_q_imu, self.q_w_curr)
        use_ground = self.USE_GROUND and have and self.gflag == 0                    # :1088
        q_f, t_f = b.fuse(q_l, t_l, self.fitness, use_ground, self.ground_last, ground_cur, q_wf, self.delta_q_imu if self.USE_IMU else None)
        R_imu = self.imu.Rwi @ self.R_il if self.USE_IMU else None                   # :1209
        self.q_w_curr, self.t_w_curr, t_lc = b.compose(self.q_w_curr, self.t_w_curr, q_f, t_f, t_l, R_imu)   # :1194-1214
        self.q_last_curr, self.t_last_curr = q_f, t_lc

    def process(self, raw_xyzi, stamp=None):
        """One LiDAR message through front-end + frame body; returns (q_w_curr xyzw, t_w_curr), or None for a dropped sweep."""
        b = self.b
        stamp = 0.1 * self.frames if stamp is None else float(stamp)
        if not self._begin_frame(stamp):
            return None
        fe = b.frontend(raw_xyzi)
        full, ground_cur = fe["cloud"], fe["groundparam"]
        full = b.deskew(full, self.q_last_curr, self.t_last_curr)                    # adjustDistortion, :958
        if self.full_last is not None and len(self.full_last):
            if self.submapflag == 0:                                                 # :963-972
                self.surrounding.append(self.full_last)
                self.surrounding_q.append(np.array([0, 0, 0, 1.0])); self.surrounding_t.append(np.zeros(3))
                self.gate.remember()
                self.submap = np.concatenate([self.submap, self.full_last])
            self.submapflag += 1
            source = b.voxelgrid(full, self.planeResolution1)                        # :976-983
            target = b.voxelgrid(self.submap, self.planeResolution2)                 # :985-991
            T2 = np.eye(4, dtype=np.float32)                                         # :993-996
            T2[:3, :3] = _q2R(self.q_last_curr).astype(np.float32)
            T2[:3, 3] = self.t_last_curr.astype(np.float32)
            T, self.fitness = b.register(source, target, T2)                         # :998-1010
            q_l, t_l = b.extract(T)                                                  # :1011-1016
            self._fuse_and_compose(q_l, t_l, ground_cur, fe["ground_valid"])         # :1025-1214
            # sub-map maintenance, :1218-1256
            if self.surrounding:
                yb, yc = b.R2ypr(self.surrounding_q[-1]), b.R2ypr(self.q_w_curr)
                d = np.float32(self.surrounding_t[-1] - self.t_w_curr)
                dy, dp_, dr = np.float32(yb[0] - yc[0]), np.float32(yb[1] - yc[1]), np.float32(yb[2] - yc[2])
                if dy > np.pi: dy -= 2 * np.pi
                if dy < -np.pi: dy += 2 * np.pi
                if (abs(dr) > self.keyframeAddingAngle or abs(dp_) > self.keyframeAddingAngle or abs(dy) > self.keyframeAddingAngle or
                        float(np.sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2])) > self.keyframeAddingDistance or self.submapflag < self.slipwide - 1):
                    self.surrounding.append(b.transform(source, self.q_w_curr, self.t_w_curr))
                    self.surrounding_q.append(self.q_w_curr.copy()); self.surrounding_t.append(self.t_w_curr.copy())
            self.submap = np.zeros((0, 4), np.float32)
            if len(self.surrounding) > self.slipwide:
                self.surrounding.pop(0); self.surrounding_q.pop(0); self.surrounding_t.pop(0)
            if len(self.surrounding) > 1:
                qi = _qconj(self.q_w_curr)
                ti = -_qrot(qi, self.t_w_curr)
                self.submap = np.concatenate([b.transform(c, qi, ti) for c in self.surrounding])
        self.full_last = full                                                        # :1319-1322
        self.ground_last = ground_cur if fe["ground_valid"] else self.ground_last
        self.frames += 1
        return self.q_w_curr.copy(), self.t_w_curr.copy()


def _R2q(R):
    """Eigen's quaternion-from-matrix, positive-trace branch (the attitudes met here are near the identity)"""
    w = np.sqrt(max(0.0, 1.0 + R[0, 0] + R[1, 1] + R[2, 2])) / 2
    q = np.array([(R[2, 1] - R[1, 2]) / (4 * w), (R[0, 2] - R[2, 0]) / (4 * w), (R[1, 0] - R[0, 1]) / (4 * w), w])
    return q / np.linalg.norm(q)


def imu_preintegrate(stamps, gyr, acc, prev_time, cur_time):
    """IMU_preintegration (RGC_odometer.cpp:883-931): the gyro's rotation between two sweeps as a quaternion xyzw (rgc_imu_preintegrate, host code
    of the library: no context needed); samples = those of getIMUInterval, bias-corrected"""
    L = _lib.load()
    st, g, a = (np.ascontiguousarray(x, np.float64) for x in (stamps, gyr, acc))
    dq = np.empty(4)
    dp = C.POINTER(C.c_double)
    rc = L.rgc_imu_preintegrate(st.ctypes.data_as(dp), g.ctypes.data_as(dp), a.ctypes.data_as(dp), len(st), float(prev_time), float(cur_time),
                                dq.ctypes.data_as(dp), None, None, None)
    if rc:
        raise RgcError(rc, "rgc_imu_preintegrate")
    return dq


def imu_rotation_priors(poses, dt=0.1, **imu_kw):
    """Initial guesses of a sequence the way the odometer forms them with USE_IMU = 1 (RGC_odometer.cpp:929-931, 993-996), from a
    synthetic IMU stream of the trajectory: guess_k = pose_{k-1} * [R(gyro delta over sweep k) | translation delta of sweep k - 1] for
    k >= 1 (world frame; pose_{k-1} is the true one, so every frame's prior stands alone).  Returns {k: 4x4 float32}."""
    from . import synth
    stamps, acc, gyr = synth.make_imu(poses, dt=dt, **imu_kw)
    bg = np.asarray(imu_kw.get("bg", (0.00127, -0.00061, -0.00267)))
    ba = np.asarray(imu_kw.get("ba", (0.23054, -0.22046, -0.14313)))
    gyr, acc = gyr - bg, acc - ba                      # the odometer subtracts its calibrated biases (utility.h:253-254)
    out = {}
    for k in range(1, len(poses)):
        t0, t1 = (k - 1) * dt, k * dt
        sel = np.where((stamps > t0) & (stamps < t1))[0]
        sel = np.append(sel, sel[-1] + 1)               # getIMUInterval keeps the first sample at or after t1 as well
        dq = imu_preintegrate(stamps[sel], gyr[sel], acc[sel], t0, t1)
        T2 = np.eye(4)
        T2[:3, :3] = _q2R(dq)
        if k >= 2:
            T2[:3, 3] = (np.linalg.inv(poses[k - 2]) @ poses[k - 1])[:3, 3]
        out[k] = (np.asarray(poses[k - 1], np.float64) @ T2).astype(np.float32)
    return out


def _q2R(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


class RollingOdometer(Odometer):
    """f2: the same frame body with the local map resident on the device in a MAP frame (world minus an origin
    kept near the sensor) instead of a host deque that is re-framed, re-filtered and re-uploaded per frame (:1218-1256,
    985-991, 1007).  The registration runs in the map frame: guess = T_w_curr * T_last_curr, result = the scan's new world pose,
    from which the reference's lidar delta (q_last_curr_l, t_last_curr_l) is recovered for the fusion.  Keyframe test, deque
    length and the first-keyframe rule are the reference's; eviction may additionally be by distance (radius > 0)."""

    rebase_distance = 50.0   # the origin follows the sensor so that fp32 map coordinates stay < ~64 m (ulp 4e-6 m)

    def __init__(self, backend, use_ground: bool = True, max_keyframes=None, radius: float = 0.0, **kw):
        super().__init__(backend, use_ground, **kw)
        self.max_keyframes = self.slipwide if max_keyframes is None else max_keyframes
        self.radius = radius
        self.origin = np.zeros(3)
        self.kf_q, self.kf_t = None, None
        self.n_commits = 0

    def process(self, raw_xyzi, stamp=None):
        b = self.b
        stamp = 0.1 * self.frames if stamp is None else float(stamp)
        if not self._begin_frame(stamp):
            return None
        fe = b.frontend(raw_xyzi)
        full, ground_cur = fe["cloud"], fe["groundparam"]
        full = b.deskew(full, self.q_last_curr, self.t_last_curr)                    # adjustDistortion, :958
        if self.full_last is not None and len(self.full_last):
            if self.submapflag == 0:                                                 # :963-972: the previous sweep is keyframe 0
                self.origin = self.t_w_curr.copy()
                b.map_reset(self.origin)
                b.map_insert(self.full_last, np.array([0, 0, 0, 1.0]), np.zeros(3))
                self.kf_q, self.kf_t = np.array([0, 0, 0, 1.0]), np.zeros(3)
                self.gate.remember()
            self.submapflag += 1
            source = b.voxelgrid(full, self.planeResolution1)                        # :976-983
            # the guess of :993-996 moved into the map frame: T_w_curr * T_last_curr
            q_g = _qmul(self.q_w_curr, self.q_last_curr)
            t_g = _qrot(self.q_w_curr, self.t_last_curr) + self.t_w_curr - self.origin
            T2 = np.eye(4, dtype=np.float32)
            T2[:3, :3] = _q2R(q_g / np.linalg.norm(q_g)).astype(np.float32)
            T2[:3, 3] = t_g.astype(np.float32)
            T, self.fitness = b.map_register(source, T2, self.planeResolution2)      # :985-991, 998-1010
            q_m, t_m = b.extract(T)                                                  # :1011-1016, map-frame pose of the scan
            qi = _qconj(self.q_w_curr)
            q_l = _qmul(qi, q_m)                                                     # back to the delta the fusion expects
            t_l = _qrot(qi, t_m + self.origin - self.t_w_curr)
            self._fuse_and_compose(q_l, t_l, ground_cur, fe["ground_valid"])         # :1025-1214
            # keyframe test of :1218-1239 against the newest keyframe's pose
            yb, yc = b.R2ypr(self.kf_q), b.R2ypr(self.q_w_curr)
            d = np.float32(self.kf_t - self.t_w_curr)
            dy, dp_, dr = np.float32(yb[0] - yc[0]), np.float32(yb[1] - yc[1]), np.float32(yb[2] - yc[2])
            if dy > np.pi: dy -= 2 * np.pi
            if dy < -np.pi: dy += 2 * np.pi
            if (abs(dr) > self.keyframeAddingAngle or abs(dp_) > self.keyframeAddingAngle or abs(dy) > self.keyframeAddingAngle or
                    float(np.sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2])) > self.keyframeAddingDistance or self.submapflag < self.slipwide - 1):
                b.map_insert(source, self.q_w_curr, self.t_w_curr)                   # :1237, once, never re-framed
                self.kf_q, self.kf_t = self.q_w_curr.copy(), self.t_w_curr.copy()
                b.map_evict(self.max_keyframes, self.t_w_curr if self.radius > 0 else None, self.radius)   # :1242-1247
                self.n_commits += 1
            if np.linalg.norm(self.t_w_curr - self.origin) > self.rebase_distance:
                self.origin = self.t_w_curr.copy()
                b.map_rebase(self.origin)
        self.full_last = full                                                        # :1319-1322
        self.ground_last = ground_cur if fe["ground_valid"] else self.ground_last
        self.frames += 1
        return self.q_w_curr.copy(), self.t_w_curr.copy()
