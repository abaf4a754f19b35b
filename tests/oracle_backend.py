"""CPU-oracle implementation of the backend interface rgc_slam_amd.odometry.Odometer expects -- used ONLY by the tests to
produce the reference trajectory of a sequence (the product's HipBackend never touches the oracle)."""
import os

import numpy as np

from oracle import oracle as orc
from oracle import py_fusion as pf


class OracleBackend:
    def __init__(self, scan_line=16):
        self.ns = scan_line

    def close(self):
        pass

    def frontend(self, raw):
        return orc.frontend(raw, n_scans=self.ns)

    def deskew(self, xyzi, q, t):
        return orc.deskew(xyzi, q, t)

    def voxelgrid(self, xyzi, leaf):
        return orc.voxelgrid_filter(np.ascontiguousarray(xyzi[:, :4]), leaf)

    def transform(self, xyzi, q, t):
        return orc.transform_cloud(xyzi, q, t)

    def register(self, source, target, guess):
        # the reference's setNumThreads(14) (RGC_odometer.cpp:1006): on a many-core host the oracle's short OpenMP loops get SLOWER with
        # more threads (0.6 against 4.3 scans/s at 256 against 14 threads on the GPU box: bench.py's cpu_baseline)
        r = orc.Registration(num_threads=min(14, os.cpu_count() or 1))
        r.set_target(target); r.set_source(source)
        T = r.align(guess)
        return T, r.fitness()

    def extract(self, T):
        T = np.asarray(T, dtype=np.float32)
        U, _, Vt = np.linalg.svd(T[:3, :3].astype(np.float64))        # Affine3f::rotation(): polar part
        R = (U @ Vt).astype(np.float32).astype(np.float64)
        # quaternion from rotation matrix (w >= 0 branch is enough for the small inter-frame rotations)
        w = np.sqrt(max(0.0, 1 + R[0, 0] + R[1, 1] + R[2, 2])) / 2
        q = np.array([(R[2, 1] - R[1, 2]) / (4 * w), (R[0, 2] - R[2, 0]) / (4 * w), (R[1, 0] - R[0, 1]) / (4 * w), w])
        return q.astype(np.float32).astype(np.float64), T[:3, 3].astype(np.float64)

    def fuse(self, q_l, t_l, fitness, use_ground, g_last, g_cur, q_wf, q_imu=None):
        c = dict(q_lidar=np.asarray(q_l), t_lidar=np.asarray(t_l), fitness=float(fitness), use_ground=bool(use_ground),
                 ground_last=np.asarray(g_last) if use_ground else None, ground_cur=np.asarray(g_cur) if use_ground else None,
                 q_w_curr_f=np.asarray(q_wf), ground_cov=0.2, use_imu=q_imu is not None,
                 q_imu=np.asarray(q_imu) if q_imu is not None else np.array([0, 0, 0, 1.0]))
        return pf.fuse(c)

    def compose(self, q_w, t_w, q_f, t_f, t_l, R_imu_wl=None):
        R, t, tl = pf.compose(np.asarray(q_w), np.asarray(t_w), np.asarray(q_f), np.asarray(t_f), np.asarray(t_l), R_imu_wl is not None, R_imu_wl)
        if R_imu_wl is not None:                      # the blended attitude comes back as a matrix
            w = np.sqrt(max(0.0, 1.0 + R[0, 0] + R[1, 1] + R[2, 2])) / 2
            q = np.array([(R[2, 1] - R[1, 2]) / (4 * w), (R[0, 2] - R[2, 0]) / (4 * w), (R[1, 0] - R[0, 1]) / (4 * w), w])
            return q / np.linalg.norm(q), t, tl
        q = pf.qmul(np.asarray(q_w), np.asarray(q_f))
        return q / np.linalg.norm(q), t, tl

    # B1: IMU attitude filter, gyro pre-integration, ground-change detector -- the numpy restatements of oracle/py_fusion.py
    def imu_filter(self):
        return pf.ImuFilter()

    def imu_preintegrate(self, stamps, gyr, acc, prev_time, cur_time):
        return pf.imu_delta_q(np.asarray(stamps, float), np.asarray(gyr, float), prev_time, cur_time)

    def ground_gate(self):
        return pf.GroundGate()

    def ypr2R(self, ypr_deg):
        return pf.ypr2R(np.asarray(ypr_deg, float))

    def R2ypr_m(self, R):
        return pf.R2ypr(np.asarray(R, float))

    def R2ypr(self, q):
        return pf.R2ypr(pf.q2R(np.asarray(q)))

    # ---- f2: the rolling local map composed from the oracle's own pieces (transform, VoxelGrid, registration), on the host ----
    def map_reset(self, origin):
        self._origin = np.asarray(origin, float).copy()
        self._kf = []                                   # [(id, points in the map frame, world translation)]
        self._next = 0
        self._target, self._leaf = None, None

    def map_insert(self, xyzi, q, t):
        pts = orc.transform_cloud(np.ascontiguousarray(xyzi, np.float32)[:, :4], q, np.asarray(t, float) - self._origin)
        self._kf.append((self._next, pts, np.asarray(t, float).copy()))
        self._next += 1
        self._target = None
        return self._next - 1

    def map_evict(self, max_keyframes, center=None, radius=0.0):
        n0 = len(self._kf)
        if center is not None and radius > 0:
            # the newest keyframe always stays (an empty map cannot be committed)
            self._kf = [k for i, k in enumerate(self._kf) if i == len(self._kf) - 1 or np.linalg.norm(k[2] - np.asarray(center, float)) <= radius]
        if max_keyframes > 0 and len(self._kf) > max_keyframes:
            self._kf = self._kf[-max_keyframes:]
        if len(self._kf) != n0:
            self._target = None
        return n0 - len(self._kf)

    def map_rebase(self, origin):
        d = self._origin - np.asarray(origin, float)
        self._kf = [(i, orc.transform_cloud(p, np.array([0, 0, 0, 1.0]), d), t) for i, p, t in self._kf]
        self._origin = np.asarray(origin, float).copy()
        self._target = None

    def map_points(self):
        return np.concatenate([p for _, p, _ in self._kf]) if self._kf else np.zeros((0, 4), np.float32)

    def map_target(self, leaf):
        if self._target is None or self._leaf != leaf:
            self._target, self._leaf = orc.voxelgrid_filter(np.ascontiguousarray(self.map_points()), leaf), leaf
        return self._target

    def map_register(self, source, guess, leaf):
        return self.register(source, self.map_target(leaf), guess)
