"""numpy/scipy restatement of the scalar host stages of the odometer frame body (B1, B7, B8, C9), used only to
check librgc_hip.so's rgc_fuse_pose / rgc_compose_pose / rgc_imu_preintegrate / rgc_extract_pose.
TEST INFRASTRUCTURE ONLY; PARITY UNPINNED (no Ceres / reference build here: the minimiser of the same residual
blocks is computed independently with scipy.optimize.least_squares).  Citations: /root/reference/rgc_slam/."""
import math

import numpy as np
from scipy.optimize import least_squares


def qmul(a, b):  # x,y,z,w Hamilton product
    ax, ay, az, aw = a
    bx, by, bz, bw = b
    return np.array([aw * bx + ax * bw + ay * bz - az * by, aw * by - ax * bz + ay * bw + az * bx,
                     aw * bz + ax * by - ay * bx + az * bw, aw * bw - ax * bx - ay * by - az * bz])


def qconj(q):
    return np.array([-q[0], -q[1], -q[2], q[3]])


def q2R(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def R2ypr(R):
    """include/rgc_slam/utility.h:105-121 (degrees)"""
    n, o, a = R[:, 0], R[:, 1], R[:, 2]
    y = math.atan2(n[1], n[0])
    p = math.atan2(-n[2], n[0] * math.cos(y) + n[1] * math.sin(y))
    r = math.atan2(a[0] * math.sin(y) - a[1] * math.cos(y), -o[0] * math.sin(y) + o[1] * math.cos(y))
    return np.array([y, p, r]) / math.pi * 180.0


def ypr2R(ypr):
    """utility.h:123-147"""
    y, p, r = np.asarray(ypr) / 180.0 * math.pi
    Rz = np.array([[math.cos(y), -math.sin(y), 0], [math.sin(y), math.cos(y), 0], [0, 0, 1]])
    Ry = np.array([[math.cos(p), 0, math.sin(p)], [0, 1, 0], [-math.sin(p), 0, math.cos(p)]])
    Rx = np.array([[1, 0, 0], [0, math.cos(r), -math.sin(r)], [0, math.sin(r), math.cos(r)]])
    return Rz @ Ry @ Rx


def residuals(q, t, c):
    """c: dict with the fields of rgc_fuse_in (lidarFactor.hpp:132-172,228-265,311-350; RGC_odometer.cpp:1031-1119)"""
    r = list(2 * qmul(qconj(c["q_lidar"]), q)[:3] / c["fitness"])
    if c["use_ground"]:
        r += list((t - c["t_lidar"]) / (c["fitness"] / 10))
        gl, gc = c["ground_last"], c["ground_cur"]
        nc = q2R(q) @ gc[0:3]
        dt = q2R(c["q_w_curr_f"]) @ t
        pv = c["ground_cov"]
        r += [(gl[9] - (gc[9] + dt[2])) / (pv / 1000), abs(gl[3:6] @ nc) / (pv * 10), abs(gl[6:9] @ nc) / (pv * 10)]
    if c["use_imu"]:
        d = R2ypr(q2R(c["q_imu"]))
        cov = 0.0005 if np.linalg.norm(d) > 0.6 else 1 - c["fitness"]
        r += list(2 * qmul(qconj(c["q_imu"]), q)[:3] / cov)
    return np.asarray(r)


def plus(q, d):
    n = np.linalg.norm(d)
    if n == 0:
        return q
    return qmul(np.array([*(math.sin(n) / n * d), math.cos(n)]), q)


def fuse(c):
    q0, t0 = np.asarray(c["q_lidar"], float), np.asarray(c["t_lidar"], float)
    nd = 6 if c["use_ground"] else 3

    def fun(x):
        d = np.zeros(6)
        d[:nd] = x
        return residuals(plus(q0, d[:3]), t0 + d[3:], c)
    sol = least_squares(fun, np.zeros(nd), method="lm", xtol=1e-15, ftol=1e-15, gtol=1e-15, x_scale=1.0)
    d = np.zeros(6)
    d[:nd] = sol.x
    q = plus(q0, d[:3])
    return q / np.linalg.norm(q), t0 + d[3:]


def compose(q_w, t_w, q_f, t_f, t_l, use_imu, R_imu):
    """RGC_odometer.cpp:1194-1214"""
    Rw = q2R(q_w)
    t1, t2 = Rw @ t_f, Rw @ t_l
    tl = Rw.T @ np.array([t2[0], t2[1], t1[2]])
    t_w2 = t_w + Rw @ tl
    q = qmul(q_w, q_f)
    q /= np.linalg.norm(q)
    if use_imu:
        yw, yi = R2ypr(q2R(q)), R2ypr(R_imu)
        yw[1] = 0.95 * yw[1] + 0.05 * yi[1]
        yw[2] = 0.95 * yw[2] + 0.05 * yi[2]
        return ypr2R(yw), t_w2, tl      # rotation matrix (quaternion sign is arbitrary)
    return q2R(q), t_w2, tl


def imu_delta_q(stamps, gyr, prev_time, cur_time):
    """RGC_odometer.cpp:899-930, 1418-1422"""
    q = np.array([0, 0, 0, 1.0])
    n = len(stamps)
    for i in range(n):
        if i == 0:
            dt = stamps[0] - prev_time
        elif i == n - 1:
            dt = cur_time - stamps[i - 1]
        else:
            dt = stamps[i] - stamps[i - 1]
        q = qmul(q, np.array([*(gyr[i] * dt / 2), 1.0]))
        q /= np.linalg.norm(q)
    return q


class ImuFilter:
    """vg_ICP::imu_callback + ComplementaryFilter + Mid_Filter (RGC_odometer.cpp:444-486, 545-625; utility.h), restated in numpy.
    push() returns the bias-free (acc, gyr) the callback buffers, or None while the first 100 messages are dropped."""
    BA = np.array([0.23054, -0.22046, -0.14313])
    BG = np.array([0.00127, -0.00061, -0.00267])
    SIZES = (201, 41, 41)

    def __init__(self):
        self.dropped, self.count, self.t_last = 0, 0, 0.0
        self.roll = self.pitch = self.yaw = 0.0
        self.roll_last = self.pitch_last = 0.0
        self.Rwi = np.eye(3)
        self.buf = [np.zeros(n) for n in self.SIZES]
        self.pos = [0, 0, 0]

    def _mf(self, axis, x):
        b, n = self.buf[axis], self.SIZES[axis]
        b[self.pos[axis]] = x
        self.pos[axis] = (self.pos[axis] + 1) % n
        return float(np.sort(b)[(n - 1) // 2])

    def push(self, t, acc, gyr):
        if self.dropped < 100:
            self.dropped += 1
            return None
        r2d = 180.0 / np.pi
        a, g = np.asarray(acc, float) - self.BA, np.asarray(gyr, float) - self.BG
        self.count += 1
        dt = 0.005 if self.count == 1 else t - self.t_last
        ax, ay, az = self._mf(0, a[0]), self._mf(1, a[1]), self._mf(2, a[2])
        k = 0.9 if self.count < 300 else 0.002
        gx, gy, gz = g
        if abs(gz * r2d) < 0.2:
            gz = 0.0
        if self.count > 300:
            m = ypr2R(np.array([0.0, self.pitch * r2d, self.roll * r2d])) @ np.array([0, 0, 9.81])
            rx = abs(m[0]) / abs(ax)
            if abs(ax) > 0.3 and rx < 0.8:
                ax = rx * ax + (1 - rx) * m[0]
            ry = abs(m[1]) / abs(ay)
            if abs(ay) > 0.3 and ry < 0.8:
                ay = ry * ay + (1 - ry) * m[1]
        roll_acc, pitch_acc = np.arctan2(ay, az), -np.arctan2(ax, az)
        cr, sr, cp, sp = np.cos(self.roll), np.sin(self.roll), np.cos(self.pitch), np.sin(self.pitch)
        M = np.array([[1, 0, -sp], [0, cr, sr * cp], [0, -sr, cr * cp]])
        gx, gy, gz = np.linalg.solve(M, np.array([gx, gy, gz]))
        roll = k * roll_acc + (1 - k) * (self.roll + gx * dt)
        pitch = k * pitch_acc + (1 - k) * (self.pitch + gy * dt)
        yaw = self.yaw + gz / 0.9998 * dt
        if abs(gz * r2d) > 5.0:
            roll = 0.005 * roll + 0.995 * self.roll_last
            pitch = 0.005 * pitch + 0.995 * self.pitch_last
        nrp = lambda x: x - np.pi if x > np.pi / 2 else (x + np.pi if x < -np.pi / 2 else x)
        na = lambda x: x - 2 * np.pi if x > np.pi else (x + 2 * np.pi if x < -np.pi else x)
        self.roll, self.pitch, self.yaw = nrp(roll), nrp(pitch), na(yaw)
        self.Rwi = ypr2R(np.array([self.yaw, self.pitch, self.roll]) * r2d)
        self.t_last, self.roll_last, self.pitch_last = t, self.roll, self.pitch
        return a, g


class GroundGate:
    """the ground-change detector of RGC_odometer.cpp:1034-1087, restated"""

    def __init__(self):
        self.gflag, self.changegroundflag = 0, 25
        self.q_delta = np.array([0, 0, 0, 1.0])
        self.history = []

    def remember(self):
        self.history.append(self.q_delta.copy())

    def step(self, g_last, g_cur, q_l, t_l, dq_imu, q_w):
        q_w = np.asarray(q_w, float)
        if g_last is not None and g_cur is not None:
            gl, gc = np.asarray(g_last, float), np.asarray(g_cur, float)
            nc = q2R(np.asarray(q_l, float)) @ gc[:3]
            dcur = gc[9] + nc @ np.asarray(t_l, float)
            e1 = np.linalg.norm(gl[9] * gl[:3] - dcur * nc)
            e2 = abs(gl[3:6] @ nc)
            pitch = R2ypr(q2R(np.asarray(dq_imu, float)))[1] if dq_imu is not None else 0.0
            if e1 >= 0.02 and e2 >= 0.02 and abs(pitch) > 0.5:
                self.changegroundflag, self.gflag = 0, 1
        if self.gflag == 1 and self.changegroundflag < 25:
            self.changegroundflag += 1
            if self.changegroundflag == 25:
                now = R2ypr(q2R(q_w))
                best, pick = 1000.0, None
                for h in self.history:
                    y = R2ypr(q2R(h))
                    e = np.hypot(y[1] - now[1], y[2] - now[2])
                    if e < best:
                        best, pick = e, h
                if best < 4 and pick is not None:
                    self.q_delta = pick.copy()
                else:
                    self.q_delta = q_w.copy()
                    self.history.append(self.q_delta.copy())
                self.gflag = 0
        qf = qmul(qconj(self.q_delta), q_w)
        return self.gflag, qf / np.linalg.norm(qf)
