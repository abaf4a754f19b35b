"""vg_ICP::ICP_thread's per-frame body (src/RGC_odometer.cpp:848-1256, 1319-1322), restated line by line in numpy -- TEST INFRASTRUCTURE.
PARITY UNPINNED: the reference holds no recorded sequence and cannot be built here; this restatement is checked against nothing but itself
(tests/golden/fx_sequence.npz is its own output) and pins the product's mirrors, not the other way round.

This is the independent pin of the frame body's ORCHESTRATION: it keeps the reference's variable names and statement order (first_flag,
submapflag, surroundingCloud, histoary_pose, gflag / changegroundflag, q_w_curr_delta, t_last_curr_l, ...) and shares no code with the
product's mirrors (rgc_slam_amd.odometry.Odometer, rgc::OdometryNode).  The per-point stages are the CPU oracle's (oracle.oracle:
front-end, de-skew, VoxelGrid, FastVGICP, transformPointCloud); the Ceres problem of :1025-1119, 1188-1193 is solved by
oracle.py_fusion.fuse (scipy); the callback side (vg_ICP::imuCallback: complementary filter + median pre-filter) is
oracle.py_fusion.ImuFilter.  Everything between those calls is written out here.

Not restated, and why:
* the ROS buffers' synchronisation (:811-850) and `fullPointsBuf.size() < firstinit` (:852): transport; every sweep is handed over once;
* the two gravity-direction solves of the first registered frame (:1121-1186): g_init / q_body2world do not enter the pose;
* publishing, TUM / PCD writing (:1264-1355): no state.
Quaternions are numpy arrays x, y, z, w.  Angles of Utility::R2ypr are DEGREES (include/rgc_slam/utility.h:119-129).
"""
import numpy as np

from . import oracle as orc
from . import py_fusion as pf


def _quat_of_matrix(R, dtype=np.float64):
    """Eigen::QuaternionBase::operator=(MatrixBase) (Eigen/src/Geometry/Quaternion.h), in the scalar type of the matrix"""
    m = np.asarray(R, dtype)
    one, half = dtype(1), dtype(0.5)
    t = m[0, 0] + m[1, 1] + m[2, 2]
    q = np.zeros(4, dtype)
    if t > 0:
        t = np.sqrt(t + one)
        q[3] = half * t
        t = half / t
        q[0] = (m[2, 1] - m[1, 2]) * t
        q[1] = (m[0, 2] - m[2, 0]) * t
        q[2] = (m[1, 0] - m[0, 1]) * t
    else:
        i = 0
        if m[1, 1] > m[0, 0]:
            i = 1
        if m[2, 2] > m[i, i]:
            i = 2
        j, k = (i + 1) % 3, (i + 2) % 3
        t = np.sqrt(m[i, i] - m[j, j] - m[k, k] + one)
        q[i] = half * t
        t = half / t
        q[3] = (m[k, j] - m[j, k]) * t
        q[j] = (m[j, i] + m[i, j]) * t
        q[k] = (m[k, i] + m[i, k]) * t
    return q


def _normalized(q):
    return q / np.linalg.norm(q)


def _rot(q, v):
    return pf.q2R(q) @ np.asarray(v, float)


class IcpThread:
    # constants of the node, RGC_odometer.cpp
    keyframeAddingDistance = np.float32(0.3)   # :280
    keyframeAddingAngle = np.float32(0.2)      # :281 (compared with DEGREES)
    slipwide = 3                               # :299
    planeResolution1, planeResolution2 = 0.2, 0.3   # :305-306
    imuflag = 1                                # :329
    changegroundflag0 = 25                     # :327

    def __init__(self, USE_IMU=1, USE_GROUND=1, firstflagnum=10, init_xyz=(0.0, 0.0, 0.0), init_yaw=0.0, N_SCANS=16):
        self.USE_IMU, self.USE_GROUND, self.firstflagnum = USE_IMU, USE_GROUND, firstflagnum
        self.init_xyz, self.init_yaw, self.N_SCANS = np.asarray(init_xyz, float), float(init_yaw), N_SCANS
        self.q_w_curr, self.t_w_curr = np.array([0, 0, 0, 1.0]), np.zeros(3)            # :12-13
        self.q_w_curr_delta = np.array([0, 0, 0, 1.0])                                   # :20
        self.q_w_curr_f = np.array([0, 0, 0, 1.0])
        self.para_q, self.para_t = np.array([0, 0, 0, 1.0]), np.zeros(3)                 # :26-34: q_last_curr / t_last_curr are maps onto these
        self.delta_q_imu = np.array([0, 0, 0, 1.0])
        self.ground_last = np.zeros(11)                                                  # ground_s(): all zero, utility.h:382-389
        self.laserCloudFullLast = np.zeros((0, 4), np.float32)
        self.laserCloudsubmap = np.zeros((0, 4), np.float32)
        self.surroundingCloud, self.surrounding_q, self.surrounding_t = [], [], []
        self.histoary_pose = []
        self.gflag, self.changegroundflag = 0, self.changegroundflag0                    # :327-328
        self.first_flag, self.submapflag, self.frameCount = 0, 0, 0
        self.prevTime = 0.0
        self.vgicp_source = 0.0
        self.IMU = pf.ImuFilter()
        self.accBuf = []                                                                 # (t, acc, gyr): accBuf and gyrBuf move together
        self.R_il = pf.ypr2R(np.array([-1.29, -0.15, 0.65]))                             # :387

    # ---- vg_ICP::imuCallback: the filter's state and the two sample queues ----
    def imuCallback(self, stamp, acc, gyr):
        s = self.IMU.push(float(stamp), acc, gyr)
        if s is not None:
            self.accBuf.append((float(stamp), s[0], s[1]))

    # ---- :1376-1416 ----
    def getIMUInterval(self, t0, t1):
        accBuf = self.accBuf
        if not accBuf:
            return None
        if t0 <= accBuf[0][0] and t1 <= accBuf[0][0]:
            return None
        if t1 <= accBuf[-1][0]:
            while accBuf[0][0] <= t0:
                accBuf.pop(0)
            vec = []
            while accBuf[0][0] < t1:
                vec.append(accBuf.pop(0))
            vec.append(accBuf[0])
            return vec
        return None

    # ---- one (cloud, ground_param) pair: the body of the inner while loop, :848-1256 + :1319-1322; None = the sweep produced no pose ----
    def handle(self, raw_xyzi, stamp):
        fe = orc.frontend(raw_xyzi, n_scans=self.N_SCANS)        # the scanRegistration node: /velodyne_cloud_2 and /ground_param
        laserCloudFullRes, ground_cur = fe["cloud"], np.array(fe["groundparam"], float)
        curTime = float(stamp)
        if self.first_flag < self.firstflagnum:                  # :857-882
            self.first_flag += 1
            self.prevTime = curTime
            self.t_w_curr = self.init_xyz.copy()
            if self.USE_IMU:
                q = _quat_of_matrix(self.IMU.Rwi @ self.R_il)
                y = pf.R2ypr(pf.q2R(q)) + np.array([self.init_yaw, 0.0, 0.0])
                self.q_w_curr = _normalized(_quat_of_matrix(pf.ypr2R(y)))
            else:
                self.q_w_curr = np.array([0, 0, 0, 1.0])
            self.q_w_curr_f = self.q_w_curr.copy()
            return None
        if self.USE_IMU:                                          # :883-931
            vec = self.getIMUInterval(self.prevTime, curTime)
            if vec is None:
                return None
            delta_q_imu = np.array([0, 0, 0, 1.0])
            for i in range(len(vec)):
                if i == 0:
                    dt = vec[i][0] - self.prevTime
                elif i == len(vec) - 1:
                    dt = curTime - vec[i - 1][0]
                else:
                    dt = vec[i][0] - vec[i - 1][0]
                gyr = vec[i][2]
                delta_q_imu = _normalized(pf.qmul(delta_q_imu, np.array([gyr[0] * dt / 2, gyr[1] * dt / 2, gyr[2] * dt / 2, 1.0])))   # :1418-1422
            self.delta_q_imu = delta_q_imu
            self.para_q = _normalized(delta_q_imu.copy())         # q_last_curr = delta_q_imu; normalize, :929-930
        farme_dt = curTime - self.prevTime                         # :955 (used by the gravity solves only)
        self.prevTime = curTime
        laserCloudFullRes = orc.deskew(laserCloudFullRes, self.para_q, self.para_t)          # adjustDistortion, :958, 1441-1481
        if len(self.laserCloudFullLast) != 0:                      # :961
            if self.submapflag == 0:                               # :963-972
                self.surroundingCloud.append(self.laserCloudFullLast)
                self.surrounding_q.append(np.array([0, 0, 0, 1.0]))   # q_init, :16
                self.surrounding_t.append(np.zeros(3))                # t_init, :18
                self.histoary_pose.append(self.q_w_curr_delta.copy())
                self.laserCloudsubmap = np.concatenate([self.laserCloudsubmap, self.laserCloudFullLast])
            self.submapflag += 1
            FullPointsLessFlat = orc.voxelgrid_filter(np.ascontiguousarray(laserCloudFullRes[:, :4]), self.planeResolution1)     # :976-983
            FullPointsLessFlatlast = orc.voxelgrid_filter(np.ascontiguousarray(self.laserCloudsubmap[:, :4]), self.planeResolution2)   # :985-991
            T2 = np.eye(4, dtype=np.float32)                       # :993-996
            T2[:3, :3] = pf.q2R(self.para_q).astype(np.float32)
            T2[:3, 3] = self.para_t.astype(np.float32)
            vgicp = orc.Registration(num_threads=0)                # :998-1006: the constants are the oracle's defaults
            vgicp.set_target(FullPointsLessFlatlast)
            vgicp.set_source(FullPointsLessFlat)
            T_Drift = np.asarray(vgicp.align(T2), np.float32)      # :1009
            self.vgicp_source = float(vgicp.fitness())             # :1010
            t_drift = T_Drift[:3, 3]                               # Affine3f::translation()
            U, _, Vt = np.linalg.svd(T_Drift[:3, :3].astype(np.float64))     # Affine3f::rotation(): the polar factor (SVD), :1013
            q_drift = _quat_of_matrix((U @ Vt).astype(np.float32), np.float32)   # Quaternionf(q_drift), :1015
            t_last_curr_l = t_drift.astype(np.float64)
            q_last_curr_l = q_drift.astype(np.float64)
            self.para_q, self.para_t = q_last_curr_l.copy(), t_last_curr_l.copy()            # :1017-1023
            # ---- ground, :1031-1087 ----
            ground_norm_cur = _rot(q_last_curr_l, ground_cur[0:3])
            ground_distance_cur = ground_cur[9] + ground_norm_cur @ t_last_curr_l
            ground_erro_1 = np.linalg.norm(self.ground_last[9] * self.ground_last[0:3] - ground_distance_cur * ground_norm_cur)
            ground_erro_2 = abs(self.ground_last[3:6] @ ground_norm_cur)
            d_ypr = pf.R2ypr(pf.q2R(self.delta_q_imu))
            if ground_erro_1 >= 0.02 and ground_erro_2 >= 0.02 and abs(d_ypr[1]) > 0.5:
                self.changegroundflag = 0
                self.gflag = 1
            if self.gflag == 1 and self.changegroundflag < 25:
                self.changegroundflag += 1
                if self.changegroundflag == 25:
                    last_q, pr_erro = None, 1000.0
                    now_ypr = pf.R2ypr(pf.q2R(self.q_w_curr))
                    for h in self.histoary_pose:
                        temp = pf.R2ypr(pf.q2R(h))
                        p_erro, r_erro = temp[1] - now_ypr[1], temp[2] - now_ypr[2]
                        if np.sqrt(p_erro * p_erro + r_erro * r_erro) < pr_erro:
                            pr_erro = np.sqrt(p_erro * p_erro + r_erro * r_erro)
                            last_q = h
                    if pr_erro < 4:
                        self.q_w_curr_delta = last_q.copy()
                        self.gflag = 0
                    else:
                        self.q_w_curr_delta = self.q_w_curr.copy()
                        self.histoary_pose.append(self.q_w_curr_delta.copy())
                        self.gflag = 0
            self.q_w_curr_f = _normalized(pf.qmul(pf.qconj(self.q_w_curr_delta), self.q_w_curr))   # :1086-1087
            # ---- the fusion problem, :1025-1032, 1088-1119, 1188-1193 ----
            use_ground = bool(self.USE_GROUND and self.gflag == 0)
            use_imu = bool(self.USE_IMU and self.imuflag == 1)
            c = dict(q_lidar=q_last_curr_l, t_lidar=t_last_curr_l, fitness=self.vgicp_source, use_ground=use_ground,
                     ground_last=self.ground_last, ground_cur=ground_cur, q_w_curr_f=self.q_w_curr_f, ground_cov=0.2,
                     use_imu=use_imu, q_imu=self.delta_q_imu)
            self.para_q, self.para_t = pf.fuse(c)                  # (without the ground block para_t has no residual: it stays t_last_curr_l, :1099)
            # ---- composition, :1194-1203 ----
            t_last_tmp1 = _rot(self.q_w_curr, self.para_t)
            t_last_tmp2 = _rot(self.q_w_curr, t_last_curr_l)
            t_last_tmp = np.array([t_last_tmp2[0], t_last_tmp2[1], t_last_tmp1[2]])
            self.para_t = _rot(pf.qconj(self.q_w_curr), t_last_tmp)
            self.t_w_curr = self.t_w_curr + _rot(self.q_w_curr, self.para_t)
            self.q_w_curr = _normalized(pf.qmul(self.q_w_curr, self.para_q))
            if self.USE_IMU and self.imuflag == 1:                 # :1206-1214
                ypr_w = pf.R2ypr(pf.q2R(self.q_w_curr))
                ypr_i = pf.R2ypr(self.IMU.Rwi @ self.R_il)
                ypr_w[1] = 0.95 * ypr_w[1] + 0.05 * ypr_i[1]
                ypr_w[2] = 0.95 * ypr_w[2] + 0.05 * ypr_i[2]
                self.q_w_curr = _normalized(_quat_of_matrix(pf.ypr2R(ypr_w)))
            # ---- the local map, :1218-1256 ----
            if self.surroundingCloud:
                ypr_b = pf.R2ypr(pf.q2R(self.surrounding_q[-1]))
                ypr_c = pf.R2ypr(pf.q2R(self.q_w_curr))
                f32 = np.float32
                dx = f32(self.surrounding_t[-1][0] - self.t_w_curr[0])
                dy = f32(self.surrounding_t[-1][1] - self.t_w_curr[1])
                dz = f32(self.surrounding_t[-1][2] - self.t_w_curr[2])
                dyaw, dpitch, droll = f32(ypr_b[0] - ypr_c[0]), f32(ypr_b[1] - ypr_c[1]), f32(ypr_b[2] - ypr_c[2])
                if dyaw > np.pi:
                    dyaw = f32(dyaw - np.pi * 2)
                if dyaw < -np.pi:
                    dyaw = f32(dyaw + np.pi * 2)
                if (abs(droll) > self.keyframeAddingAngle or abs(dpitch) > self.keyframeAddingAngle or abs(dyaw) > self.keyframeAddingAngle or
                        np.sqrt(f32(dx * dx + dy * dy + dz * dz)) > self.keyframeAddingDistance or self.submapflag < self.slipwide - 1):
                    self.surroundingCloud.append(orc.transform_cloud(FullPointsLessFlat, self.q_w_curr, self.t_w_curr))
                    self.surrounding_q.append(self.q_w_curr.copy())
                    self.surrounding_t.append(self.t_w_curr.copy())
            self.laserCloudsubmap = np.zeros((0, 4), np.float32)
            if len(self.surroundingCloud) > self.slipwide:
                self.surroundingCloud.pop(0)
                self.surrounding_q.pop(0)
                self.surrounding_t.pop(0)
            if len(self.surroundingCloud) > 1:
                qc = pf.qconj(self.q_w_curr)
                tc = -1 * _rot(qc, self.t_w_curr)
                for cloud in self.surroundingCloud:
                    self.laserCloudsubmap = np.concatenate([self.laserCloudsubmap, orc.transform_cloud(cloud, qc, tc)])
        self.laserCloudFullLast = laserCloudFullRes               # :1319-1320
        self.ground_last = ground_cur                              # :1322
        self.frameCount += 1
        return self.q_w_curr.copy(), self.t_w_curr.copy()
