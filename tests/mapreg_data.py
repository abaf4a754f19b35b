"""Synthetic input of the mapping node's feature registration (f1): corner / surf feature maps in the world frame built
from the front-end's sharp / flat features of a few synthetic VLP-16 scans, plus the features of the two newest frames."""
import numpy as np


def rot_to_quat_xyzw(R):
    w = np.sqrt(max(0.0, 1.0 + R[0, 0] + R[1, 1] + R[2, 2])) / 2.0
    x = (R[2, 1] - R[1, 2]) / (4 * w)
    y = (R[0, 2] - R[2, 0]) / (4 * w)
    z = (R[1, 0] - R[0, 1]) / (4 * w)
    return np.array([x, y, z, w])


def make_case(synth, frontend, n_map_frames=8, seed=None, n_az=1800, voxelgrid=None, corner_leaf=0.4, surf_leaf=0.8):
    """frontend(xyzi) -> dict with 'sharp' and 'flat' (n,5: x,y,z,intensity,normal_x).  Returns dict(corner_map, surf_map (n,4
    float32: x,y,z,pad; leaf-filtered when `voxelgrid(xyzi, leaf)` is given), corner_cur, surf_cur, corner_last, surf_last (n,4: x,y,z,weight), T_cur, T_last (true poses))."""
    seed = synth.SEED if seed is None else seed
    world = synth.make_world(seed=seed)
    poses = synth.make_trajectory(n_map_frames + 2, seed=seed)
    feats = []
    for i, T in enumerate(poses):
        sc = synth.make_scan(world, T, n_az=n_az, seed=seed + 10 + i)
        xyzi = np.concatenate([sc["xyz"], sc["intensity"][:, None]], axis=1).astype(np.float32)
        fe = frontend(xyzi)
        feats.append((fe["sharp"][:, [0, 1, 2, 4]].astype(np.float32), fe["flat"][:, [0, 1, 2, 4]].astype(np.float32)))
    def to_world(f, T):
        o = np.zeros((len(f), 4), np.float32)
        o[:, :3] = (f[:, :3].astype(np.float64) @ T[:3, :3].T + T[:3, 3]).astype(np.float32)
        return o
    corner_map = np.concatenate([to_world(feats[i][0], poses[i]) for i in range(n_map_frames)])
    surf_map = np.concatenate([to_world(feats[i][1], poses[i]) for i in range(n_map_frames)])
    if voxelgrid is not None:  # laserCloudCornerFromMapDS / laserCloudSurfFromMapDS are leaf-filtered in the reference
        corner_map = np.ascontiguousarray(voxelgrid(corner_map, corner_leaf), dtype=np.float32)
        surf_map = np.ascontiguousarray(voxelgrid(surf_map, surf_leaf), dtype=np.float32)
    return dict(corner_map=corner_map, surf_map=surf_map, corner_cur=feats[-1][0], surf_cur=feats[-1][1], corner_last=feats[-2][0],
                surf_last=feats[-2][1], T_cur=poses[-1], T_last=poses[-2])


def poses14(T_cur, T_last):
    return np.concatenate([rot_to_quat_xyzw(T_cur[:3, :3]), T_cur[:3, 3], rot_to_quat_xyzw(T_last[:3, :3]), T_last[:3, 3]])


def perturb(T, rng, ang=0.01, trans=0.05):
    a = rng.normal(0, ang, 3)
    th = np.linalg.norm(a)
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    R = np.eye(3) + (np.sin(th) / th) * K + ((1 - np.cos(th)) / th ** 2) * K @ K if th > 0 else np.eye(3)
    o = T.copy()
    o[:3, :3] = R @ T[:3, :3]
    o[:3, 3] = T[:3, 3] + rng.normal(0, trans, 3)
    return o


def make_ground(T_cur, T_last, height=0.56, p_var=0.2, tilt=(0.01, -0.008)):
    """A Ground_DeltaFactor_goable input consistent with a ground plane at z = -height in the map frame (RGC_mapping.cpp:1326-1331):
    g_last / g_cur are the plane seen from the last / current sensor frame, q_history the last pose's rotation."""
    def plane_in(T):
        R = T[:3, :3]
        n = R.T @ np.array([0.0, 0.0, 1.0])            # map-frame up expressed in the sensor frame
        d = height + T[2, 3]                            # sensor height above the plane
        v1 = np.cross(n, [1.0, 0.0, 0.0]); v1 /= np.linalg.norm(v1)
        v2 = np.cross(n, v1)
        return n, v1, v2, d
    nl, v1, v2, dl = plane_in(T_last)
    nc, _, _, dc = plane_in(T_cur)
    nc = nc + np.array([tilt[0], tilt[1], 0.0]); nc /= np.linalg.norm(nc)   # measurement noise on the current plane
    return dict(last_v1=v1, last_v2=v2, last_norm=nl, last_distance=dl, cur_norm=nc, cur_distance=dc + 0.01,
                q_history=rot_to_quat_xyzw(T_last[:3, :3]), last_q=rot_to_quat_xyzw(T_last[:3, :3]), last_t=T_last[:3, 3].copy(), p_var=p_var)


def make_imu(T_cur, T_last, noise=(0.004, -0.003, 0.002), imu_cov=0.4, pr_var=0.02):
    """The IMU block's inputs (RGC_mapping.cpp:1285-1312) consistent with the true poses: delta_q_imu = q_last^-1 (x) q_cur with a
    small gyro error, pitch / roll targets = those of each pose's rotation (Quaternion2EulerAngle convention) with an offset."""
    def pr(q):
        x, y, z, w = q
        return np.arcsin(np.clip(2 * (w * y - x * z), -1, 1)), np.arctan2(2 * (w * x + y * z), 1 - 2 * (x * x + y * y))
    Rd = T_last[:3, :3].T @ T_cur[:3, :3]
    e = perturb(np.eye(4), np.random.default_rng(5), ang=0.003, trans=0.0)[:3, :3]
    dq = rot_to_quat_xyzw(e @ Rd)
    pc, rc = pr(rot_to_quat_xyzw(T_cur[:3, :3]))
    pl, rl = pr(rot_to_quat_xyzw(T_last[:3, :3]))
    return dict(delta_q=dq, imu_cov=imu_cov, pitch_cur=pc + noise[0], roll_cur=rc + noise[1], pitch_last=pl + noise[2], roll_last=rl - noise[0],
                pr_var=pr_var)
