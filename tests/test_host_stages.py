"""Scalar host stages in librgc_hip.so (B1 IMU delta-q, B7 pose fusion, B8 composition, C9 extraction, R2ypr/ypr2R)
against the independent numpy/scipy restatement oracle/py_fusion.py.  No GPU needed."""
import ctypes as C
import math

import numpy as np
import pytest


@pytest.fixture(scope="module")
def L():
    from rgc_slam_amd import _lib
    return _lib, _lib.load()


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _rand_q(rng, angle):
    w = rng.normal(size=3); w *= angle / np.linalg.norm(w)
    n = np.linalg.norm(w)
    return np.array([*(math.sin(n / 2) / n * w), math.cos(n / 2)])


def _case(rng, use_ground, use_imu, big_imu=False):
    q_l = _rand_q(rng, 0.03)
    t_l = rng.normal(0, 0.1, 3)
    n_last = np.array([0.01, -0.02, 1.0]); n_last /= np.linalg.norm(n_last)
    v1 = np.cross(n_last, [1, 0, 0]); v1 /= np.linalg.norm(v1)
    v2 = np.cross(n_last, v1)
    n_cur = n_last + rng.normal(0, 0.01, 3); n_cur /= np.linalg.norm(n_cur)
    g_last = np.array([*n_last, *v1, *v2, 0.56, 0.02])
    g_cur = np.array([*n_cur, *v1, *v2, 0.56 + rng.normal(0, 0.01), 0.03])
    return dict(q_lidar=q_l, t_lidar=t_l, fitness=float(rng.uniform(0.02, 0.3)), use_ground=use_ground, ground_last=g_last,
                ground_cur=g_cur, q_w_curr_f=_rand_q(rng, 0.05), ground_cov=0.2, use_imu=use_imu,
                q_imu=_rand_q(rng, 0.05 if big_imu else 0.005))


def _fuse_lib(L, c):
    lib, h = L
    fin = lib.FuseIn()
    h.rgc_default_fuse_in(C.byref(fin))
    fin.q_lidar_xyzw[:] = list(c["q_lidar"]); fin.t_lidar[:] = list(c["t_lidar"]); fin.fitness = c["fitness"]
    fin.use_ground = int(c["use_ground"]); fin.ground_last[:] = list(c["ground_last"]); fin.ground_cur[:] = list(c["ground_cur"])
    fin.q_w_curr_f_xyzw[:] = list(c["q_w_curr_f"]); fin.ground_cov = c["ground_cov"]
    fin.use_imu = int(c["use_imu"]); fin.q_imu_xyzw[:] = list(c["q_imu"])
    q, t, it = np.empty(4), np.empty(3), C.c_int(0)
    assert h.rgc_fuse_pose(C.byref(fin), _dp(q), _dp(t), C.byref(it)) == 0
    return q, t, it.value


@pytest.mark.parametrize("use_ground,use_imu,big", [(True, True, False), (True, True, True), (True, False, False),
                                                    (False, True, False), (False, False, False)])
def test_fuse_pose(L, use_ground, use_imu, big):
    from oracle import py_fusion as pf
    rng = np.random.default_rng(11 + 2 * use_ground + use_imu + 4 * big)
    for _ in range(5):
        c = _case(rng, use_ground, use_imu, big)
        q, t, it = _fuse_lib(L, c)
        qo, to = pf.fuse(c)
        if np.dot(q, qo) < 0:
            qo = -qo
        assert np.abs(q - qo).max() < 1e-6 and np.abs(t - to).max() < 1e-6, (q, qo, t, to, it)
        assert abs(np.linalg.norm(q) - 1) < 1e-12 and it <= 6
        if not use_ground:
            assert np.array_equal(t, c["t_lidar"])          # para_t has no residual block (RGC_odometer.cpp:1098-1102)
        if not use_ground and not use_imu:
            assert np.abs(q - c["q_lidar"]).max() < 1e-9    # only the lidar rotation prior: stays where it is


def test_compose_and_ypr(L):
    from oracle import py_fusion as pf
    lib, h = L
    rng = np.random.default_rng(5)
    for use_imu in (0, 1):
        for _ in range(5):
            q_w, q_f = _rand_q(rng, 0.8), _rand_q(rng, 0.03)
            t_w, t_f, t_l = rng.normal(0, 5, 3), rng.normal(0, 0.1, 3), rng.normal(0, 0.1, 3)
            R_imu = pf.q2R(_rand_q(rng, 0.8))
            qo, to, tl = np.empty(4), np.empty(3), np.empty(3)
            Rr = np.ascontiguousarray(R_imu)
            assert h.rgc_compose_pose(_dp(q_w), _dp(t_w), _dp(q_f), _dp(t_f), _dp(t_l), use_imu, _dp(Rr), _dp(qo), _dp(to), _dp(tl)) == 0
            Re, te, tle = pf.compose(q_w, t_w, q_f, t_f, t_l, use_imu, R_imu)
            assert np.abs(pf.q2R(qo) - Re).max() < 1e-12 and np.abs(to - te).max() < 1e-12 and np.abs(tl - tle).max() < 1e-12
    for _ in range(20):
        R = np.ascontiguousarray(pf.q2R(_rand_q(rng, rng.uniform(0.01, 1.2))))
        ypr, R2 = np.empty(3), np.empty(9)
        h.rgc_R2ypr(_dp(R), _dp(ypr)); h.rgc_ypr2R(_dp(ypr), _dp(R2))
        assert np.abs(ypr - pf.R2ypr(R)).max() < 1e-12 and np.abs(R2.reshape(3, 3) - R).max() < 1e-12
    # degrees, order Rz*Ry*Rx: a pure 10 degree yaw
    ypr = np.array([10.0, 0, 0]); R2 = np.empty(9)
    h.rgc_ypr2R(_dp(ypr), _dp(R2))
    assert abs(R2[1] + math.sin(math.radians(10))) < 1e-15 and abs(R2[0] - math.cos(math.radians(10))) < 1e-15


def test_imu_preintegrate(L):
    from oracle import py_fusion as pf
    lib, h = L
    rng = np.random.default_rng(9)
    n = 21
    stamps = np.ascontiguousarray(100.0 + 0.005 * np.arange(1, n + 1) + rng.uniform(-1e-4, 1e-4, n))
    gyr = np.ascontiguousarray(rng.normal(0, 0.3, (n, 3)))
    acc = np.ascontiguousarray(rng.normal(0, 0.5, (n, 3)) + [0, 0, 9.8])
    prev, cur = 100.0, float(stamps[-1] - 0.002)
    dq, dq2, dp_, dv = np.empty(4), np.empty(4), np.empty(3), np.empty(3)
    assert h.rgc_imu_preintegrate(_dp(stamps), _dp(gyr), _dp(acc), n, prev, cur, _dp(dq), _dp(dq2), _dp(dp_), _dp(dv)) == 0
    assert np.abs(dq - pf.imu_delta_q(stamps, gyr, prev, cur)).max() < 1e-14
    assert abs(np.linalg.norm(dq2) - 1) < 1e-12 and np.abs(dv - [0, 0, 9.8 * (cur - prev)]).max() < 0.3
    # constant rate about z: delta-q is a rotation by rate * T about z (first-order quaternion integration)
    g2 = np.ascontiguousarray(np.tile([0, 0, 0.2], (n, 1)))
    assert h.rgc_imu_preintegrate(_dp(stamps), _dp(g2), None, n, prev, cur, _dp(dq), None, None, None) == 0
    ang = 2 * math.atan2(dq[2], dq[3])
    assert abs(ang - 0.2 * (cur - prev)) < 1e-6


def test_extract_pose(L):
    from oracle import py_fusion as pf
    lib, h = L
    rng = np.random.default_rng(2)
    for _ in range(10):
        q = _rand_q(rng, rng.uniform(0.001, 0.5))
        T = np.eye(4, dtype=np.float32)
        T[:3, :3] = pf.q2R(q).astype(np.float32)
        T[:3, 3] = rng.normal(0, 1, 3).astype(np.float32)
        qo, to = np.empty(4), np.empty(3)
        assert h.rgc_extract_pose(T.ctypes.data_as(C.POINTER(C.c_float)), _dp(qo), _dp(to)) == 0
        if np.dot(qo, q) < 0:
            qo = -qo
        assert np.abs(qo - q).max() < 2e-7 and np.array_equal(to, T[:3, 3].astype(np.float64))


def test_imu_filter_vs_restatement(L):
    """vg_ICP::imu_callback + ComplementaryFilter (RGC_odometer.cpp:444-486, 545-625): the library against oracle/py_fusion.ImuFilter on a
    synthetic stream -- a resting lead-in (first 100 messages dropped, fast phase of 300), then a turning platform; and a known answer:
    at rest on a tilted platform the filter settles on the tilt the accelerometer shows."""
    import rgc_slam_amd.synth as synth
    from oracle import py_fusion as pf
    lib, h = L
    poses = synth.make_trajectory(12, seed=synth.SEED)
    stamps, acc, gyr = synth.make_imu(poses, seed=synth.SEED)
    f = lib.ImuFilter(); h.rgc_imu_filter_init(C.byref(f))
    ref = pf.ImuFilter()
    ao, go = np.empty(3), np.empty(3)
    worst, accepted = 0.0, 0
    for t, a, g in zip(stamps, acc, gyr):
        rc = h.rgc_imu_filter_push(C.byref(f), float(t), _dp(np.ascontiguousarray(a)), _dp(np.ascontiguousarray(g)), _dp(ao), _dp(go))
        r = ref.push(float(t), a, g)
        assert (rc == 1) == (r is not None)
        if rc == 1:
            accepted += 1
            assert np.abs(ao - r[0]).max() < 1e-15 and np.abs(go - r[1]).max() < 1e-15
            worst = max(worst, float(np.abs(np.array(f.Rwi[:]).reshape(3, 3) - ref.Rwi).max()))
    assert accepted == len(stamps) - 100 and f.count == accepted
    assert worst < 1e-12, worst
    # known answer: 4 s at rest, pitched by 3 degrees and rolled by -2 (accelerometer = R^T g): the filter reports that attitude
    R = pf.ypr2R(np.array([0.0, 3.0, -2.0]))
    f2 = lib.ImuFilter(); h.rgc_imu_filter_init(C.byref(f2))
    a = R.T @ np.array([0, 0, 9.81]) + np.array(f2.ba[:]); g = np.array(f2.bg[:])
    for j in range(800):
        h.rgc_imu_filter_push(C.byref(f2), 0.005 * j, _dp(a), _dp(g), None, None)
    ypr = np.empty(3); h.rgc_R2ypr(_dp(np.array(f2.Rwi[:])), _dp(ypr))
    assert abs(ypr[1] - 3.0) < 0.05 and abs(ypr[2] + 2.0) < 0.05 and abs(ypr[0]) < 1e-9, ypr


def test_ground_gate_vs_restatement(L):
    """The ground-change detector (RGC_odometer.cpp:1034-1087): a plane mismatch while the IMU pitches switches the ground factor off
    for 25 frames, then the plane is re-associated with a remembered attitude (or remembered).  Library vs oracle/py_fusion.GroundGate
    over a scripted drive: flat, a ramp (new plane remembered), flat again (the first plane is found in the history)."""
    from oracle import py_fusion as pf
    lib, h = L
    rng = np.random.default_rng(5)
    g = lib.GroundGate(); h.rgc_ground_gate_init(C.byref(g))
    ref = pf.GroundGate()
    h.rgc_ground_gate_remember(C.byref(g)); ref.remember()

    def plane(pitch_deg):
        n = pf.ypr2R(np.array([0.0, pitch_deg, 0.0])) @ np.array([0, 0, 1.0])
        v1 = np.cross(n, [0, 1.0, 0]); v1 /= np.linalg.norm(v1)
        return np.array([*n, *v1, *np.cross(n, v1), 0.56, 0.01])
    flags, q_w = [], np.array([0, 0, 0, 1.0])
    for k in range(90):
        on_ramp = 20 <= k < 55
        change = k in (20, 55)                       # the frames in which the plane under the sensor changes
        g_last, g_cur = plane(6.0 if (on_ramp and not change) or k == 55 else 0.0), plane(6.0 if on_ramp else 0.0)
        dq_imu = _rand_q(rng, 0.0005) if not change else np.array([0, math.sin(math.radians(0.6)), 0, math.cos(math.radians(0.6))])
        q_l, t_l = _rand_q(rng, 0.002), np.array([0.1, 0.0, 0.0])
        q_w = pf.qmul(q_w, pf.qmul(np.array([0, math.sin(math.radians(3.0 if k == 20 else (-3.0 if k == 55 else 0.0))), 0,
                                             math.cos(math.radians(3.0 if k == 20 else (-3.0 if k == 55 else 0.0)))]), q_l))
        q_w /= np.linalg.norm(q_w)
        qf = np.empty(4)
        rc = h.rgc_ground_gate_step(C.byref(g), _dp(g_last), _dp(g_cur), _dp(q_l), _dp(t_l), _dp(dq_imu), _dp(q_w), _dp(qf))
        rf, qfr = ref.step(g_last, g_cur, q_l, t_l, dq_imu, q_w)
        assert rc == rf and np.abs(qf - qfr).max() < 1e-14
        flags.append(rc)
    assert flags[19] == 0 and all(f == 1 for f in flags[20:44]) and flags[44] == 0      # 25-frame hold-off after the ramp's foot
    assert all(f == 1 for f in flags[55:79]) and flags[79] == 0                          # and after its top
    assert g.n_history == len(ref.history) == 2                                          # flat, ramp; the second flat was FOUND, not added
    assert np.abs(np.array(g.q_w_curr_delta[:]) - ref.q_delta).max() < 1e-14
