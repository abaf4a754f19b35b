"""BASELINE config 2 stand-in (the example bag is not reachable offline): a synthetic VLP-16 sequence through the whole
frame body -- front-end, de-skew, VoxelGrid 0.2/0.3, FastVGICP, fitness, ground-constrained pose fusion, sliding
3-keyframe sub-map -- on the GPU vs the same frame body driven by the CPU oracle.  Per-frame pose deltas must agree to
1e-4 m / 1e-4 rad (north_star).  -m gpu."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _angle(qa, qb):
    d = abs(float(np.dot(qa, qb)))
    return 2 * np.arccos(min(1.0, d))


def test_sequence_with_ground_constraint():
    import rgc_slam_amd.synth as synth
    from rgc_slam_amd import odometry
    from oracle_backend import OracleBackend
    world = synth.make_world(half_extent=45.0, seed=synth.SEED)
    poses = synth.make_trajectory(9, seed=synth.SEED)
    raws = []
    for k in range(8):   # sweeps with motion distortion: pose interpolated between consecutive trajectory poses
        sc = synth.make_scan(world, poses[k], n_az=1200, seed=synth.SEED + 50 + k, T_ws_end=poses[k + 1])
        raws.append(np.concatenate([sc["xyz"], sc["intensity"][:, None]], axis=1).astype(np.float32))
    hb = odometry.HipBackend(0)
    og, oc = odometry.Odometer(hb), odometry.Odometer(OracleBackend())
    prev_g, prev_c = (np.array([0, 0, 0, 1.0]), np.zeros(3)), (np.array([0, 0, 0, 1.0]), np.zeros(3))
    worst_t = worst_r = 0.0
    for k, raw in enumerate(raws):
        qg, tg = og.process(raw)
        qc, tc = oc.process(raw)
        # per-frame pose delta of each path
        dtg, dtc = tg - prev_g[1], tc - prev_c[1]
        worst_t = max(worst_t, float(np.abs(dtg - dtc).max()))
        worst_r = max(worst_r, abs(_angle(qg, prev_g[0]) - _angle(qc, prev_c[0])), _angle(qg, qc) if k < 3 else 0.0)
        prev_g, prev_c = (qg, tg), (qc, tc)
    hb.close()
    assert og.frames == 8 and np.linalg.norm(og.t_w_curr) > 0.3               # the platform really moved
    assert worst_t <= 1e-4 and worst_r <= 1e-4, (worst_t, worst_r)
    # and the estimate follows the true motion (sensor frame of pose 8 vs pose 1 ... coarse sanity, not parity)
    true = np.linalg.inv(poses[1]) @ poses[8]
    assert np.linalg.norm(og.t_w_curr - true[:3, 3]) < 0.5


def test_sequence_with_imu():
    """The launch file's default, USE_IMU = 1 (launch/run.launch:18): imu_callback -> attitude filter + gyro pre-integration as the
    registration's guess (RGC_odometer.cpp:883-931, 993-996), the IMU factor of the fusion (:1104-1119), the gravity blend
    (:1206-1214), the pose initialised from the filter's attitude during the first sweeps (:857-882) and the ground-change detector
    (:1034-1087) -- GPU frame body vs the same frame body on the CPU oracle, per-frame pose deltas within 1e-4 m / 1e-4 rad."""
    import rgc_slam_amd.synth as synth
    from rgc_slam_amd import odometry
    from oracle_backend import OracleBackend
    world = synth.make_world(half_extent=45.0, seed=synth.SEED)
    poses = synth.make_trajectory(11, seed=synth.SEED + 2)
    stamps, acc, gyr = synth.make_imu(poses, seed=synth.SEED + 2)
    raws = []
    for k in range(10):
        sc = synth.make_scan(world, poses[k], n_az=1200, seed=synth.SEED + 70 + k, T_ws_end=poses[k + 1])
        raws.append(np.concatenate([sc["xyz"], sc["intensity"][:, None]], axis=1).astype(np.float32))
    hb = odometry.HipBackend(0)
    og = odometry.Odometer(hb, use_imu=True, first_frames=2)
    oc = odometry.Odometer(OracleBackend(), use_imu=True, first_frames=2)
    j = 0
    prev_g = prev_c = None
    worst_t = worst_r = 0.0
    n_done = 0
    for k, raw in enumerate(raws):
        t_k = 0.1 * (k + 1)                                   # the sweep ends at pose k + 1
        while j < len(stamps) and stamps[j] <= t_k + 0.011:   # the IMU messages that arrived before the cloud (one past its stamp, :1405-1406)
            og.imu_callback(stamps[j], acc[j], gyr[j]); oc.imu_callback(stamps[j], acc[j], gyr[j]); j += 1
        rg, rc = og.process(raw, t_k), oc.process(raw, t_k)
        assert (rg is None) == (rc is None) == (k < 2)
        if rg is None:
            assert np.abs(og.q_w_curr - oc.q_w_curr).max() < 1e-12          # pose initialised from IMU.Rwi * R_il
            continue
        (qg, tg), (qc, tc) = rg, rc
        if prev_g is not None:
            worst_t = max(worst_t, float(np.abs((tg - prev_g[1]) - (tc - prev_c[1])).max()))
            worst_r = max(worst_r, abs(_angle(qg, prev_g[0]) - _angle(qc, prev_c[0])))
        worst_r = max(worst_r, _angle(qg, qc) if n_done < 3 else 0.0)
        prev_g, prev_c = (qg, tg), (qc, tc)
        n_done += 1
    hb.close()
    assert n_done == 8 and og.delta_q_imu is not None
    assert worst_t <= 1e-4 and worst_r <= 1e-4, (worst_t, worst_r)
    # the gyro's pre-integrated rotation of the last sweep is the true one to within the sensor noise (it is the registration's guess)
    true_dq = poses[9][:3, :3].T @ poses[10][:3, :3]
    Rg = odometry._q2R(og.delta_q_imu)
    assert np.abs(Rg - true_dq).max() < 2e-3
    # and the estimate follows the true motion from the first processed sweep on (coarse sanity, not parity)
    true = np.linalg.inv(poses[3]) @ poses[10]
    assert np.linalg.norm(og.t_w_curr - true[:3, 3]) < 0.6


def test_sequence_against_the_literal_frame_body():
    """The independent pin of the frame body: tests/golden/fx_sequence.npz is what oracle/py_odometer.py -- a line by
    line restatement of vg_ICP::ICP_thread (RGC_odometer.cpp:848-1256, USE_IMU = 1, USE_GROUND = 1) that shares no code with the mirrors --
    produced on the CPU oracle's stages.  The Python mirror on the HIP library follows it sweep by sweep: pose deltas within 1e-4 m /
    1e-4 rad, the same ground flag, the same number of keyframes in the window, the same sub-map size."""
    import os, sys
    from rgc_slam_amd import odometry
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tests", "golden"))
    import gen_sequence
    fx = np.load(os.path.join(root, "tests", "golden", "fx_sequence.npz"))
    raws, sweep_stamps, imu = gen_sequence.inputs()
    assert gen_sequence.digest(raws, imu) == str(fx["inputs_sha256"])
    hb = odometry.HipBackend(0)
    od = odometry.Odometer(hb, use_imu=True, first_frames=2)
    rows = []

    def handle(raw, t_k):
        r = od.process(raw, t_k)
        rows.append((r is not None, od.q_w_curr.copy(), od.t_w_curr.copy(), od.gflag, len(od.surrounding), len(od.submap)))
        return r
    gen_sequence.feed(od, raws, sweep_stamps, imu, od.imu_callback, handle)
    hb.close()
    assert [r[0] for r in rows] == list(fx["produced"])
    worst_t = worst_r = 0.0
    for i in range(1, len(rows)):
        dq_g, dq_f = _angle(rows[i][1], rows[i - 1][1]), _angle(fx["q"][i], fx["q"][i - 1])
        worst_t = max(worst_t, float(np.abs((rows[i][2] - rows[i - 1][2]) - (fx["t"][i] - fx["t"][i - 1])).max()))
        worst_r = max(worst_r, abs(dq_g - dq_f))
    assert worst_t <= 1e-4 and worst_r <= 1e-4, (worst_t, worst_r)
    assert max(_angle(r[1], q) for r, q in zip(rows, fx["q"])) <= 5e-4 and max(np.abs(r[2] - t).max() for r, t in zip(rows, fx["t"])) <= 5e-4
    assert [r[3] for r in rows] == list(fx["gflag"]) and [r[4] for r in rows] == list(fx["keyframes"])
    assert [r[5] for r in rows] == list(fx["submap"])     # the leaf filters' output sizes: the same clouds went in


def c2_standin(n_sweeps=200, slope=0.06, n_az=1200):
    """BASELINE.md's c2 stand-in at its stated length: `n_sweeps` synthetic VLP-16 sweeps with motion distortion along a trajectory that
    climbs a ramp (the ground CHANGES under the vehicle: the odometer's ground-change detector, RGC_odometer.cpp:1034-1087, trips there)
    and a 200 Hz IMU stream.  -> (raw sweeps, stamps, (imu stamps, acc, gyr), true poses)"""
    import math
    import rgc_slam_amd.synth as synth
    world = synth.make_world(half_extent=45.0, seed=synth.SEED)
    base = synth.make_trajectory(n_sweeps + 1, seed=synth.SEED + 2)
    x0 = base[min(25, n_sweeps // 2)][0, 3]
    world.ramp = (x0, x0 + 4.0, slope)
    poses = []
    for P in base:
        Q = P.copy()
        Q[2, 3] += float(world.ground_height(Q[0, 3])) - world.ground_z
        if world.ramp[0] <= Q[0, 3] <= world.ramp[1]:
            Q[:3, :3] = Q[:3, :3] @ synth.rot_zyx(0.0, -math.atan(slope), 0.0)      # nose up by the ramp's angle
        poses.append(Q)
    imu = synth.make_imu(poses, seed=synth.SEED + 2)
    raws = []
    for k in range(n_sweeps):
        sc = synth.make_scan(world, poses[k], n_az=n_az, seed=synth.SEED + 70 + k, T_ws_end=poses[k + 1])
        raws.append(np.concatenate([sc["xyz"], sc["intensity"][:, None]], axis=1).astype(np.float32))
    return raws, [0.1 * (k + 1) for k in range(n_sweeps)], imu, poses


def test_c2_standin_200_sweeps(tmp_path):
    """BASELINE.md section 3, config 2 at its stated size: 200 sweeps through the whole frame body (front-end, IMU guess, de-skew, leaf
    filters, 3-keyframe sub-map with turnover, FastVGICP, ground gate, fusion, composition), USE_IMU = 1 and USE_GROUND = 1 --
      (a) the Python mirror on the HIP library against the SAME frame body on the CPU oracle, sweep by sweep: pose deltas within
          1e-4 m / 1e-4 rad, the same ground flag every sweep (the gate trips on the ramp and recovers 25 sweeps later), the same keyframes;
      (b) rgc::OdometryNode (C++: PointCloud2 bytes + IMU samples in, odometry out) on the same sweeps against (a)'s HIP poses to 1e-9."""
    import os
    import subprocess
    from rgc_slam_amd import odometry
    from oracle_backend import OracleBackend
    raws, sweep_stamps, (stamps, acc, gyr), poses = c2_standin()
    hb = odometry.HipBackend(0)
    og = odometry.Odometer(hb, use_imu=True, first_frames=2)
    oc = odometry.Odometer(OracleBackend(), use_imu=True, first_frames=2)
    j = 0
    prev_g = prev_c = None
    worst_t = worst_r = 0.0
    flags, turnover, ref = [], 0, []
    newest = None
    for k, raw in enumerate(raws):
        t_k = sweep_stamps[k]
        while j < len(stamps) and stamps[j] <= t_k + 0.011:
            og.imu_callback(stamps[j], acc[j], gyr[j]); oc.imu_callback(stamps[j], acc[j], gyr[j]); j += 1
        rg, rc = og.process(raw, t_k), oc.process(raw, t_k)
        ref.append(np.concatenate([og.q_w_curr, og.t_w_curr]))
        assert (rg is None) == (rc is None) == (k < 2)
        if rg is None:
            continue
        (qg, tg), (qc, tc) = rg, rc
        if prev_g is not None:
            worst_t = max(worst_t, float(np.abs((tg - prev_g[1]) - (tc - prev_c[1])).max()))
            worst_r = max(worst_r, abs(_angle(qg, prev_g[0]) - _angle(qc, prev_c[0])))
        prev_g, prev_c = (qg, tg), (qc, tc)
        assert og.gflag == oc.gflag and len(og.surrounding) == len(oc.surrounding), k
        flags.append(og.gflag)
        if og.surrounding_t and (newest is None or not np.array_equal(newest, og.surrounding_t[-1])):
            turnover += newest is not None
            newest = og.surrounding_t[-1].copy()
    hb.close()
    assert worst_t <= 1e-4 and worst_r <= 1e-4, (worst_t, worst_r)
    trace = "".join(map(str, flags))
    assert flags[0] == 0 and sum(flags) >= 25 and "1" * 24 + "0" in trace, trace       # the gate trips on the ramp, holds the ground factor off for 25 sweeps, and lets it back in
    assert turnover >= 30 and len(og.surrounding) == 3                                                 # keyframes came and went the whole way
    true = np.linalg.inv(poses[3]) @ poses[len(raws)]
    assert np.linalg.norm(og.t_w_curr - true[:3, 3]) < 0.05 * np.linalg.norm(true[:3, 3]) + 0.5       # (coarse sanity, not parity)
    # ---- (b) the C++ node on the same sweeps ----
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "test_odometry_node")
    subprocess.check_call(["g++", "-std=c++14", "-O2", "-pthread", os.path.join(root, "tests", "cpp", "test_odometry_node.cpp"), "-o", exe,
                           "-L", os.path.join(root, "rgc-slam_amd"), "-lrgc_hip", "-Wl,-rpath," + os.path.join(root, "rgc-slam_amd")])
    path, ipath = str(tmp_path / "sweeps.bin"), str(tmp_path / "imu.bin")
    dt = np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("intensity", "<f4"), ("ring", "<u2"), ("time", "<f4")])
    with open(path, "wb") as f:
        f.write(np.int32(len(raws)).tobytes())
        for r in raws:
            rec = np.zeros(len(r), dt)
            rec["x"], rec["y"], rec["z"], rec["intensity"] = r[:, 0], r[:, 1], r[:, 2], r[:, 3]
            f.write(np.int32(len(r)).tobytes()); f.write(rec.tobytes())
    with open(ipath, "wb") as f:
        f.write(np.int32(len(stamps)).tobytes())
        f.write(np.concatenate([stamps[:, None], acc, gyr], axis=1).astype("<f8").tobytes())
    out = subprocess.run([exe, path, "0", "1", "50.0", "0", "0", ipath, "2"], capture_output=True, text=True, timeout=900).stdout
    assert "EXCEPTION" not in out, out[-2000:]
    cpp = np.array([[float(x) for x in l.split()[2:9]] for l in out.splitlines() if l.startswith("pose")])
    ref = np.array(ref)
    # the same C-ABI calls on the same inputs; the scalar host arithmetic in between (quaternion products, the fusion solve) is written twice,
    # C++ and Python: last-bit differences per sweep, carried along the 200 poses they accumulate into (measured: 2e-8; two orders under the parity bar either way)
    assert cpp.shape == ref.shape, (cpp.shape, ref.shape)
    assert np.abs(np.diff(cpp, axis=0) - np.diff(ref, axis=0)).max() < 1e-6 and np.abs(cpp - ref).max() < 1e-6, \
        (np.abs(np.diff(cpp, axis=0) - np.diff(ref, axis=0)).max(), np.abs(cpp - ref).max())
