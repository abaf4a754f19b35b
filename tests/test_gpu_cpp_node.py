"""The C++ host side of the path (rgc-slam_amd/cpp/odometry_node.hpp: PointCloud2 bytes in, odometry pose + ground message out)
against the Python mirrors of the same frame body driven through the same C-ABI, and through them against the CPU oracle
(tests/test_gpu_sequence.py, tests/test_gpu_rolling_map.py).  -m gpu."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("cppnode") / "test_odometry_node")
    subprocess.check_call(["g++", "-std=c++14", "-O2", "-Wall", "-pthread", os.path.join(ROOT, "tests", "cpp", "test_odometry_node.cpp"), "-o", out,
                           "-L", os.path.join(ROOT, "rgc-slam_amd"), "-lrgc_hip", "-Wl,-rpath," + os.path.join(ROOT, "rgc-slam_amd")])
    return out


@pytest.fixture(scope="module")
def sweeps(tmp_path_factory):
    import rgc_slam_amd.synth as synth
    world = synth.make_world(half_extent=45.0, seed=synth.SEED)
    poses = synth.make_trajectory(9, seed=synth.SEED)
    raws = []
    for k in range(8):
        sc = synth.make_scan(world, poses[k], n_az=1200, seed=synth.SEED + 50 + k, T_ws_end=poses[k + 1])
        raws.append(np.concatenate([sc["xyz"], sc["intensity"][:, None]], axis=1).astype(np.float32))
    path = str(tmp_path_factory.mktemp("sweeps") / "sweeps.bin")
    dt = np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("intensity", "<f4"), ("ring", "<u2"), ("time", "<f4")])   # packed, 22 bytes
    assert dt.itemsize == 22
    with open(path, "wb") as f:
        f.write(np.int32(len(raws)).tobytes())
        for r in raws:
            rec = np.zeros(len(r), dt)
            rec["x"], rec["y"], rec["z"], rec["intensity"] = r[:, 0], r[:, 1], r[:, 2], r[:, 3]
            rec["ring"] = 7
            f.write(np.int32(len(r)).tobytes()); f.write(rec.tobytes())
    return raws, path


def _run(exe, path, resident, as_message, rebase=None, chain=False, pipeline=False, imu=None, first_frames=0):
    args = [exe, path, str(int(resident)), str(int(as_message)), str(rebase if rebase is not None else 50.0), str(int(chain)), str(int(pipeline))]
    if imu is not None:
        args += [imu, str(first_frames)]
    out = subprocess.run(args, capture_output=True, text=True, timeout=600).stdout
    assert "EXCEPTION" not in out, out
    poses, ground = [], []
    for line in out.splitlines():
        w = line.split()
        if w[0] == "pose":
            poses.append([float(x) for x in w[2:9]]); ground.append((int(w[10]), float(w[11]), float(w[12])))
    summary = dict(zip(out.splitlines()[-1].split()[1::2], out.splitlines()[-1].split()[2::2]))
    return np.array(poses), ground, summary


@pytest.mark.parametrize("resident", [False, True])
def test_cpp_node_matches_python_frame_body(exe, sweeps, resident):
    from rgc_slam_amd import odometry
    raws, path = sweeps
    hb = odometry.HipBackend(0)
    od = (odometry.RollingOdometer if resident else odometry.Odometer)(hb)
    if resident:
        od.rebase_distance = 0.5
    ref = np.array([np.concatenate(od.process(r)) for r in raws])
    hb.close()
    for as_message in (False, True):     # the converted cloud and the raw PointCloud2 bytes give the same poses
        poses, ground, summary = _run(exe, path, resident, as_message, 0.5 if resident else None)
        assert len(poses) == len(raws) == int(summary["frames"])
        # same C-ABI calls on the same inputs; the scalar host arithmetic (quaternion products) may differ in the last bits
        assert np.abs(poses - ref).max() < 1e-9, np.abs(poses - ref).max()
        assert all(g[0] == 1 for g in ground) and all(abs(abs(g[2]) - 0.56) < 0.1 for g in ground)   # ground plane: normal z ~ 1, distance ~ laderH
        assert int(summary["sharp"]) > 100 and int(summary["flat"]) > 100
    assert np.linalg.norm(ref[-1, 4:7]) > 0.3


def test_cpp_node_device_chain(exe, sweeps):
    """the sweep stays on the device between unpack, front-end, de-skew, VoxelGrid and setInputSource / keyframe insert: the same
    kernels on the same data, so the poses equal the host-staged run exactly"""
    raws, path = sweeps
    for as_message in (True, False):
        a, ga, sa = _run(exe, path, True, as_message, 0.5, chain=False)
        b, gb, sb = _run(exe, path, True, as_message, 0.5, chain=True)
        assert np.array_equal(a, b) and ga == gb
        assert sa["keyframes"] == sb["keyframes"] and sa["sharp"] == sb["sharp"] and sa["flat"] == sb["flat"]
    # (reference semantics on the device: test_cpp_node_reference_semantics_on_the_device)


def test_cpp_replay_pipeline(exe, sweeps):
    """front-end of sweep k+1 (own context, stream and host thread) overlapped with the frame body of sweep k: the stages see the same
    data in the same order, so poses and ground messages equal the unpipelined node's exactly"""
    raws, path = sweeps
    a, ga, sa = _run(exe, path, True, True, 0.5, chain=True)
    for _ in range(2):
        b, gb, sb = _run(exe, path, True, True, 0.5, chain=True, pipeline=True)
        assert np.array_equal(a, b) and ga == gb and sa["frames"] == sb["frames"] and sa["keyframes"] == sb["keyframes"]


def test_cpp_node_with_imu(exe, tmp_path):
    """USE_IMU = 1 (launch/run.launch:18) in the C++ node: imuCallback -> attitude filter / gyro guess / IMU factor / gravity blend /
    ground-change detector, the first sweeps initialising the pose -- same poses as the Python mirror of the frame body, which
    tests/test_gpu_sequence.py::test_sequence_with_imu holds against the CPU oracle"""
    import rgc_slam_amd.synth as synth
    from rgc_slam_amd import odometry
    world = synth.make_world(half_extent=45.0, seed=synth.SEED)
    poses = synth.make_trajectory(9, seed=synth.SEED + 2)
    stamps, acc, gyr = synth.make_imu(poses, seed=synth.SEED + 2)
    raws = []
    for k in range(8):
        sc = synth.make_scan(world, poses[k], n_az=1200, seed=synth.SEED + 70 + k, T_ws_end=poses[k + 1])
        raws.append(np.concatenate([sc["xyz"], sc["intensity"][:, None]], axis=1).astype(np.float32))
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
    hb = odometry.HipBackend(0)
    od = odometry.Odometer(hb, use_imu=True, first_frames=2)
    ref, j = [], 0
    for k, raw in enumerate(raws):
        t_k = 0.1 * (k + 1)
        while j < len(stamps) and stamps[j] <= t_k + 0.011:
            od.imu_callback(stamps[j], acc[j], gyr[j]); j += 1
        od.process(raw, t_k)
        ref.append(np.concatenate([od.q_w_curr, od.t_w_curr]))
    hb.close()
    ref = np.array(ref)
    poses_cpp, ground, summary = _run(exe, path, False, True, imu=ipath, first_frames=2)
    assert len(poses_cpp) == len(raws) and int(summary["frames"]) == len(raws) - 2      # two sweeps only initialised the pose
    assert np.abs(poses_cpp - ref).max() < 1e-9, np.abs(poses_cpp - ref).max()
    assert np.linalg.norm(ref[-1, 4:7]) > 0.2


def test_cpp_node_against_the_literal_frame_body(exe, tmp_path):
    """rgc::OdometryNode (the frame body in the reference's language) on the ten sweeps + IMU stream of tests/golden/fx_sequence.npz: the
    poses of oracle/py_odometer.py -- the line by line restatement of vg_ICP::ICP_thread on the CPU oracle's stages -- sweep by sweep,
    pose deltas within 1e-4 m / 1e-4 rad."""
    import os, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tests", "golden"))
    import gen_sequence
    fx = np.load(os.path.join(root, "tests", "golden", "fx_sequence.npz"))
    raws, sweep_stamps, (stamps, acc, gyr) = gen_sequence.inputs()
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
    poses_cpp, ground, summary = _run(exe, path, False, True, imu=ipath, first_frames=2)
    assert len(poses_cpp) == len(raws) and int(summary["frames"]) == int(np.sum(fx["produced"]))
    q, t = poses_cpp[:, :4], poses_cpp[:, 4:7]
    ang = lambda a, b: 2 * np.arccos(min(1.0, abs(float(np.dot(a, b)))))
    for i in range(1, len(raws)):
        assert np.abs((t[i] - t[i - 1]) - (fx["t"][i] - fx["t"][i - 1])).max() <= 1e-4, i
        assert abs(ang(q[i], q[i - 1]) - ang(fx["q"][i], fx["q"][i - 1])) <= 1e-4, i
    assert np.abs(t - fx["t"]).max() <= 5e-4 and max(ang(a, b) for a, b in zip(q, fx["q"])) <= 5e-4


def test_cpp_node_reference_semantics_on_the_device(exe, sweeps):
    """device_chain without the resident map: the reference's keyframe window, its re-framing and both leaf filters stay on the device --
    the same poses, to the last bit, as the host-staged reference-semantics mode (which the frame-body tests hold against the oracle)."""
    raws, path = sweeps
    host, _, s0 = _run(exe, path, False, True)
    dev, _, s1 = _run(exe, path, False, True, chain=True)
    assert len(host) == len(dev) == len(raws) and s0["keyframes"] == s1["keyframes"]
    assert np.array_equal(host, dev)


@pytest.mark.parametrize("resident,chain", [(False, False), (False, True), (True, True)])
def test_cpp_node_with_the_lazy_target(exe, sweeps, monkeypatch, resident, chain):
    """rgc::OdometryNode::Options::lazy_target_margin (rgc_set_target_lazy: the map's covariances and voxels only where the sweep can look):
    the poses and ground messages of the node without it, bit for bit, in the reference's map semantics (host-staged and on the device)
    and with the resident map."""
    _, path = sweeps
    ref = _run(exe, path, resident, True, chain=chain)
    monkeypatch.setenv("RGC_NODE_LAZY_MARGIN", "2")
    got = _run(exe, path, resident, True, chain=chain)
    assert np.array_equal(ref[0], got[0]) and ref[1] == got[1]
    assert ref[2]["keyframes"] == got[2]["keyframes"]
