"""The frame body of vg_ICP::ICP_thread (RGC_odometer.cpp:848-1256), pinned independently of the product's mirrors: the literal
restatement oracle/py_odometer.py against its committed fixture tests/golden/fx_sequence.npz (tests/golden/gen_sequence.py), and the
Python mirror rgc_slam_amd.odometry.Odometer -- whose orchestration the GPU sequence tests share between both sides -- held to it on the
CPU oracle's stages: same poses, same ground flag, same number of keyframes, sweep by sweep.  CPU only."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import gen_sequence  # noqa: E402


def _fixture():
    return np.load(os.path.join(ROOT, "tests", "golden", "fx_sequence.npz"))


def test_literal_frame_body_reproduces_its_fixture():
    from oracle import py_odometer
    fx = _fixture()
    raws, sweep_stamps, imu = gen_sequence.inputs()
    assert gen_sequence.digest(raws, imu) == str(fx["inputs_sha256"])          # the regenerated inputs are the fixture's
    node = py_odometer.IcpThread(USE_IMU=1, USE_GROUND=1, firstflagnum=2)
    rows = []

    def handle(raw, t_k):
        r = node.handle(raw, t_k)
        rows.append((r is not None, node.q_w_curr.copy(), node.t_w_curr.copy(), node.gflag, len(node.surroundingCloud), len(node.laserCloudsubmap),
                     node.vgicp_source))
        return r
    gen_sequence.feed(node, raws, sweep_stamps, imu, node.imuCallback, handle)
    assert [r[0] for r in rows] == list(fx["produced"]) == [False, False] + [True] * 8
    assert np.abs(np.array([r[1] for r in rows]) - fx["q"]).max() <= 1e-12 and np.abs(np.array([r[2] for r in rows]) - fx["t"]).max() <= 1e-12
    assert [r[3] for r in rows] == list(fx["gflag"]) and [r[4] for r in rows] == list(fx["keyframes"]) and [r[5] for r in rows] == list(fx["submap"])
    assert np.abs(np.array([r[6] for r in rows]) - fx["fitness"]).max() <= 1e-12
    assert fx["keyframes"][-1] == 3 and np.linalg.norm(fx["t"][-1]) > 0.4      # the window is full and the platform moved


def test_python_mirror_follows_the_literal_frame_body():
    from rgc_slam_amd import odometry
    from oracle_backend import OracleBackend
    fx = _fixture()
    raws, sweep_stamps, imu = gen_sequence.inputs()
    od = odometry.Odometer(OracleBackend(), use_imu=True, first_frames=2)
    rows = []

    def handle(raw, t_k):
        r = od.process(raw, t_k)
        rows.append((r is not None, od.q_w_curr.copy(), od.t_w_curr.copy(), od.gflag, len(od.surrounding), len(od.submap), od.fitness))
        return r
    gen_sequence.feed(od, raws, sweep_stamps, imu, od.imu_callback, handle)
    assert [r[0] for r in rows] == list(fx["produced"])
    for i, r in enumerate(rows):
        q, qf = r[1], fx["q"][i]
        assert min(np.abs(q - qf).max(), np.abs(q + qf).max()) <= 1e-9, i     # same stages, same order of operations: rounding only
        assert np.abs(r[2] - fx["t"][i]).max() <= 1e-9, i
        assert r[3] == fx["gflag"][i] and r[4] == fx["keyframes"][i] and r[5] == fx["submap"][i], i
        if fx["produced"][i] and fx["submapflag"][i] > 0:
            assert abs(r[6] - fx["fitness"][i]) <= 1e-9, i
