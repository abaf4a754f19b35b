"""BASELINE.json's configurations at FULL size on the MI355X (-m gpu): c-main 30 k vs 1 M, c3 130 k (HDL-64) vs 5 M,
c5 250 k (two interleaved HDL-64 patterns) vs 20 M with an IMU-like prior.

At these sizes the checks are (i) the CPU oracle itself on all host cores of the GPU box (it needs ~0.3 s per million map
points there) with the path's tolerance -- pose delta <= 1e-4 m / 1e-4 rad -- and (ii) properties that do not depend on
size: every regularised covariance has trace 2.001; the voxel table partitions the map (sum of counts = N_t, one voxel
per occupied cell of floor(x/res - 0.5), counted independently with numpy); the known motion is recovered; aligning again
from the result does not move (idempotence); the fitness is the mean squared 1-NN distance (checked on a sample)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _rot_angle(Ra, Rb):
    R = Ra.astype(np.float64) @ Rb.astype(np.float64).T
    w = 0.5 * np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    return float(np.arcsin(min(1.0, np.linalg.norm(w))))


@pytest.fixture(scope="module")
def synth():
    import rgc_slam_amd.synth as s
    return s


@pytest.fixture(scope="module")
def map5m(synth):
    world, tgt = synth.make_world_and_map(5_000_000, seed=synth.SEED + 7)
    return world, tgt


def _properties(v, tgt, res=1.0):
    n_t = len(tgt)
    vm = v.getVoxels()
    assert int(vm["num"].sum()) == n_t
    c = np.floor(tgt.astype(np.float32).astype(np.float64) / res - 0.5).astype(np.int64)
    c -= c.min(0)
    key = (c[:, 2] * (c[:, 1].max() + 1) + c[:, 1]) * (c[:, 0].max() + 1) + c[:, 0]
    assert len(vm["num"]) == len(np.unique(key)) == v.stats()["n_voxels"]
    # C = I - 0.999 n n^T with |n| = 1: trace 2.001, on a strided sample of the map and on every source point
    ct = v.getTargetCovariances()[:: max(1, n_t // 200000)]
    assert np.abs(np.trace(ct, axis1=1, axis2=2) - 2.001).max() < 1e-9
    cs = v.getSourceCovariances()
    assert np.abs(np.trace(cs, axis1=1, axis2=2) - 2.001).max() < 1e-9


def _check_case(synth, world, tgt, src, T_true, guess, oracle_pose_check=True, recover_tol=(0.03, 3e-3)):
    from rgc_slam_amd import registration
    from oracle import oracle
    v = registration.odometer_vgicp(0)
    v.setInputTarget(tgt)
    v.setInputSource(src)
    v.align(guess, want_output=False, want_fitness=True)
    T = v.getFinalTransformation().copy()
    fit = v.getFitnessScore()
    assert v.hasConverged()
    # known motion recovered (range noise 1 cm, map sampling 0.3 m)
    assert np.abs(T[:3, 3] - T_true[:3, 3]).max() < recover_tol[0] and _rot_angle(T[:3, :3], T_true[:3, :3]) < recover_tol[1]
    _properties(v, tgt)
    # idempotence: from the answer, the solver stays at the answer
    v.align(T, want_output=False)
    T2 = v.getFinalTransformation()
    assert np.abs(T2[:3, 3] - T[:3, 3]).max() < 1e-4 and _rot_angle(T2[:3, :3], T[:3, :3]) < 1e-4
    # fitness = mean squared nearest-neighbour distance: brute force on a sample of the scan against the map points near it
    rng = np.random.default_rng(5)
    sel = rng.choice(len(src), 64, replace=False)
    p = (T[:3, :3].astype(np.float32) @ src[sel].T.astype(np.float32)).T + T[:3, 3].astype(np.float32)
    near = tgt[(np.abs(tgt[:, 0] - T[0, 3]) < 130) & (np.abs(tgt[:, 1] - T[1, 3]) < 130)]
    d2 = np.array([np.min(np.sum((near - q) ** 2, axis=1)) for q in p])
    v2 = registration.odometer_vgicp(0)
    v2.setInputTarget(tgt)
    v2.setInputSource(np.ascontiguousarray(src[sel]))
    assert abs(v2.fitnessAt(T) - float(d2.mean())) <= 1e-4 * float(d2.mean()) + 1e-9
    v2.close()
    if oracle_pose_check:
        o = oracle.Registration(num_threads=0)
        o.set_target(tgt)
        o.set_source(src)
        To = o.align(guess)
        assert np.abs(T[:3, 3] - To[:3, 3]).max() <= 1e-4 and _rot_angle(T[:3, :3], To[:3, :3]) <= 1e-4
        assert abs(fit - o.fitness()) <= 1e-5 * abs(o.fitness())
    v.close()
    return T


def test_c_main_30k_vs_1M(synth):
    world, tgt = synth.make_world_and_map(1_000_000, seed=synth.SEED)
    T_true = synth.se3(synth.rot_zyx(0.015, 0.002, -0.001), [0.12, 0.01, 0.003])
    src = synth.make_scan_n(world, T_true, 30000, seed=synth.SEED + 1)["xyz"]
    _check_case(synth, world, tgt, src, T_true, np.eye(4, dtype=np.float32))


def test_c3_hdl64_130k_vs_5M(synth, map5m):
    world, tgt = map5m
    T_true = synth.se3(synth.rot_zyx(-0.02, 0.001, 0.002), [0.18, -0.02, 0.004])
    src = synth.make_scan_n(world, T_true, 130000, elev_deg=synth.hdl64_elev(), seed=synth.SEED + 2)["xyz"]
    _check_case(synth, world, tgt, src, T_true, np.eye(4, dtype=np.float32))


def test_c5_250k_vs_20M_with_prior(synth, map5m):
    world, tile = map5m
    # 20 M-point map: the 5 M world plus three translated copies (the scan only sees the original tile)
    L = 2.0 * world.half_extent + 4.0
    tgt = np.concatenate([tile, tile + np.float32([L, 0, 0]), tile + np.float32([0, L, 0]), tile + np.float32([L, L, 0])]).astype(np.float32)
    assert len(tgt) == 20_000_000
    T_true = synth.se3(synth.rot_zyx(0.03, -0.002, 0.001), [0.25, 0.02, -0.002])
    e = synth.hdl64_elev()
    a = synth.make_scan_n(world, T_true, 125000, elev_deg=e, seed=synth.SEED + 3)["xyz"]
    b = synth.make_scan_n(world, T_true, 125000, elev_deg=e + 0.5 * float(np.abs(np.diff(np.sort(e))).min()), seed=synth.SEED + 4)["xyz"]
    src = np.concatenate([a, b]).astype(np.float32)
    # IMU-preintegrated prior (RGC_odometer.cpp:929-931, 993-996): the rotation the gyro measured over the sweep -- a synthetic 200 Hz
    # IMU stream of the motion through rgc_imu_preintegrate -- and no translation (the first frame of a sequence has no previous delta)
    from rgc_slam_amd import odometry
    guess = odometry.imu_rotation_priors([np.eye(4), T_true])[1]
    # the CPU oracle's pose on the same clouds, always: on all 20 M points where the host has the cores for it, otherwise on the ONE tile
    # the scan can see -- the three translated copies lie a map's width away, no voxel the solve looks up and no neighbour of a point of
    # the tile is in them, so the oracle's registration against the tile alone is its registration against the whole map
    T = _check_case(synth, world, tgt, src, T_true, guess, oracle_pose_check=(os.cpu_count() or 1) >= 64)
    if (os.cpu_count() or 1) < 64:
        from oracle import oracle as orc
        o = orc.Registration(num_threads=min(14, os.cpu_count() or 1))
        o.set_target(tile)
        o.set_source(src)
        To = o.align(guess)
        assert np.abs(T[:3, 3] - To[:3, 3]).max() <= 1e-4 and _rot_angle(T[:3, :3], To[:3, :3]) <= 1e-4


def test_c_main_dependent_sequence_vs_oracle(synth):
    """The headline workload as bench.py runs it, at full size: six frames of the DEPENDENT c-main sequence -- every 30 k-point scan
    registered to the 1 M-point map re-expressed on the device in the previous pose's body frame (rgc_set_target_reframed) -- on one
    context and on two taking turns.  Each frame's motion against the CPU oracle started from the SAME previous pose and guess (its
    own transform_cloud of the map, its own registration): <= 1e-4 m / 1e-4 rad; the re-framed cloud itself equals the oracle's to the
    last bit; both modes give bit-identical motions; the accumulated world pose follows the known trajectory."""
    import bench
    from oracle import oracle as orc
    from rgc_slam_amd import registration
    K = 6
    world, tgt = synth.make_world_and_map(1_000_000, seed=synth.SEED)
    poses = synth.make_trajectory(K + 2, seed=synth.SEED)
    scans = [synth.make_scan_n(world, poses[i + 1], 30000, seed=synth.SEED + 100 + i)["xyz"] for i in range(K + 1)]
    pv = registration.PipelinedVGICP(0, depth=2)
    v = pv.v[0]
    def to_dev(xyz):
        a = np.zeros((xyz.shape[0], 4), np.float32); a[:, :3] = xyz
        p = v.device_alloc(a.nbytes); v.upload(p, a); return p, a
    (d_map, map4), d_scans = to_dev(tgt), [to_dev(s)[0] for s in scans]
    seq = bench.DependentSequence(pv.v, d_map, len(tgt), d_scans, [len(s) for s in scans])
    I4 = np.eye(4, dtype=np.float32)
    Tw0 = np.asarray(poses[0], np.float64)
    for w in pv.v:
        seq.v = [w]; _, w0, _ = seq.run(0, 1, Tw0, I4, False)
    seq.v = pv.v
    m1, worlds, guesses = seq.run(1, K, w0[0], I4, False)
    m2, _, _ = seq.run(1, K, w0[0], I4, True)
    assert all(np.array_equal(a, b) for a, b in zip(m1, m2))
    o = orc.Registration(num_threads=min(14, os.cpu_count() or 1))
    for j in range(K):
        Tw_prev = w0[0] if j == 0 else worlds[j - 1]
        q, t = bench.world_to_body(Tw_prev)
        body = orc.transform_cloud(map4, q, t)
        if j in (0, K - 1):
            v.transformCloudDevice(d_map, len(tgt), 16, q, t, seq.d_body[id(v)])
            assert np.array_equal(v.download(seq.d_body[id(v)], (len(tgt), 4)), body)
        o.set_target(body[:, :3].copy()); o.set_source(scans[1 + j])
        To = o.align(guesses[j])
        assert np.abs(m1[j][:3, 3] - To[:3, 3]).max() <= 1e-4 and _rot_angle(m1[j][:3, :3], To[:3, :3]) <= 1e-4, j
    assert np.abs(worlds[-1][:3, 3] - poses[K + 1][:3, 3]).max() < 0.05      # the synthetic trajectory is recovered (metres travelled: ~1)
    seq.close()
    pv.close()


def test_c_main_200_frames(synth):
    """c-main at its stated length -- 200 frames of the dependent sequence (30 k-point scans against the 1 M-point map
    re-expressed by the previous pose), through the C++ frame loop bench.py's timed region runs (librgc_seq.so):
      * two contexts taking turns == one frame at a time, every motion bit for bit; the library's default reuse mode (seeds + lists) ==
        nothing kept, bit for bit;
      * every 40th frame against the CPU oracle started from the SAME previous pose and guess: <= 1e-4 m / 1e-4 rad;
      * the accumulated world pose follows the known trajectory over its whole length."""
    import bench
    from oracle import oracle as orc
    from rgc_slam_amd import registration
    K = 200
    world, tgt = synth.make_world_and_map(1_000_000, seed=synth.SEED)
    poses = synth.make_trajectory(K + 1, seed=synth.SEED)
    scans = [synth.make_scan_n(world, poses[i + 1], 30000, seed=synth.SEED + 100 + i)["xyz"] for i in range(K)]
    pv = registration.PipelinedVGICP(0, depth=2)
    v = pv.v[0]
    def to_dev(xyz):
        a = np.zeros((xyz.shape[0], 4), np.float32); a[:, :3] = xyz
        p = v.device_alloc(a.nbytes); v.upload(p, a); return p, a
    (d_map, map4), d_scans = to_dev(tgt), [to_dev(s)[0] for s in scans]
    seq = bench.DependentSequence(pv.v, d_map, len(tgt), d_scans, [len(s) for s in scans])
    I4, Tw0 = np.eye(4, dtype=np.float32), np.asarray(poses[0], np.float64)
    for w in pv.v:
        w.setNeighbourReuse(registration.FastVGICP.REUSE_NONE)
    m2, worlds, guesses = seq.run_cpp(0, K, Tw0, I4, True)
    m1, _, _ = seq.run_cpp(0, K, Tw0, I4, False)
    assert all(np.array_equal(a, b) for a, b in zip(m1, m2))
    for w in pv.v:
        w.setNeighbourReuse(registration.FastVGICP.REUSE_LISTS)
    ml, _, _ = seq.run_cpp(0, K, Tw0, I4, True)
    assert all(np.array_equal(a, b) for a, b in zip(m2, ml))
    assert v.stats()["searched_target"] < 0.05 * len(tgt) or pv.v[1].stats()["searched_target"] < 0.05 * len(tgt)
    o = orc.Registration(num_threads=min(14, os.cpu_count() or 1))
    for j in (0, 40, 80, 120, 160, K - 1):
        q, t = bench.world_to_body(Tw0 if j == 0 else worlds[j - 1])
        body = orc.transform_cloud(map4, q, t)
        o.set_target(body[:, :3].copy()); o.set_source(scans[j])
        To = o.align(guesses[j])
        assert np.abs(m2[j][:3, 3] - To[:3, 3]).max() <= 1e-4 and _rot_angle(m2[j][:3, :3], To[:3, :3]) <= 1e-4, j
    path = float(sum(np.linalg.norm(poses[i + 1][:3, 3] - poses[i][:3, 3]) for i in range(K)))
    err = float(np.abs(worlds[-1][:3, 3] - poses[K][:3, 3]).max())
    assert path > 10.0 and err < 0.005 * path + 0.1, (path, err)      # the synthetic trajectory is recovered over its whole length
    seq.close()
    pv.close()
