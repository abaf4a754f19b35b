"""f2: the local map resident on the device (rgc_map_*).  Parity definition of this row:
(1) the device map and its committed target equal, bit for bit, the composition of the oracle's transformPointCloud and
VoxelGrid on the same keyframes; (2) the frame body driven through the resident map matches the same frame body driven by the
CPU oracle to 1e-4 m / 1e-4 rad per frame; (3) against the reference's own semantics (keyframes re-framed into the previous
body frame every frame) only the lattice alignment differs: both follow the true motion equally well.  -m gpu."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _angle(qa, qb):
    return 2 * np.arccos(min(1.0, abs(float(np.dot(qa, qb)))))


def _quat(rng, ang):
    a = rng.normal(0, 1, 3); a *= ang / np.linalg.norm(a)
    th = np.linalg.norm(a)
    return np.concatenate([np.sin(th / 2) * a / th, [np.cos(th / 2)]])


@pytest.fixture(scope="module")
def clouds():
    import rgc_slam_amd.synth as synth
    world = synth.make_world(half_extent=45.0, seed=synth.SEED)
    poses = synth.make_trajectory(6, seed=synth.SEED)
    out = []
    for k in range(5):
        sc = synth.make_scan(world, poses[k], n_az=900, seed=synth.SEED + 70 + k)
        out.append(np.concatenate([sc["xyz"], sc["intensity"][:, None]], axis=1).astype(np.float32))
    return out, poses


def test_map_state_matches_oracle_composition(clouds):
    from rgc_slam_amd import registration, local_map, _lib
    from oracle_backend import OracleBackend
    scans, _ = clouds
    rng = np.random.default_rng(3)
    reg = registration.odometer_vgicp(0)
    m = local_map.RollingLocalMap(reg)
    ob = OracleBackend()
    origin = np.array([100.0, -40.0, 2.0])
    m.reset(origin); ob.map_reset(origin)
    with pytest.raises(_lib.RgcError):
        m.commit(0.3)                                        # empty map
    kf_poses = []
    for k, sc in enumerate(scans):
        q, t = _quat(rng, 0.05 * (k + 1)), origin + rng.normal(0, 1.0, 3) + np.array([0.4 * k, 0, 0])
        kf_poses.append((q, t))
        assert m.insert(sc, q, t) == ob.map_insert(sc, q, t) == k
        assert np.array_equal(m.points(), ob.map_points())   # fp64 q*p + (t - origin), stored fp32: bit for bit
    info = m.info()
    assert info["n_keyframes"] == 5 and info["n_points"] == sum(len(s) for s in scans) and info["n_target"] == -1
    n = m.commit(0.3)
    tg = m.target()
    assert n == len(tg) and np.array_equal(tg, ob.map_target(0.3))       # VoxelGrid over the resident store
    rev = m.info()["revision"]
    assert m.commit(0.3) == n and m.info()["revision"] == rev           # unchanged map: no rebuild
    # pop_front to the reference's deque length
    assert m.evict(3) == ob.map_evict(3) == 2
    i2 = m.info()
    assert (i2["n_keyframes"], i2["oldest_id"], i2["newest_id"], i2["n_target"]) == (3, 2, 4, -1)
    assert np.array_equal(m.points(), ob.map_points())
    assert np.array_equal(m.target() if m.commit(0.3) else None, ob.map_target(0.3))
    # eviction by distance: keep keyframes within 1.2 m of the newest pose
    c = kf_poses[4][1]
    assert m.evict(0, c, 1.2) == ob.map_evict(0, c, 1.2)
    assert np.array_equal(m.points(), ob.map_points()) and m.info()["n_keyframes"] == len(ob._kf) >= 1
    # moving the origin shifts every stored point once (fp64 -> fp32)
    o2 = origin + np.array([30.0, -20.0, 0.5])
    m.rebase(o2); ob.map_rebase(o2)
    assert np.array_equal(m.points(), ob.map_points()) and np.array_equal(m.info()["origin"], o2)
    m.commit(0.3)
    assert np.array_equal(m.target(), ob.map_target(0.3))
    # a plain setInputTarget unbinds the map; the next commit re-binds it even though no keyframe changed
    reg.setInputTarget(scans[0][:, :3])
    assert m.info()["n_target"] == -1
    assert m.commit(0.3) == len(ob.map_target(0.3))
    reg.close()


def test_registration_against_resident_map_matches_oracle(clouds):
    """the same scan registered (a) to the committed device map and (b) by the oracle to the oracle-composed map: same pose; and
    a second scan registered WITHOUT a map change re-uses the resident target (no rebuild) with the same result as a rebuild"""
    from rgc_slam_amd import registration, local_map
    from oracle_backend import OracleBackend
    scans, poses = clouds
    reg = registration.odometer_vgicp(0)
    m = local_map.RollingLocalMap(reg)
    ob = OracleBackend()
    m.reset(None); ob.map_reset(np.zeros(3))
    q0 = np.array([0, 0, 0, 1.0])
    for k in range(3):     # keyframes at their true poses relative to pose 0
        T = np.linalg.inv(poses[0]) @ poses[k]
        w = np.sqrt(max(0.0, 1 + np.trace(T[:3, :3]))) / 2
        q = np.array([(T[2, 1] - T[1, 2]) / (4 * w), (T[0, 2] - T[2, 0]) / (4 * w), (T[1, 0] - T[0, 1]) / (4 * w), w])
        m.insert(scans[k], q, T[:3, 3]); ob.map_insert(scans[k], q, T[:3, 3])
    m.commit(0.3)
    pre = OracleBackend()
    for k in (3, 4):
        src = pre.voxelgrid(scans[k], 0.2)
        guess = (np.linalg.inv(poses[0]) @ poses[k - 1]).astype(np.float32)
        reg.setInputSource(src)
        reg.align(guess, want_output=False, want_fitness=True)
        Tg, fg = reg.getFinalTransformation(), reg.getFitnessScore()
        To, fo = ob.map_register(src, guess, 0.3)
        assert np.abs(Tg - To).max() < 1e-5 and abs(fg - fo) <= 1e-6 * max(1.0, fo)
        true = np.linalg.inv(poses[0]) @ poses[k]
        assert np.linalg.norm(Tg[:3, 3] - true[:3, 3]) < 0.1
        assert m.commit(0.3) > 0      # still bound, nothing to rebuild
    reg.close()


def test_rolling_sequence_vs_oracle_and_vs_reference_semantics():
    import rgc_slam_amd.synth as synth
    from rgc_slam_amd import odometry
    from oracle_backend import OracleBackend
    world = synth.make_world(half_extent=45.0, seed=synth.SEED)
    poses = synth.make_trajectory(11, seed=synth.SEED)
    raws = []
    for k in range(10):
        sc = synth.make_scan(world, poses[k], n_az=1200, seed=synth.SEED + 50 + k, T_ws_end=poses[k + 1])
        raws.append(np.concatenate([sc["xyz"], sc["intensity"][:, None]], axis=1).astype(np.float32))
    hb = odometry.HipBackend(0)
    og, oc = odometry.RollingOdometer(hb), odometry.RollingOdometer(OracleBackend())
    og.rebase_distance = oc.rebase_distance = 0.5           # exercise the re-basing inside the sequence
    hb2 = odometry.HipBackend(0)
    oref = odometry.Odometer(hb2)                           # the reference's semantics (body-frame sub-map rebuilt per frame)
    prev_g, prev_c = (np.array([0, 0, 0, 1.0]), np.zeros(3)), (np.array([0, 0, 0, 1.0]), np.zeros(3))
    worst_t = worst_r = 0.0
    err_roll, err_ref = [], []
    for k, raw in enumerate(raws):
        qg, tg = og.process(raw)
        qc, tc = oc.process(raw)
        qr, tr = oref.process(raw)
        worst_t = max(worst_t, float(np.abs((tg - prev_g[1]) - (tc - prev_c[1])).max()))
        worst_r = max(worst_r, abs(_angle(qg, prev_g[0]) - _angle(qc, prev_c[0])), _angle(qg, qc))
        prev_g, prev_c = (qg, tg), (qc, tc)
        if k >= 1:
            true = np.linalg.inv(poses[1]) @ poses[k + 1]   # the estimate's world frame is the sensor frame of sweep 1's END... coarse
            err_roll.append(np.linalg.norm(tg - true[:3, 3])); err_ref.append(np.linalg.norm(tr - true[:3, 3]))
    assert og.frames == 10 and og.n_commits >= 2 and np.linalg.norm(og.t_w_curr) > 0.3
    assert not np.array_equal(og.origin, np.zeros(3))       # the origin moved at least once
    assert worst_t <= 1e-4 and worst_r <= 1e-4, (worst_t, worst_r)
    # (3): same accuracy class as the reference's semantics -- the two differ only by lattice alignment
    assert np.linalg.norm(og.t_w_curr - oref.t_w_curr) < 0.05, (og.t_w_curr, oref.t_w_curr)
    assert max(err_roll) < max(0.05, 1.5 * max(err_ref)) + 0.5
    hb.close(); hb2.close()


def test_map_edge_cases_and_growth(clouds):
    import ctypes as C
    from rgc_slam_amd import registration, local_map, _lib
    from oracle_backend import OracleBackend
    scans, _ = clouds
    reg = registration.odometer_vgicp(0)
    m = local_map.RollingLocalMap(reg)
    L, h = reg._L, reg._h
    q, t = np.array([0, 0, 0, 1.0]), np.zeros(3)
    dp = C.POINTER(C.c_double)
    a = np.ascontiguousarray(scans[0])
    # bad arguments: empty keyframe, x,y,z-only stride, null pose
    assert L.rgc_map_insert(h, a.ctypes.data, 0, 16, q.ctypes.data_as(dp), t.ctypes.data_as(dp), 0, None) == _lib.ERR_INVALID
    assert L.rgc_map_insert(h, a.ctypes.data, 10, 12, q.ctypes.data_as(dp), t.ctypes.data_as(dp), 0, None) == _lib.ERR_INVALID
    assert L.rgc_map_insert(h, a.ctypes.data, 10, 16, None, t.ctypes.data_as(dp), 0, None) == _lib.ERR_INVALID
    assert L.rgc_map_commit(h, C.c_float(-1.0), None) == _lib.ERR_INVALID
    # a map with fewer points than k: the commit fails like setInputTarget on such a cloud, and nothing is bound
    m.reset(None)
    m.insert(scans[0][:7], q, t)
    with pytest.raises(_lib.RgcError) as e:
        m.commit(0.3)
    assert e.value.status == -3 and m.info()["n_target"] == -1          # RGC_ERR_TOO_FEW_POINTS
    # reset drops the bound target: align has no input any more
    m.reset(None)
    m.insert(scans[0], q, t); m.commit(0.3)
    reg.setInputSource(scans[1][:, :3])
    reg.align(np.eye(4, dtype=np.float32), want_output=False)
    m.reset(None)
    with pytest.raises(_lib.RgcError):
        reg.align(np.eye(4, dtype=np.float32), want_output=False)
    # a keyframe already resident on the device, and growth of the store across many inserts (the content must survive)
    ob = OracleBackend(); ob.map_reset(np.zeros(3))
    d = reg.device_alloc(a.nbytes); reg.upload(d, a)
    kid = C.c_int(-1)
    assert L.rgc_map_insert(h, C.c_void_p(d), len(a), 16, q.ctypes.data_as(dp), t.ctypes.data_as(dp), 1, C.byref(kid)) == 0 and kid.value >= 0
    ob.map_insert(a, q, t)
    rng = np.random.default_rng(9)
    for k in range(40):                                   # 41 keyframes x ~14 k points: several re-allocations of the store
        tk = rng.normal(0, 2.0, 3)
        m.insert(scans[k % 5], q, tk); ob.map_insert(scans[k % 5], q, tk)
    assert np.array_equal(m.points(), ob.map_points())
    assert m.evict(0, np.zeros(3), 2.0) == ob.map_evict(0, np.zeros(3), 2.0)   # scattered survivors: several compaction runs
    assert np.array_equal(m.points(), ob.map_points())
    n = m.commit(0.3)
    assert n == len(ob.map_target(0.3)) and np.array_equal(m.target(), ob.map_target(0.3))
    reg.device_free(d)
    reg.close()
