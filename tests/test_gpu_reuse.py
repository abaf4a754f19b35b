"""What a context keeps between the targets rgc_set_target_reframed prepares (rgc_set_knn_reuse: nothing / seeds / seeds + neighbour
lists) must never show in a result.  Parity here is HIP against HIP -- a context that keeps NOTHING (the plain exact search of every
query, the route tests/test_gpu_parity.py pins to the CPU oracle) -- bit for bit, plus the oracle's covariances (<= 1e-9; the oracle is a
restatement of impl/fast_gicp_impl.hpp:241-298, "parity unpinned": DESIGN.md section 3) on the frames named below.
Also here (round 6's other API additions): the C++ frame loop against the Python one, and every RegularizationMethod / VoxelAccumulationMode
-- the general covariance route -- against the oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def reg_mod():
    from rgc_slam_amd import registration
    return registration


def _pair(reg_mod):
    """(a context with the library's default, a context that keeps nothing)"""
    v = reg_mod.odometer_vgicp(0)
    w = reg_mod.odometer_vgicp(0)
    w.setNeighbourReuse(reg_mod.FastVGICP.REUSE_NONE)
    assert v.getNeighbourReuse() == reg_mod.FastVGICP.REUSE_LISTS and w.getNeighbourReuse() == reg_mod.FastVGICP.REUSE_NONE
    return v, w


def _same_target(v, w, what):
    cv, cw = v.getTargetCovariances(), w.getTargetCovariances()
    assert np.array_equal(cv, cw), what
    xv, xw = v.getVoxels(), w.getVoxels()
    assert np.array_equal(xv["coords"], xw["coords"]) and np.array_equal(xv["cov"], xw["cov"]) and np.array_equal(xv["mean"], xw["mean"]), what
    return cv


def _pose(j):
    import bench
    import rgc_slam_amd.synth as synth
    Tw = synth.se3(synth.rot_zyx(0.55 * j - 0.5, 0.03 * np.sin(j), -0.02 * np.cos(j)), [7.5 * np.cos(j), -11.0 * np.sin(1.3 * j), 0.3 * j])
    return bench.world_to_body(Tw)


def test_knn_reuse_modes(reg_mod, orc):
    """rgc_set_knn_reuse on a live context: LISTS -> NONE -> SEEDS -> LISTS, the same map handed over throughout.  rgc_stats::searched_target
    says which route ran; every covariance and the voxel table are those of the context that never kept anything."""
    import rgc_slam_amd.synth as synth
    n = 100000
    _, tgt = synth.make_world_and_map(n, seed=synth.SEED)
    a = np.zeros((n, 4), np.float32); a[:, :3] = tgt
    v, w = _pair(reg_mod)
    d_map, d_body = v.device_alloc(16 * n), v.device_alloc(16 * n)
    d_map_w, d_body_w = w.device_alloc(16 * n), w.device_alloc(16 * n)
    v.upload(d_map, a); w.upload(d_map_w, a)
    F = reg_mod.FastVGICP
    plan = [(None, "all"), (None, "few"), (None, "few"), (F.REUSE_NONE, "all"), (None, "all"), (F.REUSE_SEEDS, "all"), (None, "all"),
            (F.REUSE_LISTS, "all"), (None, "few"), (None, "few")]
    for j, (mode, searched) in enumerate(plan):
        if mode is not None:
            v.setNeighbourReuse(mode)
            assert v.getNeighbourReuse() == mode
        q, t = _pose(j)
        v.setInputTargetReframed(d_map, n, 16, q, t, d_body)
        w.setInputTargetReframed(d_map_w, n, 16, q, t, d_body_w)
        cv = _same_target(v, w, (j, mode))
        got = v.stats()["searched_target"]
        assert (got == n) if searched == "all" else (got <= 0.15 * n), (j, mode, got)
        assert w.stats()["searched_target"] == n
    body = v.download(d_body, (n, 4))
    ocov, _ = orc.covariances(body[:, :3].copy(), k=20)
    assert np.abs(cv - ocov).max() <= 1e-9
    with pytest.raises(reg_mod.RgcError):
        v.setNeighbourReuse(3)
    for p in (d_map, d_body):
        v.device_free(p)
    for p in (d_map_w, d_body_w):
        w.device_free(p)
    v.close(); w.close()


def test_quaternion_next_to_the_unit_gate(reg_mod):
    """The certificates' error budget is a rigid motion's; rgc_set_target_reframed applies q as Eigen does, without normalising it, so a
    quaternion off the unit sphere scales every distance.  |q|^2 within 1e-9 of one (any fp64-normalised quaternion): lists serve, and must
    give the search's bits on either side of one; |q|^2 = 1 +- 2e-7 (a quaternion normalised in fp32): no lists that frame, none trusted
    after it, the same bits."""
    import rgc_slam_amd.synth as synth
    n = 100000
    _, tgt = synth.make_world_and_map(n, seed=synth.SEED + 3)
    a = np.zeros((n, 4), np.float32); a[:, :3] = tgt
    v, w = _pair(reg_mod)
    d_map, d_body = v.device_alloc(16 * n), v.device_alloc(16 * n)
    d_map_w, d_body_w = w.device_alloc(16 * n), w.device_alloc(16 * n)
    v.upload(d_map, a); w.upload(d_map_w, a)
    scales = [(1.0, "all"), (1.0, "few"), (np.sqrt(1.0 + 4e-10), "few"), (np.sqrt(1.0 - 4e-10), "few"), (np.sqrt(1.0 + 2e-7), "all"),
              (1.0, "all"), (1.0, "few"), (np.sqrt(1.0 - 2e-7), "all"), (1.0, "all"), (1.0, "few")]
    for j, (sc, searched) in enumerate(scales):
        q, t = _pose(j)
        q = [x * sc for x in q]
        v.setInputTargetReframed(d_map, n, 16, q, t, d_body)
        w.setInputTargetReframed(d_map_w, n, 16, q, t, d_body_w)
        _same_target(v, w, (j, sc))
        got = v.stats()["searched_target"]
        assert (got == n) if searched == "all" else (got <= 0.15 * n), (j, sc, got)
    for p in (d_map, d_body):
        v.device_free(p)
    for p in (d_map_w, d_body_w):
        w.device_free(p)
    v.close(); w.close()


def test_map_edited_every_frame_at_full_size(reg_mod, orc):
    """The c-main map (1 M points) with 0.1 %, 1 % and 10 % of its rows overwritten before every frame -- what a rolling map does when
    keyframes come and go (RGC_odometer.cpp:1236-1247) -- and frames in between where nothing changes: the default context (seeds + lists,
    all-or-nothing invalidation) against one that keeps nothing, every covariance and the voxel table bit for bit; against the CPU
    oracle's covariances (<= 1e-9) once per fraction."""
    import os
    import rgc_slam_amd.synth as synth
    n = 1000000
    world, tgt = synth.make_world_and_map(n, seed=synth.SEED)
    a = np.zeros((n, 4), np.float32); a[:, :3] = tgt
    rng = np.random.default_rng(41)
    v, w = _pair(reg_mod)
    d_map, d_body = v.device_alloc(16 * n), v.device_alloc(16 * n)
    d_map_w, d_body_w = w.device_alloc(16 * n), w.device_alloc(16 * n)
    v.upload(d_map, a); w.upload(d_map_w, a)
    j = 0
    o_threads = min(14, os.cpu_count() or 1)
    for frac in (0.001, 0.01, 0.1):
        m = int(n * frac)
        for rep in range(3):
            # rows [lo, lo + m) replaced by other points of the same surfaces (a keyframe's worth of new returns), then a frame; then one more
            # frame with the buffer left alone
            lo = int(rng.integers(0, n - m))
            src = rng.integers(0, n, m)
            a[lo:lo + m, :3] = tgt[src] + rng.normal(0.0, 0.02, (m, 3)).astype(np.float32)
            v.upload(d_map + 16 * lo, a[lo:lo + m]); w.upload(d_map_w + 16 * lo, a[lo:lo + m])
            for changed in (True, False):
                q, t = _pose(j); j += 1
                v.setInputTargetReframed(d_map, n, 16, q, t, d_body)
                w.setInputTargetReframed(d_map_w, n, 16, q, t, d_body_w)
                cv = _same_target(v, w, (frac, rep, changed))
                got = v.stats()["searched_target"]
                assert (got == n) if changed else (got < 0.1 * n), (frac, rep, changed, got)
        body = v.download(d_body, (n, 4))
        ocov, _ = orc.covariances(body[:, :3].copy(), k=20, threads=o_threads)
        assert np.abs(cv - ocov).max() <= 1e-9, frac
    for p in (d_map, d_body):
        v.device_free(p)
    for p in (d_map_w, d_body_w):
        w.device_free(p)
    v.close(); w.close()


def test_soak_of_the_neighbour_lists(reg_mod):
    """scripts/soak_lists.py as a test (it found the one real bug of the lists, a flag word overwritten by the launch that reads it): a 1 M-point
    map under 30 random poses -- any yaw, +-30 m, every seventh frame +-400 m -- with a handful of points moved every thirteenth frame;
    every frame's covariances and voxel table against a context that keeps nothing."""
    import bench
    import rgc_slam_amd.synth as synth
    n = 1000000
    _, tgt = synth.make_world_and_map(n, seed=synth.SEED)
    a = np.zeros((n, 4), np.float32); a[:, :3] = tgt
    v, w = _pair(reg_mod)
    dm, db = v.device_alloc(a.nbytes), v.device_alloc(a.nbytes)
    dmw, dbw = w.device_alloc(a.nbytes), w.device_alloc(a.nbytes)
    v.upload(dm, a); w.upload(dmw, a)
    rng = np.random.default_rng(5)
    searched = []
    for f in range(30):
        if f and f % 13 == 0:
            jj = rng.integers(0, n, 5)
            a[jj, :3] += rng.normal(0, 0.2, (5, 3)).astype(np.float32)
            v.upload(dm, a); w.upload(dmw, a)
        ang = rng.uniform(-np.pi, np.pi, 3) * np.array([1.0, 0.02, 0.02])
        scale = 400.0 if f % 7 == 6 else 30.0
        Tw = synth.se3(synth.rot_zyx(*ang), rng.uniform(-scale, scale, 3) * np.array([1, 1, 0.05]))
        q, t = bench.world_to_body(Tw)
        v.setInputTargetReframed(dm, n, 16, q, t, db)
        w.setInputTargetReframed(dmw, n, 16, q, t, dbw)
        _same_target(v, w, f)
        searched.append(int(v.stats()["searched_target"]))
    assert min(searched) < 0.05 * n and searched[13] == n, searched
    for p in (dm, db):
        v.device_free(p)
    for p in (dmw, dbw):
        w.device_free(p)
    v.close(); w.close()


def test_cpp_frame_loop_equals_the_python_one(reg_mod):
    """librgc_seq.so (rgc_seq_run_dependent: the dependent frame loop of rgc::DependentSequence in C++, what bench.py's timed region runs)
    against bench.DependentSequence.run, the same calls issued from Python: every motion and every world pose bit for bit, on two contexts
    and one frame at a time."""
    import bench
    import rgc_slam_amd.synth as synth
    from rgc_slam_amd import _lib
    assert _lib.load_seq() is not None, "librgc_seq.so missing: python rgc-slam_amd/build.py"
    world, tgt = synth.make_world_and_map(120000, seed=synth.SEED + 5)
    poses = synth.make_trajectory(9, seed=synth.SEED + 5)
    scans = [synth.make_scan_n(world, poses[i + 1], 15000, seed=synth.SEED + 900 + i)["xyz"] for i in range(8)]
    pv = reg_mod.PipelinedVGICP(0, depth=2)
    v = pv.v[0]
    def to_dev(xyz):
        a = np.zeros((xyz.shape[0], 4), np.float32); a[:, :3] = xyz
        p = v.device_alloc(a.nbytes); v.upload(p, a); return p
    d_map, d_scans = to_dev(tgt), [to_dev(s_) for s_ in scans]
    seq = bench.DependentSequence(pv.v, d_map, len(tgt), d_scans, [len(s_) for s_ in scans])
    Tw0, I4 = np.asarray(poses[0], np.float64), np.eye(4, dtype=np.float32)
    for overlap in (True, False):
        for mode in (reg_mod.FastVGICP.REUSE_NONE, reg_mod.FastVGICP.REUSE_LISTS):
            for w in pv.v:
                w.setNeighbourReuse(mode)
            mp, wp, gp = seq.run(0, len(scans), Tw0, I4, overlap)
            st = []
            mc, wc, gc = seq.run_cpp(0, len(scans), Tw0, I4, overlap, stamps=st)
            assert len(st) == len(scans) and all(b > a for a, b in zip(st, st[1:]))
            for i in range(len(scans)):
                assert np.array_equal(mp[i], mc[i]) and np.array_equal(gp[i], gc[i]), (overlap, mode, i)
            for i in range(len(scans) - 1):   # (the last frame's world pose is composed by the caller: numpy's matmul there, a loop here)
                assert np.array_equal(wp[i], wc[i]), (overlap, mode, i)
            assert np.abs(wp[-1] - wc[-1]).max() <= 1e-12
    seq.close(); pv.close()


def _general_case(reg_mod, orc, reg, vox, tgt, src, guess):
    """one (RegularizationMethod, VoxelAccumulationMode) on the HIP library against the CPU oracle: covariances, voxel table, H / b / cost at
    the guess, the registration itself"""
    v = reg_mod.odometer_vgicp(0)
    v.setRegularizationMethod(reg)
    v.setVoxelAccumulationMode(vox)
    v.setInputTarget(tgt); v.setInputSource(src)
    o = orc.Registration(max_iterations=25, translation_eps=1e-6, regularization=reg, voxel_mode=vox, num_threads=min(14, __import__("os").cpu_count() or 1))
    o.set_target(tgt); o.set_source(src); o.prepare()
    ct, cs = v.getTargetCovariances(), v.getSourceCovariances()
    ot, os_ = orc.covariances_m(tgt, reg), orc.covariances_m(src, reg)
    scale = max(1.0, float(np.abs(ot).max()))
    assert np.abs(ct - ot).max() <= 1e-9 * scale and np.abs(cs - os_).max() <= 1e-9 * scale, (reg, vox, np.abs(ct - ot).max(), np.abs(cs - os_).max())
    xv, xo = v.getVoxels(), o.voxelmap()       # (both sorted by voxel coordinates)
    assert np.array_equal(xv["coords"], xo["coords"]) and np.array_equal(xv["num"], xo["num"])
    vs = max(1.0, float(np.abs(xo["cov"]).max()))
    assert np.abs(xv["mean"] - xo["mean"]).max() <= 1e-8 and np.abs(xv["cov"] - xo["cov"]).max() <= 1e-8 * vs, (reg, vox)
    cost, H, b = v.linearize(guess.astype(np.float64))
    co, Ho, bo = o.linearize(guess.astype(np.float64))
    assert abs(cost - co) <= 1e-8 * abs(co) and np.abs(H - Ho).max() <= 1e-8 * np.abs(Ho).max() and np.abs(b - bo).max() <= 1e-8 * np.abs(bo).max(), (reg, vox)
    v.align(guess, want_output=False, want_fitness=True)
    T, To = v.getFinalTransformation(), o.align(guess)
    R = T[:3, :3].astype(np.float64) @ To[:3, :3].astype(np.float64).T
    dth = float(np.arcsin(min(1.0, 0.5 * np.linalg.norm([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]]))))
    assert np.abs(T[:3, 3] - To[:3, 3]).max() <= 1e-4 and dth <= 1e-4, (reg, vox, T, To)
    assert abs(v.getFitnessScore() - o.fitness()) <= 1e-5 * o.fitness()
    # the two halves (on this route the solve runs in the first) and a second solve give the same pose
    v.align_begin(guess, want_fitness=True); T2 = v.align_end()
    assert np.array_equal(T, T2)
    v.close()
    return T


def test_every_regularization_method_and_voxel_mode(reg_mod, orc):
    """FastGICP::setRegularizationMethod x FastVGICP::setVoxelAccumulationMode (gicp_settings.hpp:6,10; fast_gicp_impl.hpp:262-293;
    fast_vgicp_voxel.hpp:76-122): NONE, MIN_EIG, NORMALIZED_MIN_EIG, FROBENIUS and MULTIPLICATIVE run on the library's general route (a 3x3
    per point), PLANE with an additive mode on the tuned kernels.  Each combination against the CPU oracle -- covariances and voxel table to
    1e-9 / 1e-8 relative, H / b / cost to 1e-8, the pose to 1e-4 m / 1e-4 rad -- whose general methods tests/test_oracle_golden.py pins to
    the literal numpy restatement."""
    import rgc_slam_amd.synth as synth
    F = reg_mod.FastVGICP
    world, tgt = synth.make_world_and_map(40000, seed=synth.SEED + 8)
    T_true = synth.se3(synth.rot_zyx(0.015, 0.002, -0.002), [0.12, -0.02, 0.003])
    src = synth.make_scan_n(world, T_true, 9000, seed=synth.SEED + 8)["xyz"]
    I4 = np.eye(4, dtype=np.float32)
    poses = {}
    for reg in (F.REG_MIN_EIG, F.REG_NORMALIZED_MIN_EIG, F.REG_FROBENIUS, F.REG_PLANE):
        for vox in (F.VOXEL_ADDITIVE, F.VOXEL_MULTIPLICATIVE):
            poses[(reg, vox)] = _general_case(reg_mod, orc, reg, vox, tgt, src, I4)
    _general_case(reg_mod, orc, F.REG_NONE, F.VOXEL_ADDITIVE, tgt, src, I4)     # (NONE with MULTIPLICATIVE inverts rank-deficient covariances: undefined in the reference)
    # every regularised variant recovers the known motion of the synthetic scan
    for key, T in poses.items():
        assert np.abs(T[:3, 3] - T_true[:3, 3]).max() < 0.05, (key, T[:3, 3])
    # ADDITIVE_WEIGHTED is ADDITIVE in the vendored FastVGICP (fast_vgicp_voxel.hpp:137-141): the tuned route, the same bits
    v = reg_mod.odometer_vgicp(0)
    v.setInputTarget(tgt); v.setInputSource(src); v.align(I4, want_output=False)
    T0 = v.getFinalTransformation()
    v.setVoxelAccumulationMode(F.VOXEL_ADDITIVE_WEIGHTED)
    v.align(I4, want_output=False)                                               # (no change of route: the clouds stay)
    assert np.array_equal(T0, v.getFinalTransformation())
    assert np.array_equal(T0, poses[(F.REG_PLANE, F.VOXEL_ADDITIVE)])
    # a change of route drops the clouds (the reference would compute covariances under the new method at align()): set them again
    v.setRegularizationMethod(F.REG_MIN_EIG)
    with pytest.raises(reg_mod.RgcError):
        v.align(I4, want_output=False)
    v.setInputTarget(tgt); v.setInputSource(src); v.align(I4, want_output=False)
    assert np.array_equal(v.getFinalTransformation(), poses[(F.REG_MIN_EIG, F.VOXEL_ADDITIVE)])
    with pytest.raises(reg_mod.RgcError):
        v.setRegularizationMethod(7)
    v.close()


def test_the_two_routes_agree_on_the_odometers_settings(reg_mod, monkeypatch):
    """PLANE / ADDITIVE through the general route (RGC_FORCE_GENERAL=1: every point through the cooperative search, a 3x3 per point, the
    host-driven LM) against the tuned kernels: the same neighbour sets, covariances to 1e-9, the same voxels, the pose to 1e-6."""
    import rgc_slam_amd.synth as synth
    world, tgt = synth.make_world_and_map(60000, seed=synth.SEED + 9)
    src = synth.make_scan_n(world, synth.se3(synth.rot_zyx(0.01, 0.0, 0.001), [0.1, 0.02, 0.0]), 12000, seed=synth.SEED + 9)["xyz"]
    a = reg_mod.odometer_vgicp(0)
    monkeypatch.setenv("RGC_FORCE_GENERAL", "1")
    b = reg_mod.odometer_vgicp(0)
    monkeypatch.delenv("RGC_FORCE_GENERAL")
    I4 = np.eye(4, dtype=np.float32)
    for v in (a, b):
        v.setInputTarget(tgt); v.setInputSource(src)
    assert np.abs(a.getTargetCovariances() - b.getTargetCovariances()).max() <= 1e-9
    assert np.abs(a.getSourceCovariances() - b.getSourceCovariances()).max() <= 1e-9
    xa, xb = a.getVoxels(), b.getVoxels()
    assert np.array_equal(xa["coords"], xb["coords"]) and np.abs(xa["cov"] - xb["cov"]).max() <= 1e-9 and np.abs(xa["mean"] - xb["mean"]).max() <= 1e-9
    a.align(I4, want_output=False); b.align(I4, want_output=False)
    assert np.abs(a.getFinalTransformation() - b.getFinalTransformation()).max() <= 1e-6
    # a general-route cloud has no normals to hand out
    import ctypes as C
    from rgc_slam_amd import _lib
    nrm = np.empty((len(tgt), 3))
    assert b._L.rgc_get_target_covariances(b._h, None, nrm.ctypes.data_as(C.POINTER(C.c_double))) == _lib.ERR_UNSUPPORTED
    assert a._L.rgc_get_target_covariances(a._h, None, nrm.ctypes.data_as(C.POINTER(C.c_double))) == 0 and abs(np.linalg.norm(nrm[0]) - 1) < 1e-12
    a.close(); b.close()


def test_lists_that_do_not_fit_drop_the_context_to_seeds(reg_mod, monkeypatch):
    """The neighbour lists are an optimisation (112 bytes per map point): when their buffers cannot be had -- RGC_TEST_FAIL_CACHE_ALLOC walks
    that path -- the call that wanted them succeeds all the same, the context reports RGC_REUSE_SEEDS from then on, and results stay those
    of a context that keeps nothing."""
    import rgc_slam_amd.synth as synth
    n = 60000
    _, tgt = synth.make_world_and_map(n, seed=synth.SEED + 4)
    a = np.zeros((n, 4), np.float32); a[:, :3] = tgt
    monkeypatch.setenv("RGC_TEST_FAIL_CACHE_ALLOC", "1")
    v = reg_mod.odometer_vgicp(0)
    monkeypatch.delenv("RGC_TEST_FAIL_CACHE_ALLOC")
    w = reg_mod.odometer_vgicp(0)
    w.setNeighbourReuse(reg_mod.FastVGICP.REUSE_NONE)
    assert v.getNeighbourReuse() == reg_mod.FastVGICP.REUSE_LISTS
    d_map, d_body = v.device_alloc(16 * n), v.device_alloc(16 * n)
    d_map_w, d_body_w = w.device_alloc(16 * n), w.device_alloc(16 * n)
    v.upload(d_map, a); w.upload(d_map_w, a)
    for j in range(4):
        q, t = _pose(j)
        v.setInputTargetReframed(d_map, n, 16, q, t, d_body)
        w.setInputTargetReframed(d_map_w, n, 16, q, t, d_body_w)
        _same_target(v, w, j)
        assert v.stats()["searched_target"] == n
        assert v.getNeighbourReuse() == reg_mod.FastVGICP.REUSE_SEEDS
    for p in (d_map, d_body):
        v.device_free(p)
    for p in (d_map_w, d_body_w):
        w.device_free(p)
    v.close(); w.close()
