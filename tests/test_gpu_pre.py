"""B2 de-skew, B3 VoxelGrid, B9 sub-map transform: HIP path vs the CPU oracle and the golden fixture.  -m gpu."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pre():
    from rgc_slam_amd import odometry
    p = odometry.Preprocessor(0)
    yield p
    p.close()


def _scan(n_az=1200, seed=3):
    import rgc_slam_amd.synth as synth
    w = synth.make_world(half_extent=40.0, seed=synth.SEED)
    sc = synth.make_scan(w, np.eye(4), n_az=n_az, seed=synth.SEED + seed)
    return np.concatenate([sc["xyz"], (sc["ring"] + 0.1 * sc["rel_time"])[:, None].astype(np.float32)], axis=1)


def test_voxelgrid_golden(pre, fx_vg):
    for leaf, key in ((0.2, "out_02"), (0.3, "out_03")):
        out = pre.voxelGridFilter(fx_vg["xyzi"], leaf)
        assert out.shape == fx_vg[key].shape
        assert np.abs(out - fx_vg[key]).max() < 1e-4


def test_voxelgrid_vs_oracle_bit_exact(pre, orc):
    xyzi = _scan()
    for leaf in (0.2, 0.3, 1.0):
        got, exp = pre.voxelGridFilter(xyzi, leaf), orc.voxelgrid_filter(xyzi, leaf)
        assert got.shape == exp.shape
        assert np.array_equal(got, exp), np.abs(got - exp).max()   # same fp32 sums in the same order
    # 32-byte stride (pcl::PointXYZI) and a single-leaf cloud
    wide = np.zeros((len(xyzi), 8), np.float32); wide[:, :3] = xyzi[:, :3]; wide[:, 3] = xyzi[:, 3]
    tight4 = np.concatenate([wide[:, :3], wide[:, 3:4]], axis=1)
    assert np.array_equal(pre.voxelGridFilter(wide, 0.2), orc.voxelgrid_filter(tight4, 0.2))
    one = pre.voxelGridFilter(xyzi[:50] * 1e-4, 1.0)
    assert one.shape == (1, 4) or one.shape == (2, 4) or one.shape[0] <= 8


def test_voxelgrid_box_kept_from_the_previous_cloud(orc):
    """The filter re-uses the (padded) leaf box of the previous cloud of the same leaf size: clouds that stay inside it, drift to its
    faces, jump out of it, carry a NaN, or turn dense must all give the oracle's output bit for bit."""
    from rgc_slam_amd import odometry, _lib
    p = odometry.Preprocessor(0)
    try:
        base = _scan()
        rng = np.random.default_rng(5)
        def check(xyzi, leaf):
            got, exp = p.voxelGridFilter(xyzi, leaf), orc.voxelgrid_filter(xyzi, leaf)
            assert got.shape == exp.shape and np.array_equal(got, exp)
        for leaf in (0.2, 0.3):
            check(base, leaf)                       # measures the box
            check(base, leaf)                       # inside the kept box
            check(_scan(seed=4), leaf)              # another sweep of the same volume
        shift = base.copy(); shift[:, 0] += 4.5     # 22 leaves of 0.2 m: inside the padded box, within 16 leaves of its face
        check(shift, 0.2); check(shift, 0.2)
        far = base.copy(); far[:, :3] += np.float32([60.0, -35.0, 8.0])   # outside the kept box: repeated with the measured one
        check(far, 0.2); check(far, 0.2)
        check(base, 0.3)                            # the other leaf size kept its own box
        bad = base.copy(); bad[7, 1] = np.nan
        with pytest.raises(_lib.RgcError):
            p.voxelGridFilter(bad, 0.2)
        check(far, 0.2)                             # the error left nothing behind
        dense = (rng.random((200000, 4)) * np.float32([6.0, 6.0, 2.0, 1.0])).astype(np.float32)   # 30 x 30 x 10 leaves: the dense path
        check(dense, 0.2)
        check(base, 0.2)
        few = base[:40]                             # fewer points than one wave
        check(few, 0.2)
        big = np.concatenate([base + np.float32([0.01 * k, 0, 0, 0]) for k in range(12)])          # 12 sweeps on top of each other: crowded rows
        check(big, 0.2)
    finally:
        p.close()


def test_deskew_on_device_between_set_target_and_set_source(pre):
    """rgc_deskew on device memory does not synchronise -- except when a map preparation is pending on the main stream: the scan's
    preparation (second stream) is ordered after a mark recorded BEFORE that preparation, so a de-skew enqueued behind it must have
    finished when rgc_set_source_device is called.  Same pose, bit for bit, as de-skewing on the host first."""
    import ctypes as C
    import rgc_slam_amd.synth as synth
    from rgc_slam_amd import registration
    world, tgt = synth.make_world_and_map(300000, seed=synth.SEED + 3)
    poses = synth.make_trajectory(3, seed=synth.SEED + 3)
    sc = synth.make_scan(world, poses[1], n_az=1800, seed=synth.SEED + 9)
    scan = np.concatenate([sc["xyz"], (sc["ring"] + 0.1 * sc["rel_time"])[:, None].astype(np.float32)], axis=1)
    q, t = np.float64([0.0, 0.0, 0.01, 1.0]), np.float64([0.3, 0.02, 0.0])
    q /= np.linalg.norm(q)
    expect = pre.adjustDistortion(scan, q, t)
    t4 = np.zeros((len(tgt), 4), np.float32); t4[:, :3] = tgt
    v = registration.odometer_vgicp(0)
    try:
        d_t = v.device_alloc(t4.nbytes); v.upload(d_t, t4)
        d_s = v.device_alloc(scan.nbytes)
        dp = C.POINTER(C.c_double)
        res = []
        for on_device in (True, False):
            v.upload(d_s, scan if on_device else expect)
            v.setInputTargetDevice(d_t, len(t4), 16)          # map preparation pending on the main stream
            if on_device:
                rc = v._L.rgc_deskew(v._h, C.c_void_p(d_s), len(scan), 16, q.ctypes.data_as(dp), t.ctypes.data_as(dp), 1)
                assert rc == 0
            v.setInputSourceDevice(d_s, len(scan), 16)
            v.align(poses[1].astype(np.float32), want_output=False)
            res.append(v.getFinalTransformation().copy())
            got = np.empty_like(scan)
            assert v._L.rgc_download(v._h, got.ctypes.data, C.c_void_p(d_s), got.nbytes) == 0
            assert np.array_equal(got, expect)
        assert np.array_equal(res[0], res[1])
    finally:
        v.close()


def test_deskew_vs_oracle(pre, orc):
    import rgc_slam_amd.synth as synth
    xyzi = _scan()
    R = synth.rot_zyx(0.02, 0.004, -0.003)
    # quaternion x,y,z,w of R
    qw = np.sqrt(1 + np.trace(R)) / 2
    q = np.array([(R[2, 1] - R[1, 2]) / (4 * qw), (R[0, 2] - R[2, 0]) / (4 * qw), (R[1, 0] - R[0, 1]) / (4 * qw), qw])
    t = np.array([0.15, -0.01, 0.004])
    got, exp = pre.adjustDistortion(xyzi, q, t), orc.deskew(xyzi, q, t)
    assert np.array_equal(got[:, 3], xyzi[:, 3])
    assert np.abs(got[:, :3] - exp[:, :3]).max() <= 2e-6      # fp64 math, results rounded to fp32
    # identity motion leaves the cloud unchanged; points at rel-time 1 (s = 0) do not move
    same = pre.adjustDistortion(xyzi, [0, 0, 0, 1], [0, 0, 0])
    assert np.abs(same[:, :3] - xyzi[:, :3]).max() <= 1e-6
    moved = np.linalg.norm(got[:, :3] - xyzi[:, :3], axis=1)
    rel = (xyzi[:, 3] - np.floor(xyzi[:, 3])) / 0.1
    assert moved[rel > 0.98].max() < 0.02 and moved[rel < 0.05].max() > 0.1


def test_transform_cloud_vs_oracle(pre, orc):
    xyzi = _scan(n_az=400)
    q = np.array([0.01, -0.02, 0.3, 0.95]); q /= np.linalg.norm(q)
    t = np.array([12.5, -3.25, 0.75])
    got, exp = pre.transformPointCloud(xyzi, q, t), orc.transform_cloud(xyzi, q, t)
    assert np.array_equal(got[:, 3], xyzi[:, 3])
    assert np.abs(got - exp).max() <= 4e-6
    # B9 round trip (RGC_odometer.cpp:1250-1255): world -> body with (q^-1, -q^-1 t) undoes body -> world
    qc = np.array([-q[0], -q[1], -q[2], q[3]])
    R = np.array([[1 - 2 * (q[1] ** 2 + q[2] ** 2), 2 * (q[0] * q[1] - q[2] * q[3]), 2 * (q[0] * q[2] + q[1] * q[3])],
                  [2 * (q[0] * q[1] + q[2] * q[3]), 1 - 2 * (q[0] ** 2 + q[2] ** 2), 2 * (q[1] * q[2] - q[0] * q[3])],
                  [2 * (q[0] * q[2] - q[1] * q[3]), 2 * (q[1] * q[2] + q[0] * q[3]), 1 - 2 * (q[0] ** 2 + q[1] ** 2)]])
    back = pre.transformPointCloud(got, qc, -(R.T @ t))
    assert np.abs(back[:, :3] - xyzi[:, :3]).max() < 2e-5


def test_voxelgrid_in_two_halves(orc):
    """rgc_voxelgrid_begin / rgc_voxelgrid_end on device clouds: the oracle's output bit for bit -- without a kept box (begin is the whole
    filter), on a kept box with another filter of another leaf size and of the SAME leaf size running in between, and for a cloud that has
    left the kept box (end repeats the filter); a second begin before the end is refused."""
    import ctypes as C
    from rgc_slam_amd import registration, _lib
    lib = _lib.load()
    v = registration.odometer_vgicp(0)
    try:
        base, other = _scan(), _scan(seed=4)
        far = base.copy(); far[:, :3] += np.float32([60.0, -35.0, 8.0])
        cap = max(len(base), len(other), len(far))
        d_a, d_b, d_oa, d_ob = (v.device_alloc(16 * cap) for _ in range(4))
        def begin(xyzi, d_in, d_out, leaf):
            v.upload(d_in, np.ascontiguousarray(xyzi, np.float32))
            rc = lib.rgc_voxelgrid_begin(v._h, C.c_void_p(d_in), len(xyzi), 16, C.c_float(leaf), C.c_void_p(d_out))
            return rc
        def end(d_out):
            n = C.c_int(0)
            assert lib.rgc_voxelgrid_end(v._h, C.byref(n)) == 0
            return v.download(d_out, (n.value, 4))
        def blocking(xyzi, d_in, d_out, leaf):
            v.upload(d_in, np.ascontiguousarray(xyzi, np.float32))
            n = C.c_int(0)
            assert lib.rgc_voxelgrid(v._h, C.c_void_p(d_in), len(xyzi), 16, C.c_float(leaf), C.c_void_p(d_out), C.byref(n), 1) == 0
            return v.download(d_out, (n.value, 4))
        assert begin(base, d_a, d_oa, 0.3) == 0                       # no box for 0.3 m yet: the whole filter inside begin
        assert np.array_equal(end(d_oa), orc.voxelgrid_filter(base, 0.3))
        assert begin(other, d_a, d_oa, 0.3) == 0                      # on the kept box, enqueued only
        assert begin(base, d_b, d_ob, 0.3) != 0                       # one at a time
        got_b = blocking(base, d_b, d_ob, 0.2)                        # another leaf size in between (its own box, the same scratch)
        got_c = blocking(base, d_b, d_ob, 0.3)                        # ... and the same leaf size
        assert np.array_equal(end(d_oa), orc.voxelgrid_filter(other, 0.3))
        assert np.array_equal(got_b, orc.voxelgrid_filter(base, 0.2))
        v.upload(d_b, base)
        assert np.array_equal(got_c, orc.voxelgrid_filter(base, 0.3))
        assert begin(far, d_a, d_oa, 0.3) == 0                        # outside the kept box: end repeats it on the measured one
        assert np.array_equal(end(d_oa), orc.voxelgrid_filter(far, 0.3))
        n = C.c_int(0)
        assert lib.rgc_voxelgrid_end(v._h, C.byref(n)) != 0           # nothing open
        for p in (d_a, d_b, d_oa, d_ob):
            v.device_free(p)
    finally:
        v.close()
