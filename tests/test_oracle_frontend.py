"""Oracle front-end (oracle/rgc_oracle_aux.c: orc_frontend) pinned by properties of the synthetic sensor model whose
ground truth is known (ring ids, rel-time, ground plane), since the reference holds no fixtures for it.  CPU only."""
import numpy as np


def _scan(n_az=900, seed=1):
    import rgc_slam_amd.synth as synth
    w = synth.make_world(half_extent=50.0, seed=synth.SEED)
    sc = synth.make_scan(w, np.eye(4), n_az=n_az, seed=synth.SEED + seed)
    raw = np.concatenate([sc["xyz"], sc["intensity"][:, None]], axis=1).astype(np.float32)
    return raw, sc


def test_ring_and_reltime(orc):
    raw, sc = _scan()
    o = orc.frontend(raw)
    assert o["n_cloud"] == len(raw)                                           # generator already applies the range / self filter
    assert np.array_equal(o["ring_count"], np.bincount(sc["ring"], minlength=16))
    # stable bucket: ring-major, firing order inside a ring
    order = np.argsort(sc["ring"], kind="stable")
    assert np.array_equal(o["cloud"][:, :3], raw[order, :3])
    enc = o["cloud"][:, 3]
    assert np.array_equal(np.floor(enc).astype(int), sc["ring"][order])
    rel = (enc - np.floor(enc)) / 0.1
    assert np.abs(rel - sc["rel_time"][order]).max() < 2e-3                  # relTime from azimuth, scanRegistration.cpp:206


def test_ring_and_reltime_32_beams(orc):
    """N_SCANS == 32 (scanRegistration.cpp:154-162): the truncating ring formula, the stable ring bucket and the per ring-sector caps on a
    synthetic 32-beam sweep whose ring ids are known."""
    import rgc_slam_amd.synth as synth
    w = synth.make_world(half_extent=50.0, seed=synth.SEED)
    sc = synth.make_scan(w, np.eye(4), elev_deg=synth.hdl32_elev(), n_az=700, seed=synth.SEED + 3)
    raw = np.concatenate([sc["xyz"], sc["intensity"][:, None]], axis=1).astype(np.float32)
    o = orc.frontend(raw, n_scans=32)
    assert o["n_cloud"] == len(raw)
    assert np.array_equal(o["ring_count"][:32], np.bincount(sc["ring"], minlength=32)) and o["ring_count"][:32].min() >= 0
    assert np.count_nonzero(o["ring_count"][:32]) >= 20                     # (the top beams of a 32-beam head see sky)
    order = np.argsort(sc["ring"], kind="stable")
    assert np.array_equal(o["cloud"][:, :3], raw[order, :3])
    enc = o["cloud"][:, 3]
    assert np.array_equal(np.floor(enc).astype(int), sc["ring"][order])
    assert np.abs((enc - np.floor(enc)) / 0.1 - sc["rel_time"][order]).max() < 2e-3
    assert o["n_sharp_own"] <= 32 * 6 * 20 and len(o["flat"]) <= 32 * 6 * 40 and len(o["inten"]) <= 32 * 6 * 20
    assert np.sum(o["label"] == 2) == o["n_sharp_own"] > 50 and np.sum(o["label"] == -1) == len(o["flat"]) > 500
    # the literal restatement on the same sweep: curvatures, labels, feature clouds
    from oracle import py_frontend as pf
    st = pf.stencils(o["cloud"][:, :3], raw[order, 3].astype(np.int64))
    assert np.array_equal(st["curvature"], o["curvature"]) and np.array_equal(st["inten_curvature"], o["inten_curvature"])
    sel = pf.select(o["cloud"], st, pf.occlusion(st["range"]), o["ground_marked"], o["scan_start"], o["scan_end"])
    assert np.array_equal(sel["label"], o["label"]) and np.array_equal(sel["picked"], o["picked"])
    assert np.array_equal(sel["sharp"], o["sharp"]) and np.array_equal(sel["flat"], o["flat"]) and np.array_equal(sel["inten"], o["inten"])


def test_ground_plane_and_caps(orc):
    raw, _ = _scan(n_az=1800, seed=4)
    o = orc.frontend(raw)
    g = o["groundparam"]
    assert o["ground_valid"] and abs(abs(g[2]) - 1) < 1e-3 and abs(g[9] - 0.56) < 0.02 and 0 <= g[10] < 0.5
    assert abs(np.linalg.norm(g[0:3]) - 1) < 1e-12 and abs(np.dot(g[0:3], g[3:6])) < 1e-9 and abs(np.dot(g[3:6], g[6:9])) < 1e-9
    assert np.all(o["cloud"][o["ground_marked"] == 1, 2] < 0.0)              # marked points lie on the z = -0.56 floor
    # per ring-sector caps (:493,546,601) and label bookkeeping
    assert o["n_sharp_own"] <= 16 * 6 * 20 and len(o["flat"]) <= 16 * 6 * 40 and len(o["inten"]) <= 16 * 6 * 20
    assert np.sum(o["label"] == 2) == o["n_sharp_own"] and np.sum(o["label"] == -1) == len(o["flat"])
    assert np.sum(o["label"] == 1) <= 16 * 6 and np.sum(o["inten_label"] == 2) == len(o["inten"])
    assert not np.any((o["label"] == 2) & (o["ground_marked"] == 1))         # ground points are never sharp (:490)
    # weights carried in normal_x (:501,554): distance_source in [0.7, 2.5] (+1 for sharp)
    assert np.all((o["flat"][:, 4] >= 0.7 - 1e-6) & (o["flat"][:, 4] <= 2.5 + 1e-6))
    k = o["n_sharp_own"]
    assert np.all((o["sharp"][:k, 4] >= 1.7 - 1e-6) & (o["sharp"][:k, 4] <= 3.5 + 1e-6))


def test_filters(orc):
    raw, _ = _scan(n_az=300, seed=2)
    junk = np.array([[0.1, 0.1, 0.0, 5], [-1.0, 0.2, 0.0, 5], [100.0, 0, 0, 5], [np.nan, 1, 1, 5]], np.float32)
    o1, o2 = orc.frontend(raw), orc.frontend(np.concatenate([raw[:100], junk, raw[100:]]))
    assert o1["n_cloud"] == o2["n_cloud"] and np.array_equal(o1["cloud"], o2["cloud"])   # :112-113, :732-763


def test_stencils_occlusion_and_selection_match_the_literal_restatement(orc):
    """A3 / A4 / A6 / A7 of the C oracle against oracle/py_frontend.py, written separately from the reference text: curvatures to the
    last bit, labels, suppression flags and the three feature clouds (points, order and weights) identical.  Two sweeps: plain, and
    one with near, grazing returns (the intensity smoothing branch) and strong intensity edges (the intensity corners)."""
    from oracle import py_frontend as pf
    for n_az, seed, near in ((420, 2, False), (360, 6, True)):
        raw, sc = _scan(n_az=n_az, seed=seed)
        if near:  # pull a stretch of ground returns close to the sensor and paint intensity stripes on the walls
            rng = np.random.default_rng(8)
            m = (sc["ring"] >= 5) & (sc["ring"] <= 6)
            raw[m, :3] *= np.float32(0.12)
            raw[:, 3] = np.where((np.arange(len(raw)) // 7) % 2 == 0, 20.0, 200.0).astype(np.float32) + rng.integers(0, 5, len(raw)).astype(np.float32)
            keep = (np.linalg.norm(raw[:, :3], axis=1) > 0.55) & ~((raw[:, 0] < 0) & (np.abs(raw[:, 1]) < 0.5))   # range and self filter
            raw, ring = raw[keep], sc["ring"][keep]
        else:
            ring = sc["ring"]
        o = orc.frontend(raw)
        assert o["n_cloud"] == len(raw)
        order = np.argsort(ring, kind="stable")
        assert np.array_equal(o["cloud"][:, :3], raw[order, :3])
        st = pf.stencils(o["cloud"][:, :3], raw[order, 3].astype(np.int64))       # deque<int>: truncation of the float intensity
        assert np.array_equal(st["curvature"], o["curvature"])
        assert np.array_equal(st["curvature2"], o["curvature2"])
        assert np.array_equal(st["inten_curvature"], o["inten_curvature"])
        if near:
            assert np.any((st["angle"] < 0.07) & (st["range"] < 2) & (st["angle"] > 0))   # the smoothing branch really ran
        sel = pf.select(o["cloud"], st, pf.occlusion(st["range"]), o["ground_marked"], o["scan_start"], o["scan_end"])
        assert np.array_equal(sel["label"], o["label"]) and np.array_equal(sel["inten_label"], o["inten_label"])
        assert np.array_equal(sel["picked"], o["picked"])
        assert sel["n_sharp_own"] == o["n_sharp_own"] and len(sel["flat"]) == len(o["flat"]) > 50 and len(sel["inten"]) == len(o["inten"])
        assert np.array_equal(sel["sharp"], o["sharp"]) and np.array_equal(sel["flat"], o["flat"]) and np.array_equal(sel["inten"], o["inten"])
        if near:
            assert len(o["inten"]) > 0


def test_deskew_and_transform_match_scipy(orc):
    """B2 adjustDistortion (RGC_odometer.cpp:1441-1481) and B9 transformPointCloud (:1495-1514) of the C oracle against scipy's
    Rotation / Slerp: s = 1 - frac(intensity) / 0.1, q_s = slerp(identity, q^-1, s), p' = q_s (p - s t), in double, stored float."""
    from scipy.spatial.transform import Rotation as R, Slerp
    rng = np.random.default_rng(12)
    n = 400
    pts = rng.normal(0, 10, (n, 3)).astype(np.float32)
    ring = rng.integers(0, 16, n)
    rel = rng.uniform(0, 1, n)
    inten = (ring + 0.1 * rel).astype(np.float32)
    cloud = np.concatenate([pts, inten[:, None]], axis=1).astype(np.float32)
    q = R.from_rotvec([0.02, -0.015, 0.05])
    t = np.array([0.12, -0.03, 0.01])
    out = orc.deskew(cloud, q.as_quat(), t)
    s = 1.0 - (inten.astype(np.float64) - np.floor(inten.astype(np.float64))) / 0.1
    key = R.from_quat(np.stack([[0, 0, 0, 1.0], q.inv().as_quat()]))
    qs = Slerp([0.0, 1.0], key)(np.clip(s, 0.0, 1.0))
    exp = qs.apply(pts.astype(np.float64) - s[:, None] * t)
    assert np.abs(out[:, :3] - exp.astype(np.float32)).max() < 2e-6 and np.array_equal(out[:, 3], inten)
    assert np.all((s >= -1e-6) & (s <= 1 + 1e-6))
    # transformPointCloud: q * p + t
    tw = orc.transform_cloud(cloud, q.as_quat(), t)
    assert np.abs(tw[:, :3] - (q.apply(pts.astype(np.float64)) + t).astype(np.float32)).max() < 2e-6
    assert np.array_equal(tw[:, 3], inten)
