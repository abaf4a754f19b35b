"""A map's covariances are a function of the cloud alone: whichever route it comes by (rgc_set_target_reframed, rgc_transform_cloud +
rgc_set_target_device), whichever kernel ends up searching a query (the dense search, the sparse map's four-lane search, the cooperative
kernel -- that depends on the grid's box, which the routes derive differently) and for any k.  Found by tests/fuzz/fuzz_modes.py: the four-lane
search and the general-k branch of the dense one used to sum a neighbourhood in key order, the cooperative kernel in ascending position."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def reg_mod():
    from rgc_slam_amd import registration
    return registration


def _sheets(n, rng):
    """two planes and a pole, a millimetre of noise: 0.2 points per 1 m cell (the sparse-map launch), degenerate neighbourhoods"""
    a = np.c_[rng.uniform(-20, 20, (n // 2, 2)), np.zeros(n // 2)]
    b = np.c_[rng.uniform(-20, 20, n // 3), np.full(n // 3, 3.0), rng.uniform(0, 6, n // 3)]
    m = n - n // 2 - n // 3
    c = np.c_[np.full(m, 1.5), np.full(m, -2.0), rng.uniform(0, 8, m)]
    p = np.vstack([a, b, c]) + rng.normal(0, 1e-3, (n, 3))
    return p.astype(np.float32)[rng.permutation(n)]


@pytest.mark.parametrize("k", [20, 10, 25])
@pytest.mark.parametrize("kind", ["sheets", "synth"])
def test_two_routes_same_bits(reg_mod, orc, k, kind):
    import rgc_slam_amd.synth as synth
    import bench
    rng = np.random.default_rng(17)
    pts = _sheets(7086, rng) if kind == "sheets" else synth.make_world_and_map(30000, seed=5)[1].astype(np.float32)
    n = len(pts)
    a = np.zeros((n, 4), np.float32)
    a[:, :3] = pts
    v, w = reg_mod.odometer_vgicp(0), reg_mod.odometer_vgicp(0)
    for x in (v, w):
        x.setCorrespondenceRandomness(k)
        x.setNeighbourReuse(0)
    dv, sv, dw, sw = (v.device_alloc(a.nbytes) for _ in range(4))
    v.upload(dv, a); w.upload(dw, a)
    deferred = []
    for yaw, t in ((2.1, [31.0, -12.0, 0.4]), (-0.6, [-3.0, 44.0, 0.0])):
        Tw = synth.se3(synth.rot_zyx(yaw, 0.02, -0.01), t)
        q, tt = bench.world_to_body(Tw)
        v.setInputTargetReframed(dv, n, 16, q, tt, sv)
        w.transformCloudDevice(dw, n, 16, q, tt, sw)
        w.setInputTargetDevice(sw, n, 16)
        cv, cw = v.getTargetCovariances(), w.getTargetCovariances()
        assert np.array_equal(cv, cw)
        xv, xw = v.getVoxels(), w.getVoxels()
        kv, kw = np.lexsort(xv["coords"].T[::-1]), np.lexsort(xw["coords"].T[::-1])
        for key in ("coords", "num", "mean", "cov"):
            assert np.array_equal(xv[key][kv], xw[key][kw])
        deferred.append((v.stats()["deferred_target"], w.stats()["deferred_target"]))
    body = v.download(sv, (n, 4))
    oc, _ = orc.covariances(body[:, :3].copy(), k=k)
    assert np.abs(cv - oc).max() <= 1e-9
    if kind == "sheets":   # the two routes' boxes differ, and with them which queries go to the cooperative kernel: the case the equality is about
        assert any(x != y for x, y in deferred), deferred
    for p in (dv, sv):
        v.device_free(p)
    for p in (dw, sw):
        w.device_free(p)
    v.close(); w.close()


def test_fuzz_campaign(reg_mod):
    """tests/fuzz/fuzz_modes.py, a short campaign: lattices with exact ties, repeated points, sheets, a clump in a sparse field, uniform noise;
    nothing kept / seeds / lists / lazy / transform + device; k = 10, 20, 25; leaf 0.5, 1, 2 m; edits between frames -- covariances, voxel
    tables and solves bit for bit across the routes, covariances against the oracle."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fuzz", "fuzz_modes.py"), "60", "11", "40000"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    rep = json.loads(r.stdout.strip().splitlines()[-1])
    assert rep["trials"] == 60 and rep["failures"] == [], rep["failures"][:5]
    assert rep["solves"] >= 180 and rep["oracle_checks"] >= 20


def test_fuzz_against_the_oracle(reg_mod):
    """tests/fuzz/fuzz_oracle.py, a short campaign: whole registrations -- both covariance sets, the voxel table, a linearisation at the guess, the
    solve's pose, the fitness -- against the oracle, on random clouds, k = 10 / 20 / 25, leaf 0.5 / 1 / 2 m and every RegularizationMethod /
    VoxelAccumulationMode of the reference's interface."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fuzz", "fuzz_oracle.py"), "80", "21", "30000"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    rep = json.loads(r.stdout.strip().splitlines()[-1])
    assert rep["trials"] == 80 and rep["failures"] == [], rep["failures"][:5]
    assert rep["general_route_trials"] >= 5 and rep["max"]["dt"] <= 1e-4


def test_fuzz_of_the_stages_in_front(reg_mod):
    """tests/fuzz/fuzz_pre.py, a short campaign: the front-end (16 / 32 / 64 beams; points dropped, NaNs, shuffled firing order, truncated,
    out of range), the leaf filter through one object (sweeps, noise, lattices on leaf boundaries, NaNs refused), de-skew and re-framing --
    against the oracle, stage by stage."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fuzz", "fuzz_pre.py"), "60", "31"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    rep = json.loads(r.stdout.strip().splitlines()[-1])
    assert rep["trials"] == 60 and rep["failures"] == [], rep["failures"][:5]
    assert rep["frontend"] >= 55 and rep["voxelgrid"] == 60 and rep["max"]["deskew"] <= 4e-6


def test_fuzz_of_the_frame_body(reg_mod):
    """tests/fuzz/fuzz_sequence.py, a short campaign: random worlds, trajectories (some up a ramp), azimuth counts, with and without the IMU path --
    the frame body on the library against the same frame body on the oracle's stages, frame by frame: 1e-4 m / 1e-4 rad until a solve runs
    out of iterations or stops an iteration apart on the two sides (a flat valley: both ends are as good), 2e-3 from there on."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fuzz", "fuzz_sequence.py"), "12", "41", "8"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    rep = json.loads(r.stdout.strip().splitlines()[-1])
    assert rep["trials"] == 12 and rep["failures"] == [], rep["failures"][:5]
    assert rep["frames"] == 96 and rep["max"]["dt_strict"] <= 1e-4


def test_fuzz_of_the_resident_map_and_the_next_rows(reg_mod):
    """tests/fuzz/fuzz_map.py (f2: random insert / evict / rebase / commit sequences, stored points and committed target bit for bit with the
    oracle's composition) and tests/fuzz/fuzz_next_rows.py (f4 loop-closure ICP, f1 mapping-node feature registration), short campaigns."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fuzz", "fuzz_map.py"), "25", "51", "25"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    rep = json.loads(r.stdout.strip().splitlines()[-1])
    assert rep["trials"] == 25 and rep["failures"] == [] and rep["commits_compared"] >= 60, rep["failures"][:5]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fuzz", "fuzz_next_rows.py"), "40", "4", "61"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    rep = json.loads(r.stdout.strip().splitlines()[-1])
    assert rep["icp_trials"] == 40 and rep["mapreg_trials"] == 4 and rep["failures"] == [], rep["failures"][:5]


def test_lazy_target_replaced_before_any_solve(reg_mod):
    """Found by tests/fuzz/fuzz_api.py: a lazy target A, a scan, a lazy target B that leaves A's (speculative) grid, then a getter.  The getter
    used to complete B first -- the completion's kernels leave at once on the tripped guard -- and resolve the guard second, which put an
    UNBUILT lazy target back: every covariance it returned was stale.  Guards first now (validate_clouds)."""
    import rgc_slam_amd.synth as synth
    world, base = synth.make_world_and_map(20000, seed=5)
    base = base.astype(np.float32)
    A, B = base[:5245], base[3000:11368]
    src = B[::3][:1500] + np.float32(0.02)
    w = reg_mod.odometer_vgicp(0)
    w.setInputTarget(B)
    full, fullv = w.getTargetCovariances(), w.getVoxels()
    w.setInputSource(src)
    w.align(np.eye(4, dtype=np.float32), want_output=False, want_fitness=True)
    for getter in ("covariances", "voxels", "solve"):
        v = reg_mod.odometer_vgicp(0)
        v.setLazyTarget(3)
        v.setInputTarget(A); v.setInputSource(src); v.setInputTarget(B)
        if getter == "covariances":
            assert np.array_equal(v.getTargetCovariances(), full)
        elif getter == "voxels":
            x = v.getVoxels()
            ka, kb = np.lexsort(x["coords"].T[::-1]), np.lexsort(fullv["coords"].T[::-1])
            assert all(np.array_equal(x[k][ka], fullv[k][kb]) for k in ("coords", "num", "mean", "cov"))
        else:
            v.align(np.eye(4, dtype=np.float32), want_output=False, want_fitness=True)
            assert np.array_equal(v.getFinalTransformation(), w.getFinalTransformation()) and v.getFitnessScore() == w.getFitnessScore()
        v.close()
    w.close()


def test_a_solve_in_flight_is_one_on_either_route(reg_mod):
    """rgc_align_begin on the general route runs the solve at once and keeps its result for rgc_align_end; until then the context refuses
    what it refuses with a solve in flight on the tuned route (tests/fuzz/fuzz_api.py: it used to accept new clouds, and a later
    rgc_align_end then found nothing)."""
    import rgc_slam_amd.synth as synth
    world, tgt = synth.make_world_and_map(15000, seed=9)
    tgt = tgt.astype(np.float32)
    src = tgt[::5][:2000] + np.float32(0.03)
    for method in (reg_mod.FastVGICP.REG_PLANE, reg_mod.FastVGICP.REG_MIN_EIG):
        v = reg_mod.odometer_vgicp(0)
        v.setRegularizationMethod(method)
        v.setInputTarget(tgt); v.setInputSource(src)
        v.align_begin(np.eye(4, dtype=np.float32))
        for refused in (lambda: v.setInputSource(src), lambda: v.setInputTarget(tgt), lambda: v.align_begin(np.eye(4, dtype=np.float32)),
                        lambda: v.align(np.eye(4, dtype=np.float32), want_output=False), lambda: v.clearSource(), lambda: v.setLazyTarget(2),
                        lambda: v.getTargetCovariances()):
            with pytest.raises(reg_mod.RgcError):
                refused()
        T = v.align_end()
        with pytest.raises(reg_mod.RgcError):
            v.align_end()
        v.align(np.eye(4, dtype=np.float32), want_output=False)
        assert np.array_equal(T, v.getFinalTransformation())
        v.close()


def test_a_cleared_cloud_leaves_no_guard_behind(reg_mod):
    """Found by tests/fuzz/fuzz_api.py: a scan prepared on a speculative grid, rgc_clear_source, then a getter of the target: the cleared scan's
    stale guard made the library prepare it "again" -- zero points, a launch of zero workgroups, RGC_ERR_HIP out of a getter."""
    import rgc_slam_amd.synth as synth
    world, base = synth.make_world_and_map(30000, seed=5)
    base = base.astype(np.float32)
    v = reg_mod.odometer_vgicp(0)
    v.setInputTarget(base[:20000])
    v.setInputSource(base[100:3000])            # measures its box
    v.setInputSource(base[20000:26000] + np.float32([40.0, 0, 0]))   # takes the previous grid speculatively -- and leaves it
    v.clearSource()
    v.setInputTarget(base[5000:21000])          # (the target's own grid is speculative too: the getter below goes through the guards)
    c = v.getTargetCovariances()
    w = reg_mod.odometer_vgicp(0)
    w.setInputTarget(base[5000:21000])
    assert np.array_equal(c, w.getTargetCovariances())
    v.clearTarget()
    v.setInputSource(base[100:3000])
    assert len(v.getSourceCovariances()) == 2900
    v.close(); w.close()


def test_a_refused_commit_writes_nothing(reg_mod):
    """Found by tests/fuzz/fuzz_api.py: rgc_map_commit with a solve in flight ran its leaf filter INTO the buffer the resident target had been set
    from and was refused only behind it (by the target's setter): the bound target kept a rewritten input, rgc_map_download(1) returned
    another cloud.  Refused before anything is written now."""
    import rgc_slam_amd.synth as synth
    from rgc_slam_amd import local_map
    world, base = synth.make_world_and_map(20000, seed=5)
    base = base.astype(np.float32)
    v = reg_mod.odometer_vgicp(0)
    lm = local_map.RollingLocalMap(v)
    lm.reset(None)
    for k in range(3):
        a = np.zeros((3000, 4), np.float32); a[:, :3] = base[k * 3000:(k + 1) * 3000]
        lm.insert(a, np.array([0, 0, 0, 1.0]), np.zeros(3))
    n0 = lm.commit(0.3)
    t0 = lm.target().copy()
    v.setInputSource(base[:2000] + np.float32(0.02))
    v.align_begin(np.eye(4, dtype=np.float32))
    with pytest.raises(reg_mod.RgcError):
        lm.commit(0.5)                                   # another leaf: a real commit -- refused, and nothing written
    assert lm.commit(0.3) == n0                          # the unchanged map at the same leaf: a no-op, allowed
    assert np.array_equal(lm.target(), t0)
    T = v.align_end()
    v.align(np.eye(4, dtype=np.float32), want_output=False)
    assert np.array_equal(T, v.getFinalTransformation())
    assert lm.commit(0.5) != n0 and not np.array_equal(lm.target()[:100], t0[:100])
    v.close()


def test_a_pose_that_is_none_is_refused(reg_mod):
    """Found by tests/fuzz/fuzz_api.py: a solve that ends in NaN (a degenerate problem) handed its pose on through rgc_align_end_reframe; the
    re-framed map's box came out NaN, the float -> int conversions behind it are undefined, and the preparation asked for 40 petabytes.
    rgc_set_target_reframed refuses a pose that is not finite (and a zero quaternion); the solve's own outputs still come back."""
    import rgc_slam_amd.synth as synth
    world, base = synth.make_world_and_map(12000, seed=5)
    a = np.zeros((len(base), 4), np.float32); a[:, :3] = base
    v = reg_mod.odometer_vgicp(0)
    d, s = v.device_alloc(a.nbytes), v.device_alloc(a.nbytes)
    v.upload(d, a)
    for q, t in (([0, 0, 0, 1.0], [np.nan, 0, 0]), ([np.nan, 0, 0, 1.0], [0, 0, 0]), ([0, 0, 0, 0.0], [0, 0, 0]), ([0, 0, 0, 1.0], [np.inf, 0, 0])):
        with pytest.raises(reg_mod.RgcError):
            v.setInputTargetReframed(d, len(a), 16, np.array(q, float), np.array(t, float), s)
    v.setInputTargetReframed(d, len(a), 16, np.array([0, 0, 0, 1.0]), np.zeros(3), s)      # the context is none the worse for it
    assert len(v.getTargetCovariances()) == len(a)
    v.device_free(d); v.device_free(s); v.close()


def test_settings_and_a_borrowed_target(reg_mod):
    """Found by tests/fuzz/fuzz_api.py (2 000 trials in): rgc_set_params with another leaf size or k prepares the context's clouds again from
    their inputs -- including a BORROWED target (rgc_share_target), whose input is the owner's and may be long gone: a memory fault on the
    device.  The alias is dropped instead (share again); and the same settings are refused with a solve in flight."""
    import rgc_slam_amd.synth as synth
    world, base = synth.make_world_and_map(20000, seed=5)
    base = base.astype(np.float32)
    a, b = reg_mod.odometer_vgicp(0), reg_mod.odometer_vgicp(0)
    a.setInputTarget(base[:8000])
    b.shareTargetFrom(a)
    for k in range(4):                       # the owner moves on: its old input buffer is re-used
        a.setInputTarget(base[2000 * k: 2000 * k + 6000])
    b.setInputSource(base[:3000])
    b.setResolution(2.0)                     # used to fault; now: the alias is gone
    b.synchronize()
    with pytest.raises(reg_mod.RgcError):
        b.align(np.eye(4, dtype=np.float32), want_output=False)
    a.setResolution(2.0)
    b.shareTargetFrom(a)
    b.align(np.eye(4, dtype=np.float32), want_output=False)
    a.setInputSource(base[:3000])
    a.align(np.eye(4, dtype=np.float32), want_output=False)
    assert np.array_equal(a.getFinalTransformation(), b.getFinalTransformation())
    a.align_begin(np.eye(4, dtype=np.float32))
    with pytest.raises(reg_mod.RgcError):
        a.setResolution(1.0)                 # the clouds would be prepared again under the running solve
    a.align_end()
    a.close(); b.close()


def test_a_map_bound_target_swapped_away(reg_mod):
    """Found by tests/fuzz/fuzz_api.py: a target committed from the resident map is set from the map's own filter output buffer; swapped into
    the scan it kept pointing there, the next commit overwrote the buffer, and swapped back it was prepared from another cloud's points
    (its voxel table missed voxels).  The swap gives it a copy of its own."""
    import rgc_slam_amd.synth as synth
    from rgc_slam_amd import local_map
    world, base = synth.make_world_and_map(20000, seed=5)
    base = base.astype(np.float32)
    rng = np.random.default_rng(2)

    def vox(v):
        x = v.getVoxels()
        o = np.lexsort(x["coords"].T[::-1])
        return {k: x[k][o] for k in ("coords", "num", "mean", "cov")}
    v = reg_mod.odometer_vgicp(0)
    lm = local_map.RollingLocalMap(v)
    lm.reset(None)
    kf = np.zeros((4000, 4), np.float32); kf[:, :3] = base[rng.choice(len(base), 4000, replace=False)]
    lm.insert(kf, np.array([0, 0, 0, 1.0]), np.zeros(3))
    lm.commit(0.5)
    T1 = lm.target()[:, :3].copy()
    v.setInputSource(T1 + np.float32(0.01))
    v.swapSourceAndTarget()                  # the map's target is the scan now
    lm.commit(0.3)                           # ... and the map writes its buffer again
    v.swapSourceAndTarget()                  # back: must still be T1
    f = reg_mod.odometer_vgicp(0)
    f.setInputTarget(T1)
    x, r = vox(v), vox(f)
    assert all(np.array_equal(x[k], r[k]) for k in ("coords", "num", "mean", "cov"))
    assert np.array_equal(v.getTargetCovariances(), f.getTargetCovariances())
    v.close(); f.close()


def test_fuzz_of_the_call_sequences(reg_mod):
    """tests/fuzz/fuzz_api.py, a short campaign: random sequences of the registration's calls against a model of what must work and what must be
    refused, every solve against a fresh context's."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fuzz", "fuzz_api.py"), "40", "71", "45"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    rep = json.loads(r.stdout.strip().splitlines()[-1])
    assert rep["trials"] == 40 and rep["failures"] == [], rep["failures"][:5]
    assert rep["solves_compared"] >= 40 and rep["refusals_expected"] >= 200


def test_fuzz_of_the_message_layouts(reg_mod):
    """tests/fuzz/fuzz_wire.py: PointCloud2 unpacking against numpy's structured dtypes over random layouts (steps, offsets, every datatype for
    every field, both byte orders, strict / converting), pack -> unpack round trips, bad layouts refused."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fuzz", "fuzz_wire.py"), "600", "81"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    rep = json.loads(r.stdout.strip().splitlines()[-1])
    assert rep["trials"] == 600 and rep["failures"] == [] and rep["refused_as_expected"] == 3 and rep["round_trips"] == 120, rep["failures"][:5]


def test_degenerate_clouds(reg_mod):
    """tests/fuzz/fuzz_degenerate.py: every point the same, a line, a lattice sheet, clumps 100 km apart, coordinates of 1e6 m, exactly k / k + 1 /
    k - 1 points, a NaN, an inf, an outlier, an empty cloud -- as target, as scan, registered to themselves: a clean refusal or a finite
    result, the oracle's covariances where they are defined."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fuzz", "fuzz_degenerate.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    rep = json.loads(r.stdout)
    assert rep["failures"] == [] and len(rep["cases"]) >= 15, rep["failures"]
    assert rep["cases"]["one NaN"]["target"].startswith("refused") and rep["cases"]["k - 1 points"]["source"].startswith("refused")
    assert "vs oracle" in rep["cases"]["coordinates of 1e6 m"]["target"]


def test_everything_on_one_context(reg_mod):
    """tests/fuzz/fuzz_one_context.py, a short campaign: the front-end, the leaf filter (blocking and in two halves), de-skew, re-framing, the
    registration and the wire unpacking interleaved at random on ONE context, each result against the same call on a context of its own."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fuzz", "fuzz_one_context.py"), "8", "91", "40"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    rep = json.loads(r.stdout.strip().splitlines()[-1])
    assert rep["trials"] == 8 and rep["failures"] == [] and rep["compared"] >= 150, rep["failures"][:5]


def test_fuzz_of_the_dependent_sequence(reg_mod):
    """tests/fuzz/fuzz_dependent.py, a short campaign: the headline's path -- rgc_set_target_reframed / rgc_align_end_reframe on one and two
    contexts, the Python and the C++ frame loop, random maps, scans, lengths, reuse modes and lazy margins -- against the plain calls
    (rgc_transform_cloud, rgc_set_target_device, rgc_set_source_device, rgc_align, the world pose composed by hand): bit for bit."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fuzz", "fuzz_dependent.py"), "12", "101"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    rep = json.loads(r.stdout.strip().splitlines()[-1])
    assert rep["trials"] == 12 and rep["failures"] == [] and rep["variants_compared"] == 48, rep["failures"][:5]


def test_every_entry_point_with_nothing(reg_mod):
    """tests/fuzz/fuzz_null_args.py: every prototype of include/rgc_hip.h called with NULL for every pointer and 0 for every number -- without a
    context, with a fresh one, with one that holds clouds: statuses, no crash (rgc_R2ypr / rgc_ypr2R used to dereference)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fuzz", "fuzz_null_args.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-1500:]
    rep = json.loads(r.stdout.strip().splitlines()[-1])
    assert rep["functions"] >= 80 and rep["failures"] == [] and rep["context_still_works"] and rep["create_with_null_out"] != 0


def test_one_bad_argument_at_a_time(reg_mod):
    """tests/fuzz/fuzz_bad_args.py: 33 entry points, each argument in turn NULL / 0 / -1 / 2^30 / NaN / inf on a context that holds clouds:
    statuses, no crash, and the context solves afterwards as before.  (It found that a HIP error reported by one call surfaced AGAIN from the
    next call's hipGetLastError -- a 1 GiB "stride" made an allocation fail -- and that strides and point counts had no upper bound.)"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fuzz", "fuzz_bad_args.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-1500:]
    rep = json.loads(r.stdout.strip().splitlines()[-1])
    assert rep["functions"] >= 33 and rep["calls"] >= 300 and rep["failures"] == [] and rep["same_result_afterwards"], rep["failures"]


def test_the_pointer_checker(reg_mod, monkeypatch):
    """RGC_CHECK_POINTERS=1 (an integrator's aid, off by default): what a caller calls a device buffer is looked up first -- a host pointer, a
    count larger than the allocation: RGC_ERR_INVALID where there would be a memory fault on the device; correct calls are untouched."""
    import rgc_slam_amd.synth as synth
    monkeypatch.setenv("RGC_CHECK_POINTERS", "1")
    world, base = synth.make_world_and_map(6000, seed=5)
    a = np.zeros((len(base), 4), np.float32); a[:, :3] = base
    v = reg_mod.odometer_vgicp(0)
    d, s = v.device_alloc(a.nbytes), v.device_alloc(a.nbytes)
    small = v.device_alloc(16 * 100)
    v.upload(d, a)
    v.setInputTargetDevice(d, len(a), 16)                                          # correct: works
    ref = v.getTargetCovariances()
    for bad in (lambda: v.setInputTargetDevice(a.ctypes.data, len(a), 16),         # a host pointer called "device"
                lambda: v.setInputSourceDevice(a.ctypes.data, len(a), 16),
                lambda: v.setInputTargetDevice(small, len(a), 16),                 # 100 points allocated, 6 000 claimed
                lambda: v.setInputTargetDevice(d + 16 * 10, len(a), 16),           # an offset into the allocation: ten points short
                lambda: v.setInputTargetReframed(d, len(a), 16, np.array([0, 0, 0, 1.0]), np.zeros(3), small),
                lambda: v.setInputTargetReframed(a.ctypes.data, len(a), 16, np.array([0, 0, 0, 1.0]), np.zeros(3), s)):
        with pytest.raises(reg_mod.RgcError):
            bad()
    v.setInputTargetDevice(d + 16 * 10, len(a) - 10, 16)                           # the same offset with the right count: fine
    v.setInputTargetReframed(d, len(a), 16, np.array([0, 0, 0, 1.0]), np.zeros(3), s)
    assert np.array_equal(v.getTargetCovariances(), ref)
    for p_ in (d, s, small):
        v.device_free(p_)
    v.close()


def test_threads_on_contexts_of_their_own(reg_mod):
    """tests/fuzz/fuzz_threads.py: eight host threads, each with contexts of its own, run random registrations, leaf filters and front-ends at
    the same time; every result is the one of the same call made alone."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fuzz", "fuzz_threads.py"), "8", "20", "111"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-1500:]
    rep = json.loads(r.stdout.strip().splitlines()[-1])
    assert rep["jobs"] == 160 and rep["failures"] == [], rep["failures"][:5]
