"""A map's covariances are a function of the cloud alone: whichever route it comes by (rgc_set_target_reframed, rgc_transform_cloud +
rgc_set_target_device), whichever kernel ends up searching a query (the dense search, the sparse map's four-lane search, the cooperative
kernel -- that depends on the grid's box, which the routes derive differently) and for any k.  Found by scripts/fuzz_modes.py: the four-lane
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
    """scripts/fuzz_modes.py, a short campaign: lattices with exact ties, repeated points, sheets, a clump in a sparse field, uniform noise;
    nothing kept / seeds / lists / lazy / transform + device; k = 10, 20, 25; leaf 0.5, 1, 2 m; edits between frames -- covariances, voxel
    tables and solves bit for bit across the routes, covariances against the oracle."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "fuzz_modes.py"), "60", "11", "40000"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    rep = json.loads(r.stdout.strip().splitlines()[-1])
    assert rep["trials"] == 60 and rep["failures"] == [], rep["failures"][:5]
    assert rep["solves"] >= 180 and rep["oracle_checks"] >= 20


def test_fuzz_against_the_oracle(reg_mod):
    """scripts/fuzz_oracle.py, a short campaign: whole registrations -- both covariance sets, the voxel table, a linearisation at the guess, the
    solve's pose, the fitness -- against the oracle, on random clouds, k = 10 / 20 / 25, leaf 0.5 / 1 / 2 m and every RegularizationMethod /
    VoxelAccumulationMode of the reference's interface."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "fuzz_oracle.py"), "80", "21", "30000"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    rep = json.loads(r.stdout.strip().splitlines()[-1])
    assert rep["trials"] == 80 and rep["failures"] == [], rep["failures"][:5]
    assert rep["general_route_trials"] >= 5 and rep["max"]["dt"] <= 1e-4


def test_fuzz_of_the_stages_in_front(reg_mod):
    """scripts/fuzz_pre.py, a short campaign: the front-end (16 / 32 / 64 beams; points dropped, NaNs, shuffled firing order, truncated,
    out of range), the leaf filter through one object (sweeps, noise, lattices on leaf boundaries, NaNs refused), de-skew and re-framing --
    against the oracle, stage by stage."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "fuzz_pre.py"), "60", "31"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    rep = json.loads(r.stdout.strip().splitlines()[-1])
    assert rep["trials"] == 60 and rep["failures"] == [], rep["failures"][:5]
    assert rep["frontend"] >= 55 and rep["voxelgrid"] == 60 and rep["max"]["deskew"] <= 4e-6


def test_fuzz_of_the_frame_body(reg_mod):
    """scripts/fuzz_sequence.py, a short campaign: random worlds, trajectories (some up a ramp), azimuth counts, with and without the IMU path --
    the frame body on the library against the same frame body on the oracle's stages, frame by frame: 1e-4 m / 1e-4 rad until a solve runs
    out of iterations or stops an iteration apart on the two sides (a flat valley: both ends are as good), 2e-3 from there on."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "fuzz_sequence.py"), "12", "41", "8"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    rep = json.loads(r.stdout.strip().splitlines()[-1])
    assert rep["trials"] == 12 and rep["failures"] == [], rep["failures"][:5]
    assert rep["frames"] == 96 and rep["max"]["dt_strict"] <= 1e-4


def test_fuzz_of_the_resident_map_and_the_next_rows(reg_mod):
    """scripts/fuzz_map.py (f2: random insert / evict / rebase / commit sequences, stored points and committed target bit for bit with the
    oracle's composition) and scripts/fuzz_next_rows.py (f4 loop-closure ICP, f1 mapping-node feature registration), short campaigns."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "fuzz_map.py"), "25", "51", "25"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    rep = json.loads(r.stdout.strip().splitlines()[-1])
    assert rep["trials"] == 25 and rep["failures"] == [] and rep["commits_compared"] >= 60, rep["failures"][:5]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "fuzz_next_rows.py"), "40", "4", "61"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    rep = json.loads(r.stdout.strip().splitlines()[-1])
    assert rep["icp_trials"] == 40 and rep["mapreg_trials"] == 4 and rep["failures"] == [], rep["failures"][:5]
