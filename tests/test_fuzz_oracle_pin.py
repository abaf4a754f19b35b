"""tests/fuzz/fuzz_oracle_pin.py as a test: the C oracle (oracle/rgc_oracle*.c) against the literal numpy / scipy restatement that pins it
(oracle/py_oracle.py), on random clouds instead of the committed fixtures' fixed ones -- neighbour sets, covariances under every
RegularizationMethod, the voxel table under every VoxelAccumulationMode, a linearisation, the solve's final pose, the fitness, the leaf filter.
No GPU needed.  (The reference cannot be built here: this pins the oracle to the builder's second restatement, not to the reference's binary --
DESIGN.md 3, "parity unpinned".)"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_fuzz_of_the_oracle_against_its_literal_restatement():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fuzz", "fuzz_oracle_pin.py"), "24", "11"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-1500:]
    rep = json.loads(r.stdout.strip().splitlines()[-1])
    assert rep["trials"] == 24 and rep["failures"] == [], rep["failures"][:5]
    m = rep["max"]
    assert m["cov"] <= 1e-9 and m["vox_mean"] <= 1e-9 and m["H_rel"] <= 1e-7 and m["b_rel"] <= 1e-6 and m["cost_rel"] <= 1e-7 and m["leaf_filter"] <= 1e-5, m


def test_fuzz_of_the_oracle_front_end_against_its_literal_restatement():
    """tests/fuzz/fuzz_oracle_pin_frontend.py: orc_frontend against oracle/py_frontend.py on random 16 / 32 / 64-beam sweeps -- ring bucket,
    curvatures bit for bit, ground marks / points / plane, suppression flags, labels, the three feature clouds."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fuzz", "fuzz_oracle_pin_frontend.py"), "40", "11"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-1500:]
    rep = json.loads(r.stdout.strip().splitlines()[-1])
    assert rep["trials"] == 40 and rep["failures"] == [], rep["failures"][:5]
    assert set(rep["by_beams"]) == {"16", "32", "64"} and rep["with_ground_plane"] >= 10 and rep["smoothing_branch_ran"] >= 5
    assert rep["ring_buckets_bit_for_bit"] == 40          # A2 against oracle/py_frontend.ring_bucket, glibc's float libm on both sides
    assert rep["max"]["ground_normal"] < 1e-7 and rep["max"]["ground_distance"] < 1e-8


def _campaign(script, *args, timeout=900):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fuzz", script), *args], capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stderr[-1500:]
    return json.loads(r.stdout.strip().splitlines()[-1])


def test_fuzz_of_the_oracle_icp_against_its_literal_restatement():
    """f4: orc_icp_align against oracle/py_icp.py on random maps, drifts, gates and iteration caps: termination state, iteration count, pose."""
    rep = _campaign("fuzz_oracle_pin_icp.py", "60", "11")
    assert rep["trials"] == 60 and rep["failures"] == [], rep["failures"][:5]
    assert rep["max"]["T"] < 1e-6 and len(rep["by_state"]) >= 2


def test_fuzz_of_the_oracle_mapping_registration_against_its_literal_restatement():
    """f1: orc_mapreg_* against oracle/py_mapreg.py (finite-difference Jacobians) on random cases with and without the ground / IMU blocks."""
    rep = _campaign("fuzz_oracle_pin_mapreg.py", "8", "11")
    assert rep["trials"] == 8 and rep["failures"] == [], rep["failures"][:5]
    assert rep["max"]["x"] < 1e-6 and rep["max"]["initial_cost_rel"] < 1e-9


def test_fuzz_of_the_frame_body_mirror_against_the_literal_restatement():
    """The frame body: oracle/py_odometer.py against rgc_slam_amd.odometry.Odometer, both on the oracle's stages, on random sequences."""
    rep = _campaign("fuzz_oracle_pin_sequence.py", "3", "11")
    assert rep["trials"] == 3 and rep["failures"] == [] and rep["sweeps"] >= 15, rep["failures"][:5]
    assert rep["max"]["q"] < 1e-6 and rep["max"]["t"] < 1e-6
