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
