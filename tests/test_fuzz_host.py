"""tests/fuzz/fuzz_host.py as a test: the frame body's scalar host stages (pose fusion, composition, extraction, gyro pre-integration, the
attitude filter, the ground gate, R2ypr / ypr2R) against oracle/py_fusion.py on random inputs.  No GPU needed."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_fuzz_of_the_host_stages():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fuzz", "fuzz_host.py"), "400", "7"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-1500:]
    rep = json.loads(r.stdout.strip().splitlines()[-1])
    assert rep["trials"] == 400 and rep["failures"] == [], rep["failures"][:5]
    assert rep["max"]["compose"] < 1e-12 and rep["max"]["filter"] < 1e-12 and rep["max"]["gate"] < 1e-12 and rep["max"]["preintegrate"] < 1e-12
