"""The reference citations (file:line) in the header, the oracle, the kernels and the documents resolve: the cited file exists in the reference and the
cited lines are inside it (scripts/check_citations.py).  Needs /root/reference, which exists in the build container only: skipped elsewhere (nothing on the
GPU box may read it).  What a citation SAYS about its lines stays a reviewer's job; that it points somewhere real need not be."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="the reference tree is only present in the build container")
def test_reference_citations_resolve():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "check_citations.py")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-3000:]
    n = int(r.stdout.split()[0])
    assert n >= 600, r.stdout[:300]
