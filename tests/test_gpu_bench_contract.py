"""bench.py's contract on a real GPU, at a reduced size: ONE JSON line with the required keys, and the same trajectory whether
`--gpus 1` is run directly or as one rank under torch.distributed.run (what the driver does for N > 1)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ARGS = ["--steps", "6", "--warmup", "2", "--n-target", "150000", "--no-cpu-baseline", "--configs", "none"]


def _line(cmd):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    return json.loads(lines[0])


def test_bench_line_direct_and_under_torchrun():
    a = _line([sys.executable, "bench.py", "--gpus", "1"] + ARGS)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "final_pose_checksum"):
        assert k in a, k
    assert a["n_gpus"] == 1 and a["steps"] == 6 and a["warmup"] == 2 and a["value"] > 0
    assert a["one_frame_at_a_time"]["same_poses"] and a["scan_h2d_and_output"]["same_final_pose"]
    r = a["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-6
    b = _line([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
               "--master-port", "29611", "bench.py", "--gpus", "1"] + ARGS)
    assert b["n_gpus"] == 1
    assert b["final_pose_checksum"] == a["final_pose_checksum"]
