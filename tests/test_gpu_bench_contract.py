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
    # `value` keeps nothing from one frame's target to the next: every query of every frame is searched
    assert a["config"]["knn_reuse"] == "none" and a["config"]["queries_searched_per_frame"] == 150000 and r["queries_searched_per_launch"] == 150000
    # the library's default beside it: the same poses, and on edited maps the poses of the same edits with nothing kept
    ru = a["reuse_of_an_unchanged_map"]
    assert ru["unchanged_map"]["same_poses_as_value"] and ru["seeds_only_unchanged_map"]["same_poses_as_value"]
    assert ru["unchanged_map"]["queries_searched_per_launch"] < 0.2 * 150000
    assert ru["one_point_edited_every_frame"]["same_poses_as_with_nothing_kept"] and ru["keyframe_every_3rd_frame"]["same_poses_as_with_nothing_kept"]
    sp = a["sequences_per_gpu"]
    assert "error" not in sp, sp
    assert [x["S"] for x in sp["runs"]] == [1, 2, 4, 8] and all(x["poses_equal_each_sequence_alone"] for x in sp["runs"])
    b = _line([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
               "--master-port", "29611", "bench.py", "--gpus", "1", "--no-two-sequences"] + ARGS)
    assert b["n_gpus"] == 1
    assert b["final_pose_checksum"] == a["final_pose_checksum"]


def test_two_ranks_with_real_hip_contexts_on_one_device():
    """The N > 1 control flow with the HIP path in MORE THAN ONE process (the driver's 2/4/8-GPU runs; here both ranks share device 0 of a
    one-GPU box -- RGC_BENCH_DEVICE -- and the barrier / MAX-reduce go over gloo): each rank runs its own sequence (two different
    checksums), each equal to the `--gpus 1` run of that sequence, and `value` is the whole job's 2 K steps over the slower rank's time.
    No scaling claim: the two ranks share one GPU."""
    args = ["--steps", "4", "--warmup", "2", "--n-target", "150000", "--no-cpu-baseline", "--configs", "none", "--no-two-sequences"]
    env2 = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", RGC_BENCH_DEVICE="0", RGC_BENCH_DIST_BACKEND="gloo")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29613", "bench.py", "--gpus", "2"] + args, cwd=ROOT, env=env2, capture_output=True, text=True, timeout=1200)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, lines                                   # rank 0 prints the ONE line
    two = json.loads(lines[0])
    assert two["n_gpus"] == 2 and two["steps"] == 4 and two["scaling"] == "weak"
    cs = two["final_pose_checksum_per_rank"]
    assert len(cs) == 2 and cs[0] != cs[1] and cs[0] == two["final_pose_checksum"]
    assert abs(two["value"] - 2 * 4 / (two["ms_per_step"] * 4 * 1e-3)) <= 1e-2 * two["value"]    # 2 K steps / MAX-over-ranks elapsed
    assert two["one_frame_at_a_time"]["same_poses"]
    for r in (0, 1):
        one = _line([sys.executable, "bench.py", "--gpus", "1", "--sequence", str(r)] + args)
        assert one["final_pose_checksum"] == cs[r], (r, one["final_pose_checksum"], cs)


def test_eight_ranks_on_one_device():
    """BASELINE config 4's launch shape -- `torch.distributed.run --nproc-per-node 8 bench.py --gpus 8` -- at a reduced size with all eight
    ranks on device 0 of a one-GPU box (RGC_BENCH_DEVICE, gloo for the barrier / MAX-reduce): eight different sequences, eight different
    checksums, `value` = 8 K steps over the slowest rank's time, every rank's poses identical one frame at a time.  No scaling claim."""
    args = ["--steps", "3", "--warmup", "1", "--n-target", "60000", "--n-source", "8000", "--no-cpu-baseline", "--configs", "none", "--no-two-sequences"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", RGC_BENCH_DEVICE="0", RGC_BENCH_DIST_BACKEND="gloo")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
                        "--master-port", "29617", "bench.py", "--gpus", "8"] + args, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, lines
    r = json.loads(lines[0])
    assert r["n_gpus"] == 8 and r["steps"] == 3 and r["scaling"] == "weak" and r["config"]["parallelism"] == "sequences x8"
    cs = r["final_pose_checksum_per_rank"]
    assert len(cs) == 8 and len(set(cs)) == 8
    assert abs(r["value"] - 8 * 3 / (r["ms_per_step"] * 3 * 1e-3)) <= 1e-2 * r["value"]
    assert r["one_frame_at_a_time"]["same_poses"]
