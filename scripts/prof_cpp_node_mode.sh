#!/bin/bash
# Developer aid: kernel timeline of the C++ node in one mode.  usage (GPU box): prof_cpp_node_mode.sh <resident 0|1> <chain 0|1> <tag>
cd "$GRAFT_REPO_ROOT"
python3 - <<'PY'
import os, sys, subprocess, tempfile, numpy as np
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import rgc_slam_amd.synth as synth
world = synth.make_world(half_extent=45.0, seed=synth.SEED)
poses = synth.make_trajectory(13, seed=synth.SEED)
dt = np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("intensity", "<f4"), ("ring", "<u2"), ("time", "<f4")])
with open("/tmp/sweeps.bin", "wb") as f:
    f.write(np.int32(12).tobytes())
    for k in range(12):
        sc = synth.make_scan(world, poses[k], n_az=1800, seed=synth.SEED + 50 + k, T_ws_end=poses[k + 1])
        rec = np.zeros(len(sc["xyz"]), dt)
        rec["x"], rec["y"], rec["z"], rec["intensity"] = sc["xyz"][:, 0], sc["xyz"][:, 1], sc["xyz"][:, 2], sc["intensity"]
        f.write(np.int32(len(rec)).tobytes()); f.write(rec.tobytes())
subprocess.check_call(["g++", "-std=c++14", "-O2", "-pthread", "tests/cpp/test_odometry_node.cpp", "-o", "/tmp/node", "-L", "rgc-slam_amd", "-lrgc_hip", "-Wl,-rpath," + os.path.join(os.getcwd(), "rgc-slam_amd")])
PY
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pn && rocprofv3 --kernel-trace --memory-copy-trace -d /tmp/pn -o run --output-format csv -- /tmp/node /tmp/sweeps.bin $1 1 50 $2 0 > /tmp/node.out 2>&1
tail -2 /tmp/node.out
cd "$GRAFT_REPO_ROOT" && python3 scripts/timeline_window.py /tmp/pn 1600 0 > gpurun_out/timeline_node_$3.txt; tail -60 gpurun_out/timeline_node_$3.txt
