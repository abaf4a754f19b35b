#!/bin/bash
# Per-sweep wall times of rgc::OdometryNode in its four unpipelined modes (tests/cpp/test_odometry_node.cpp on scripts/bench_cpp_node.py's sweeps):
# which frames are slow.   bash scripts/node_frames.sh [repetitions]
cd "$GRAFT_REPO_ROOT"
python3 - <<'PY'
import os, sys, subprocess, numpy as np
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import rgc_slam_amd.synth as synth
world = synth.make_world(half_extent=45.0, seed=synth.SEED)
poses = synth.make_trajectory(25, seed=synth.SEED)
dt = np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("intensity", "<f4"), ("ring", "<u2"), ("time", "<f4")])
with open("/tmp/sweeps.bin", "wb") as f:
    f.write(np.int32(24).tobytes())
    for k in range(24):
        sc = synth.make_scan(world, poses[k], n_az=1800, seed=synth.SEED + 50 + k, T_ws_end=poses[k + 1])
        rec = np.zeros(len(sc["xyz"]), dt)
        rec["x"], rec["y"], rec["z"], rec["intensity"] = sc["xyz"][:, 0], sc["xyz"][:, 1], sc["xyz"][:, 2], sc["intensity"]
        f.write(np.int32(len(rec)).tobytes()); f.write(rec.tobytes())
subprocess.check_call(["g++", "-std=c++14", "-O2", "-pthread", "tests/cpp/test_odometry_node.cpp", "-o", "/tmp/node", "-L", "rgc-slam_amd", "-lrgc_hip", "-Wl,-rpath," + os.path.join(os.getcwd(), "rgc-slam_amd")])
PY
for rep in $(seq 1 ${1:-2}); do
for mode in "0 0" "0 1" "1 0" "1 1"; do
  set -- $mode
  echo "resident=$1 chain=$2: $(RGC_TRACE_ALLOC=1 /tmp/node /tmp/sweeps.bin $1 1 50 $2 0 2>/tmp/node.err | grep '^pose' | awk '{printf "%s ", $NF}')"
  grep -c "grew" /tmp/node.err | xargs echo "   buffer growths:"; grep "grew\|left its" /tmp/node.err | tail -4
done
done
