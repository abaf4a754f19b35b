#!/bin/bash
# kernel timeline of one frame of the C++ node (device chain): builds the driver, writes 12 sweeps, profiles
set -e
cd "$GRAFT_REPO_ROOT"
python - <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
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
PY
g++ -std=c++14 -O2 -pthread tests/cpp/test_odometry_node.cpp -o /tmp/node -L rgc-slam_amd -lrgc_hip -Wl,-rpath,$GRAFT_REPO_ROOT/rgc-slam_amd
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/tl_node -o runc --output-format csv -- /tmp/node /tmp/sweeps.bin 1 1 50 1 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT && python scripts/timeline_any.py gpurun_out/tl_node k_pc2_unpack
