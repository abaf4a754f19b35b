#!/bin/bash
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
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pn && rocprofv3 --hip-runtime-trace -d /tmp/pn -o run --output-format csv -- /tmp/node /tmp/sweeps.bin 1 1 50 0 0 > /tmp/node.out 2>&1
grep "^pose" /tmp/node.out | awk '{print $NF}' | tr '\n' ' '; echo
python3 - <<'PY'
import csv, glob
f = glob.glob("/tmp/pn/**/*hip_api_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
big = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), (int(r["Start_Timestamp"]) - t0) / 1e6, r["Function"]) for r in rows]
big.sort(reverse=True)
for d, t, fn in big[:8]:
    print("%9.3f ms at %9.3f ms  %s" % (d / 1e6, t, fn))
# the calls around the slow copies that are not part of the start-up
slow = [i for i, r in enumerate(rows) if r["Function"] == "hipMemcpyAsync" and int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) > 2000000]
for i in slow[1:3]:
    print("---- around call", i)
    for r in rows[max(0, i - 14):i + 4]:
        print("   %9.3f ms  +%8.3f ms  %s" % ((int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, r["Function"]))
PY
