"""Developer aid: per-frame wall time of the C++ node in each mode (the same sweeps as scripts/bench_cpp_node.py), to see whether a mode's
mean is a steady figure or a few slow frames."""
import sys, os, subprocess, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import rgc_slam_amd.synth as synth
world = synth.make_world(half_extent=45.0, seed=synth.SEED)
poses = synth.make_trajectory(25, seed=synth.SEED)
tmp = tempfile.mkdtemp()
path = os.path.join(tmp, "sweeps.bin")
dt = np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("intensity", "<f4"), ("ring", "<u2"), ("time", "<f4")])
with open(path, "wb") as f:
    f.write(np.int32(24).tobytes())
    for k in range(24):
        sc = synth.make_scan(world, poses[k], n_az=1800, seed=synth.SEED + 50 + k, T_ws_end=poses[k + 1])
        rec = np.zeros(len(sc["xyz"]), dt)
        rec["x"], rec["y"], rec["z"], rec["intensity"] = sc["xyz"][:, 0], sc["xyz"][:, 1], sc["xyz"][:, 2], sc["intensity"]
        f.write(np.int32(len(rec)).tobytes()); f.write(rec.tobytes())
exe = os.path.join(tmp, "node")
subprocess.check_call(["g++", "-std=c++14", "-O2", "-pthread", os.path.join(ROOT, "tests", "cpp", "test_odometry_node.cpp"), "-o", exe,
                       "-L", os.path.join(ROOT, "rgc-slam_amd"), "-lrgc_hip", "-Wl,-rpath," + os.path.join(ROOT, "rgc-slam_amd")])
for name, resident, chain in (("reference_semantics", 0, 0), ("reference_semantics_device_chain", 0, 1), ("resident_map", 1, 0), ("resident_map_device_chain", 1, 1)):
    for rep in range(2):
        r = subprocess.run([exe, path, str(resident), "1", "50", str(chain), "0"], capture_output=True, text=True, timeout=600, env=dict(os.environ, **({"RGC_TRACE_ALLOC": "1"} if rep else {})))
        out = r.stdout
        ms = [float(l.split()[-1]) for l in out.splitlines() if l.startswith("pose")]
        print(name, "rep", rep, "mean of frames 4..", round(float(np.mean(ms[4:])), 3), "median", round(float(np.median(ms[4:])), 3), [round(x, 2) for x in ms])
        if rep:
            print("   allocations traced:", r.stderr.strip().splitlines()[-30:])
