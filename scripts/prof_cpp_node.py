"""Developer aid: kernel / copy timeline of ONE steady-state frame of the C++ odometry node (resident map, device chain) under rocprofv3.
Run on the GPU box from the repo root: python scripts/prof_cpp_node.py [out_dir]"""
import sys, os, subprocess, csv, glob
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import rgc_slam_amd.synth as synth
out = os.path.abspath(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "nodeprof"))
os.makedirs(out, exist_ok=True)
world = synth.make_world(half_extent=45.0, seed=synth.SEED)
poses = synth.make_trajectory(25, seed=synth.SEED)
path = os.path.join(out, "sweeps.bin")
dt = np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("intensity", "<f4"), ("ring", "<u2"), ("time", "<f4")])
with open(path, "wb") as f:
    f.write(np.int32(24).tobytes())
    for k in range(24):
        sc = synth.make_scan(world, poses[k], n_az=1800, seed=synth.SEED + 50 + k, T_ws_end=poses[k + 1])
        rec = np.zeros(len(sc["xyz"]), dt)
        rec["x"], rec["y"], rec["z"], rec["intensity"] = sc["xyz"][:, 0], sc["xyz"][:, 1], sc["xyz"][:, 2], sc["intensity"]
        f.write(np.int32(len(rec)).tobytes()); f.write(rec.tobytes())
exe = os.path.join(out, "node")
subprocess.check_call(["g++", "-std=c++14", "-O2", "-pthread", os.path.join(ROOT, "tests", "cpp", "test_odometry_node.cpp"), "-o", exe,
                       "-L", os.path.join(ROOT, "rgc-slam_amd"), "-lrgc_hip", "-Wl,-rpath," + os.path.join(ROOT, "rgc-slam_amd")])
env = dict(os.environ, TMPDIR="/tmp")
pipe = sys.argv[2] if len(sys.argv) > 2 else "0"
subprocess.run(["rocprofv3", "--kernel-trace", "--memory-copy-trace", "--stats", "-d", out, "-o", "run", "--output-format", "csv", "--", exe, path, "1", "1", "50", "1", pipe],
               cwd="/tmp", env=env, stdout=open(os.path.join(out, "log.txt"), "w"), stderr=subprocess.STDOUT)
print(open(os.path.join(out, "log.txt")).read().strip().splitlines()[-1])
ev = []
for r in csv.DictReader(open(glob.glob(out + "/**/run_kernel_trace.csv", recursive=True)[0])):
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("rgck::", "").replace("void ", "").split("(")[0][:40], r.get("Stream_Id", "")))
for r in csv.DictReader(open(glob.glob(out + "/**/run_memory_copy_trace.csv", recursive=True)[0])):
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r["Direction"], ""))
ev.sort()
# frames: the front-end's first kernel opens one
starts = [i for i, e in enumerate(ev) if e[2].startswith("k_fe_") and (i == 0 or not any(x[2].startswith("k_fe_") for x in ev[max(0, i - 6):i]))]
print("frames seen", len(starts))
a, b = starts[15], starts[16]
t0 = ev[a][0]
print("frame 15: %.1f us from its first kernel to the next frame's first kernel; busy %.1f us" % ((ev[b][0] - t0) / 1e3, sum(e[1] - e[0] for e in ev[a:b]) / 1e3))
prev_end = t0
for e in ev[a:b]:
    print("%8.1f  +%6.1f gap %6.1f  %s %s" % ((e[0] - t0) / 1e3, (e[1] - e[0]) / 1e3, (e[0] - prev_end) / 1e3, e[2], e[3]))
    prev_end = max(prev_end, e[1])
