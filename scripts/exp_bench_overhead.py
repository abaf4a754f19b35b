"""Where does the one-time ~40 ms stall of a fresh process come from?  Per-call wall times of 150 identical frames; prints the
outliers with the time since the first GPU call."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rgc_slam_amd.synth as synth
from rgc_slam_amd import registration
world, tgt = synth.make_world_and_map(1000000, seed=synth.SEED)
poses = synth.make_trajectory(24, seed=synth.SEED)
scans = [synth.make_scan_n(world, poses[i + 1], 30000, seed=synth.SEED + 100 + i)["xyz"] for i in range(8)]
T0 = time.perf_counter()
v = registration.odometer_vgicp(0)
def to_dev(xyz):
    a = np.zeros((xyz.shape[0], 4), np.float32); a[:, :3] = xyz
    p = v.device_alloc(a.nbytes); v.upload(p, a); return p
d_tgt = to_dev(tgt); d_s = [to_dev(s) for s in scans]
print("setup done at", round(time.perf_counter() - T0, 4))
g = poses[0].astype(np.float32)
for i in range(int(os.environ.get("FRAMES", "150"))):
    t = [time.perf_counter()]
    v.setInputTargetDevice(d_tgt, len(tgt), 16); t.append(time.perf_counter())
    v.setInputSourceDevice(d_s[i % 8], 30000, 16); t.append(time.perf_counter())
    v.align(g, want_output=False, want_fitness=True); t.append(time.perf_counter())
    d = [round(1e3 * (b - a), 3) for a, b in zip(t, t[1:])]
    if sum(d) > 2.0 or i < 3:
        print("frame", i, "at", round(t[0] - T0, 4), "s: set_target, set_source, align ms =", d)
print("end at", round(time.perf_counter() - T0, 4))
