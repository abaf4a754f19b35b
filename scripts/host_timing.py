"""Host-side wall time of each API call in the bench loop (where does the host wait?)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import rgc_slam_amd.synth as synth
from rgc_slam_amd import registration
frames = 30
world, tgt = synth.make_world_and_map(1000000)
poses = synth.make_trajectory(frames + 1)
scans = [synth.make_scan_n(world, poses[i + 1], 30000, seed=synth.SEED + 100 + i)["xyz"] for i in range(frames)]
v = registration.odometer_vgicp(0)
def to_dev(xyz):
    a = np.zeros((xyz.shape[0], 4), np.float32); a[:, :3] = xyz
    p = v.device_alloc(a.nbytes); v.upload(p, a); return p
d_t = to_dev(tgt); d_s = [to_dev(s) for s in scans]
g = poses[0].astype(np.float32)
T = np.zeros((frames, 5))
for i in range(frames):
    t0 = time.perf_counter(); v.setInputTargetDevice(d_t, tgt.shape[0], 16)
    t1 = time.perf_counter(); v.setInputSourceDevice(d_s[i], scans[i].shape[0], 16)
    t2 = time.perf_counter(); v.align(g, want_output=False, want_fitness=True)
    t3 = time.perf_counter(); g = v.getFinalTransformation()
    t4 = time.perf_counter(); st = v.stats()
    t5 = time.perf_counter()
    T[i] = [t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4]
T = T[5:] * 1e6
print("us per call  set_target %.1f  set_source %.1f  align %.1f  getFinal %.1f  stats %.1f  total %.1f" % (*T.mean(0), T.sum(1).mean()))
