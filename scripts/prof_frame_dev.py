"""Driver for rocprofv3: N registrations (30 k scan vs 1 M map) with the clouds resident on the device, like bench.py's timed loop."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rgc_slam_amd.synth as synth
from rgc_slam_amd import registration
nt = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 8
world, tgt = synth.make_world_and_map(nt)
poses = synth.make_trajectory(frames + 1)
scans = [synth.make_scan_n(world, poses[i + 1], 30000, seed=synth.SEED + 100 + i)["xyz"] for i in range(frames)]
v = registration.odometer_vgicp(0)
def to_dev(xyz):
    a = np.zeros((xyz.shape[0], 4), np.float32); a[:, :3] = xyz
    p = v.device_alloc(a.nbytes); v.upload(p, a); return p
d_tgt = to_dev(tgt); d_s = [to_dev(s) for s in scans]
g = np.eye(4, dtype=np.float32)
for rep in range(3):
    for i in range(frames):
        v.setInputTargetDevice(d_tgt, len(tgt), 16)
        v.setInputSourceDevice(d_s[i], 30000, 16)
        v.align(g, want_output=False, want_fitness=True)
        g = v.getFinalTransformation()
v.synchronize()
print("done", v.stats())
