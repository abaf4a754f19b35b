"""Small driver for rocprofv3: N full registrations (30k scan vs 1M map), nothing else."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rgc_slam_amd.synth as synth
from rgc_slam_amd import registration
nt = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 3
world, tgt = synth.make_world_and_map(nt)
poses = synth.make_trajectory(frames + 1)
scans = [synth.make_scan_n(world, poses[i + 1], 30000, seed=synth.SEED + 100 + i)["xyz"] for i in range(frames)]
v = registration.odometer_vgicp(0)
g = np.eye(4, dtype=np.float32)
for i in range(frames):
    v.setInputTarget(tgt)
    v.setInputSource(scans[i])
    v.align(g, want_output=False, want_fitness=True)
    g = v.getFinalTransformation()
v.synchronize()
print("done", v.stats())
