"""Small driver for rocprofv3: N dependent steps (bench.DependentSequence, one context), nothing else.  python scripts/prof_dependent.py [frames] [overlap 0|1]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
import rgc_slam_amd.synth as synth
from rgc_slam_amd import registration
K = int(sys.argv[1]) if len(sys.argv) > 1 else 12
overlap = len(sys.argv) > 2 and sys.argv[2] == "1"
world, tgt = synth.make_world_and_map(1000000, seed=synth.SEED)
poses = synth.make_trajectory(K + 2, seed=synth.SEED)
scans = [synth.make_scan_n(world, poses[i + 1], 30000, seed=synth.SEED + 100 + i)["xyz"] for i in range(K + 1)]
pv = registration.PipelinedVGICP(0, depth=2)
v = pv.v[0]
def to_dev(xyz):
    a = np.zeros((xyz.shape[0], 4), np.float32); a[:, :3] = xyz
    p = v.device_alloc(a.nbytes); v.upload(p, a); return p
d_map, d_scans = to_dev(tgt), [to_dev(s) for s in scans]
seq = bench.DependentSequence(pv.v, d_map, len(tgt), d_scans, [len(s) for s in scans])
I4 = np.eye(4, dtype=np.float32)
for w in pv.v:
    seq.v = [w]; seq.run(0, 1, poses[0], I4, False)
seq.v = pv.v
seq.run(1, K, poses[0], I4, overlap)
seq.run(1, K, poses[0], I4, overlap)
pv.synchronize()
print("done")
