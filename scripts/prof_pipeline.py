"""Driver for rocprofv3: the two-context pipeline (PipelinedVGICP) over device-resident clouds, like bench.py's timed loop."""
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
pv = registration.PipelinedVGICP(0, depth=int(sys.argv[3]) if len(sys.argv) > 3 else 2)
v = pv.v[0]
def to_dev(xyz):
    a = np.zeros((xyz.shape[0], 4), np.float32); a[:, :3] = xyz
    p = v.device_alloc(a.nbytes); v.upload(p, a); return p
d_tgt = to_dev(tgt); d_s = [to_dev(s) for s in scans]
def setc(j, w):
    w.setInputTargetDevice(d_tgt, len(tgt), 16)
    w.setInputSourceDevice(d_s[j % frames], 30000, 16)
pv.run(3 * frames, setc, poses[0].astype(np.float32), want_fitness=True)
pv.synchronize()
print("done")
