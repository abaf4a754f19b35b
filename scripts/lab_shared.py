"""tests/test_gpu_parity.py::test_shared_target's sequence: one context against two contexts sharing the target, frame by frame."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rgc_slam_amd.synth as synth
from rgc_slam_amd import registration as reg
world, tgt = synth.make_world_and_map(100000, seed=synth.SEED)
T_true = synth.se3(synth.rot_zyx(0.02, 0.003, -0.002), [0.15, 0.01, 0.002])
src = synth.make_scan_n(world, T_true, 30000, seed=synth.SEED)["xyz"]
a, b = reg.odometer_vgicp(0), reg.odometer_vgicp(0)
a.setInputTarget(tgt); b.shareTargetFrom(a)
g = np.eye(4, dtype=np.float32)
for v in (a, b):
    v.setInputSource(src); v.align(g, want_output=False, want_fitness=True)
poses = synth.make_trajectory(6, seed=synth.SEED + 3)
scans = [synth.make_scan_n(world, poses[i + 1], 15000, seed=synth.SEED + 500 + i)["xyz"] for i in range(5)]
seq, covs, gg = [], [], poses[0].astype(np.float32)
for s in scans:
    a.setInputSource(s); covs.append(a.getSourceCovariances().copy()); print("a stats", {k: a.stats()[k] for k in ("source_cells", "deferred_source")})
    a.align(gg, want_output=False); gg = a.getFinalTransformation(); seq.append(gg)
pv = reg.PipelinedVGICP(0, depth=2, contexts=[a, b]); pv.share_target()
cov2 = {}
def setc(i, w):
    w.setInputSource(scans[i])
out = pv.run(len(scans), setc, poses[0].astype(np.float32))
for i, (x, y) in enumerate(zip(out, seq)):
    print(i, "pose equal", np.array_equal(x, y), float(np.abs(x - y).max()))
# the same scans' covariances on b (other history)
for i, s in enumerate(scans):
    b.setInputSource(s); cb = b.getSourceCovariances()
    print(i, "cov differ a-vs-b:", int(np.any((cb != covs[i]).reshape(len(s), -1), axis=1).sum()), {k: b.stats()[k] for k in ("source_cells", "deferred_source")})
