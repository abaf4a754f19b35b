"""A cloud whose covariances the caller set, swapped into the source and back into the target with a resident map's commits in between
(found by tests/fuzz/fuzz_api.py: the voxel table of the final target missed voxels)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rgc_slam_amd.synth as synth
from rgc_slam_amd import registration as reg, local_map
world, base = synth.make_world_and_map(20000, seed=5)
base = base.astype(np.float32)
rng = np.random.default_rng(2)
def vox(v):
    x = v.getVoxels(); o = np.lexsort(x["coords"].T[::-1]); return {k: x[k][o] for k in ("coords", "num", "mean", "cov")}
def check(v, cloud, what):
    f = reg.odometer_vgicp(0); f.setInputTarget(cloud); ref = vox(f); x = vox(v)
    same = len(x["coords"]) == len(ref["coords"]) and np.array_equal(x["coords"], ref["coords"]) and float(np.abs(x["cov"] - ref["cov"]).max()) < 1e-12
    print(what, ": voxels", len(x["coords"]), "of", len(ref["coords"]), "ok" if same else "WRONG"); f.close()
v = reg.odometer_vgicp(0); v.setNeighbourReuse(0)
lm = local_map.RollingLocalMap(v); lm.reset(None)
a = np.zeros((4000, 4), np.float32); a[:, :3] = base[rng.choice(len(base), 4000, replace=False)]
lm.insert(a, np.array([0, 0, 0, 1.0]), np.zeros(3))
lm.commit(0.5); T1 = lm.target()[:, :3].copy(); check(v, T1, "after commit 0.5")
S1 = T1[rng.choice(len(T1), len(T1), replace=False)] + np.float32(0.01)
v.setInputSource(S1)
v.setTargetCovariances(v.getTargetCovariances()); check(v, T1, "own covariances set")
v.swapSourceAndTarget(); check(v, S1, "swap 1 (target = the old scan)")
lm.commit(0.3); T2 = lm.target()[:, :3].copy(); check(v, T2, "commit 0.3 re-binds")
v.swapSourceAndTarget(); check(v, T1, "swap 2 (target = the first target, with the caller's covariances)")
