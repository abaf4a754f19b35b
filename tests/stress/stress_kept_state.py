"""Randomised soak of the paths that keep state from one call to the next (run by hand on the GPU box: python tests/stress/stress_kept_state.py [iterations]):
the leaf filter on the box kept from the previous cloud (against the CPU oracle, bit for bit) and the front-end sized from the previous
sweep (against the synchronous path on a second context, bit for bit, and its decisions against the CPU oracle)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import rgc_slam_amd.synth as synth
from rgc_slam_amd import odometry, frontend
from oracle import oracle
oracle.build()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(11)
world = synth.make_world(half_extent=45.0, seed=synth.SEED)
poses = synth.make_trajectory(8, seed=synth.SEED)
base = [synth.make_scan(world, poses[i], n_az=int(rng.integers(300, 1900)), seed=synth.SEED + 900 + i) for i in range(6)]
p = odometry.Preprocessor(0)
bad = 0
for it in range(N):
    sc = base[int(rng.integers(0, len(base)))]
    xyzi = np.concatenate([sc["xyz"], (sc["ring"] + 0.1 * sc["rel_time"])[:, None].astype(np.float32)], axis=1)
    kind = int(rng.integers(0, 6))
    if kind == 1: xyzi = xyzi[: int(rng.integers(1, len(xyzi)))]
    if kind == 2: xyzi = xyzi + np.float32([rng.uniform(-12, 12), rng.uniform(-12, 12), rng.uniform(-3, 3), 0])
    if kind == 3: xyzi = xyzi * np.float32([0.05, 0.05, 0.05, 1])                      # dense: the other path
    if kind == 4: xyzi = np.concatenate([xyzi, xyzi + np.float32([0.013, 0, 0, 0])])   # crowded rows
    leaf = float(rng.choice([0.2, 0.3, 0.2, 0.5]))
    got, exp = p.voxelGridFilter(xyzi, leaf), oracle.voxelgrid_filter(xyzi, leaf)
    if got.shape != exp.shape or not np.array_equal(got, exp):
        bad += 1
        print("leaf filter differs: iteration", it, "kind", kind, "leaf", leaf, got.shape, exp.shape)
p.close()
print("leaf filter:", N, "calls,", bad, "differ")
a, b = frontend.ScanRegistration(16), frontend.ScanRegistration(16)
bad2 = 0
keys = ("curvature", "curvature2", "inten_curvature", "ground_marked", "picked", "label", "inten_label", "sharp", "flat", "inten", "groundparam", "ring_count")
for it in range(N):
    T = poses[int(rng.integers(0, 8))]
    sc = synth.make_scan(world, T, n_az=int(rng.choice([200, 500, 900, 1800, 1800, 1800, 2400])), seed=synth.SEED + 2000 + it)
    raw = np.concatenate([sc["xyz"], sc["intensity"][:, None]], axis=1).astype(np.float32)
    if rng.random() < 0.1: raw = raw[: int(rng.integers(0, 400))]
    if len(raw) == 0: continue
    x, y = a.laserCloudHandler(raw, cloud=False), b.laserCloudHandler(raw)
    if x["n_cloud"] != y["n_cloud"] or any(not np.array_equal(x[k], y[k]) for k in keys):
        bad2 += 1
        print("front-end differs: iteration", it, len(raw))
    o = oracle.frontend(raw, n_scans=16)   # and the selection (six sectors at once, ordered repeats) against the CPU oracle's serial walk
    if x["n_cloud"] != o["n_cloud"] or any(not np.array_equal(x[k], o[k]) for k in ("picked", "label", "inten_label", "ground_marked")) or \
            any(x[k].shape != o[k].shape or not np.array_equal(x[k][:, :3], o[k][:, :3]) for k in ("sharp", "flat", "inten")):
        bad2 += 1
        print("front-end differs from the oracle: iteration", it, len(raw))
a.close(); b.close()
print("front-end:", N, "sweeps,", bad2, "differ")
sys.exit(1 if bad or bad2 else 0)
