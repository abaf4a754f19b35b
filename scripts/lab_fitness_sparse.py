"""Developer aid: the fitness score (exact 1-NN of every scan point) on the odometer's own kind of map -- three 0.3 m-filtered keyframes of
a 16-beam sensor -- and how far the scan's points are from it (where the growing-cube search spends its time)."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from scipy.spatial import cKDTree
import rgc_slam_amd.synth as synth
from rgc_slam_amd import registration
from oracle import oracle as orc
world = synth.make_world(half_extent=45.0, seed=synth.SEED)
poses = synth.make_trajectory(6, seed=synth.SEED)
def sweep(k):
    sc = synth.make_scan(world, poses[k], n_az=1800, seed=synth.SEED + 50 + k)
    return np.concatenate([sc["xyz"], sc["intensity"][:, None]], axis=1).astype(np.float32)
def to_frame(xyzi, T_from, T_to):
    M = np.linalg.inv(T_to) @ T_from
    out = xyzi.copy()
    out[:, :3] = (xyzi[:, :3].astype(np.float64) @ M[:3, :3].T + M[:3, 3]).astype(np.float32)
    return out
kf = np.concatenate([to_frame(sweep(k), poses[k], poses[3]) for k in range(3)])
tgt = orc.voxelgrid_filter(kf, 0.3)[:, :3].copy()
src = orc.voxelgrid_filter(sweep(4), 0.2)[:, :3].copy()
T = (np.linalg.inv(poses[3]) @ poses[4]).astype(np.float32)
moved = src.astype(np.float64) @ T[:3, :3].T.astype(np.float64) + T[:3, 3]
d, _ = cKDTree(tgt).query(moved)
out = {"n_map": int(len(tgt)), "n_scan": int(len(src)), "nn_distance_m": {"median": float(np.median(d)), "p99": float(np.percentile(d, 99)), "max": float(d.max())},
       "points_farther_than_m": {str(x): int((d > x).sum()) for x in (0.5, 1, 2, 3, 5, 8)}}
v = registration.odometer_vgicp(0)
v.setInputTarget(tgt); v.setInputSource(src)
v.align(T, want_output=False, want_fitness=True)
for name, S in (("all points", src), ("points within 1 m of the map", src[d <= 1.0]), ("points within 2 m", src[d <= 2.0])):
    v.setInputSource(S); v.synchronize()
    f = v.fitnessAt(T)
    t0 = time.perf_counter()
    for _ in range(20):
        f = v.fitnessAt(T)
    out[name] = {"n": int(len(S)), "ms_per_call_incl_round_trip": round((time.perf_counter() - t0) / 20 * 1e3, 4), "fitness": float(f)}
print(json.dumps(out))
v.close()
