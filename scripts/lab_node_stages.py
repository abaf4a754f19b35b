"""Developer aid: stage times (the library's profiling regions) of the registration inside the odometer's frame body on the config-2
stand-in sweeps, and how far each frame's scan points are from its map (what the fitness score's exact 1-NN has to search)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from scipy.spatial import cKDTree
import rgc_slam_amd.synth as synth
from rgc_slam_amd import odometry
world = synth.make_world(half_extent=45.0, seed=synth.SEED)
poses = synth.make_trajectory(13, seed=synth.SEED)
raws = []
for k in range(12):
    sc = synth.make_scan(world, poses[k], n_az=1800, seed=synth.SEED + 50 + k, T_ws_end=poses[k + 1])
    raws.append(np.concatenate([sc["xyz"], sc["intensity"][:, None]], axis=1).astype(np.float32))
hb = odometry.HipBackend(0)
od = odometry.Odometer(hb)
seen = []
reg0 = hb.register
def register(source, target, guess):
    T, f = reg0(source, target, guess)
    moved = source[:, :3].astype(np.float64) @ T[:3, :3].T.astype(np.float64) + T[:3, 3]
    d, _ = cKDTree(target[:, :3]).query(moved)
    seen.append({"n_source": int(len(source)), "n_target": int(len(target)), "nn_p99_m": round(float(np.percentile(d, 99)), 2), "nn_max_m": round(float(d.max()), 2),
                 "farther_than_1_2_3_5_m": [int((d > x).sum()) for x in (1, 2, 3, 5)], "fitness": float(f)})
    return T, f
hb.register = register
for r in raws[:4]:
    od.process(r)
hb.reg.profile_enable(True); hb.reg.profile_reset()
n0 = len(seen)
for r in raws[4:]:
    od.process(r)
hb.reg.synchronize()
p = hb.reg.profile()
frames = len(seen) - n0
print(json.dumps({"stage_ms_per_frame": {k: round(x["total_ms"] / frames, 4) for k, x in p.items() if x["launches"]}, "frames": seen[n0:]}))
hb.close()
