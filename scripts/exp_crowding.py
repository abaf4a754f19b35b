"""Crowding (sum of count^2 / n: the mean population of a point's own cell) of the c3 and c5 scans at several cell sizes."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rgc_slam_amd.synth as synth
world, tile = synth.make_world_and_map(1000000, seed=synth.SEED + 7)
poses = synth.make_trajectory(3, seed=synth.SEED + 7)
e64 = synth.hdl64_elev()
a = synth.make_scan_n(world, poses[1], 130000, elev_deg=e64, seed=synth.SEED + 200)["xyz"]
b1 = synth.make_scan_n(world, poses[1], 125000, elev_deg=e64, seed=synth.SEED + 300)["xyz"]
b2 = synth.make_scan_n(world, poses[1], 125000, elev_deg=e64 + 0.5 * float(np.abs(np.diff(np.sort(e64))).min()), seed=synth.SEED + 400)["xyz"]
v16 = synth.make_scan_n(world, poses[1], 30000, seed=synth.SEED + 100)["xyz"]
out = {}
for name, s in (("vlp16_30k", v16), ("c3_hdl64_130k", a), ("c5_2x64_250k", np.concatenate([b1, b2]))):
    for res in (1.0, 0.5, 0.25):
        c = np.floor(s / res - 0.5).astype(np.int64)
        key = (c[:, 2] * 100000 + c[:, 1]) * 100000 + c[:, 0]
        _, cnt = np.unique(key, return_counts=True)
        out[f"{name}@{res}"] = round(float((cnt.astype(float) ** 2).sum() / len(s)), 1)
print(json.dumps(out))
