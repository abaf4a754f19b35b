"""getFitnessScore when much of the scan is far from the map: ms per call (rgc_fitness) for a scan that overlaps the map fully, partly,
and from a pose that is metres off."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rgc_slam_amd.synth as synth
from rgc_slam_amd import registration
world, tgt = synth.make_world_and_map(1000000)
poses = synth.make_trajectory(3)
src = synth.make_scan_n(world, poses[1], 30000, seed=synth.SEED + 100)["xyz"]
v = registration.odometer_vgicp(0)
out = {}
far = np.array([[1, 0, 0, -40.0], [0, 1, 0, 25.0], [0, 0, 1, 3.0], [0, 0, 0, 1]], np.float32)
for name, t, T in (("whole map, true pose", tgt, poses[1].astype(np.float32)), ("a quarter of the map, true pose", tgt[(tgt[:, 0] > 0) & (tgt[:, 1] > 0)], poses[1].astype(np.float32)),
                   ("whole map, pose off by 47 m", tgt, far)):
    v.setInputTarget(t); v.setInputSource(src); v.synchronize()
    v.fitnessAt(T)
    t0 = time.perf_counter()
    for _ in range(20):
        f = v.fitnessAt(T)
    out[name] = {"ms_per_call": round((time.perf_counter() - t0) / 20 * 1e3, 3), "fitness": round(f, 4)}
print(json.dumps(out))
