"""Time of the kNN / covariance stage on the map and the scan of the c-main workload, per profiling region, and the deferred counts.
    python scripts/lab_knn.py [n_target] [reps]        (RGC_SRC_RES fixes the scan's cell size)"""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rgc_slam_amd.synth as synth
from rgc_slam_amd import registration
nt = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
world, tgt = synth.make_world_and_map(nt, seed=synth.SEED)
poses = synth.make_trajectory(4, seed=synth.SEED)
src = synth.make_scan_n(world, poses[1], 30000, seed=synth.SEED + 100)["xyz"]
v = registration.odometer_vgicp(0)
out = {}
for name, cloud, setter in (("target", tgt, v.setInputTarget), ("source", src, v.setInputSource)):
    setter(cloud); v.synchronize()
    v.profile_enable(True); v.profile_reset()
    t0 = time.perf_counter()
    for _ in range(reps):
        setter(cloud)
    v.synchronize()
    wall = (time.perf_counter() - t0) / reps * 1e3
    p = v.profile()
    out[name] = {k: round(x["total_ms"] / reps, 4) for k, x in p.items() if x["launches"]}
    out[name]["wall_ms"] = round(wall, 4)
    v.profile_enable(False)
    out[name]["deferred"] = v.stats()["deferred_target" if name == "target" else "deferred_source"]
print(json.dumps(out), flush=True)
v.close()
