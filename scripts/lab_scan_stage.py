"""Developer aid: the scan's preparation stages by themselves (HIP-event regions), RGC_COOP_STREAM=0/1 in the environment."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rgc_slam_amd.synth as synth
from rgc_slam_amd import registration
world, tgt = synth.make_world_and_map(100000, seed=synth.SEED)
poses = synth.make_trajectory(4, seed=synth.SEED)
src = synth.make_scan_n(world, poses[1], 30000, seed=synth.SEED + 100)["xyz"]
v = registration.odometer_vgicp(0)
v.setInputTarget(tgt)
for rep in range(3):
    v.setInputSource(src); v.align(np.eye(4, dtype=np.float32))
v.profile_enable(True); v.profile_select(["grid_build", "knn_cov_source"] if os.environ.get("RGC_COOP_STREAM", "1") != "0" else ["grid_build", "knn_cov_source", "knn_coop_source"]); v.profile_reset()
import time
for rep in range(10):
    v.setInputSource(src); v.synchronize()
p = v.profile()
print(json.dumps({k: round(x["total_ms"] / max(x["launches"], 1), 4) for k, x in p.items() if x["launches"]}), "deferred", v.stats()["deferred_source"])
v.profile_enable(False)
t0 = time.perf_counter()
for rep in range(50):
    v.setInputSource(src); v.synchronize()
print("ms per setInputSource", round((time.perf_counter() - t0) / 50 * 1e3, 4))
v.close()
