"""Experiment: the scan's preprocessing alone (setInputSource x reps): grid build, bulk kNN, cooperative kNN."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rgc_slam_amd.synth as synth
from rgc_slam_amd import registration
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
world, tgt = synth.make_world_and_map(200000)
scan = synth.make_scan_n(world, np.eye(4), 30000)["xyz"]
v = registration.odometer_vgicp(0)
v.setInputTarget(tgt); v.setInputSource(scan); v.synchronize()
v.profile_enable(True); v.profile_reset()
for _ in range(reps):
    v.setInputSource(scan)
v.synchronize()
p = v.profile()
print(os.environ.get("RGC_HIP_LIB", "default"), os.environ.get("RGC_KNN_HEAVY", ""), {k: round(x["total_ms"] / reps, 4) for k, x in p.items() if x["launches"]}, "deferred", v.stats()["deferred_source"])
