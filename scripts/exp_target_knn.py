"""Experiment: time of the map's preprocessing stages alone (setInputTarget x reps), for A/B library builds (RGC_HIP_LIB)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rgc_slam_amd.synth as synth
from rgc_slam_amd import registration
nt = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
world, tgt = synth.make_world_and_map(nt)
v = registration.odometer_vgicp(0)
if os.environ.get("RGC_EXP_RES"): v.setResolution(float(os.environ["RGC_EXP_RES"]))
v.setInputTarget(tgt); v.synchronize()
v.profile_enable(True); v.profile_reset()
for _ in range(reps):
    v.setInputTarget(tgt)
v.synchronize()
p = v.profile()
print(os.environ.get("RGC_HIP_LIB", "default"), {k: round(x["total_ms"] / reps, 4) for k, x in p.items() if x["launches"]})
