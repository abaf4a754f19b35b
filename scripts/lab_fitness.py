"""Developer aid: the fitness score's stage time (HIP-event region) for a map / scan size.   python scripts/lab_fitness.py [n_map] [n_scan] [64]"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rgc_slam_amd.synth as synth
from rgc_slam_amd import registration
nm = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
ns = int(sys.argv[2]) if len(sys.argv) > 2 else 30000
world, tgt = synth.make_world_and_map(nm, seed=synth.SEED + (7 if nm > 1000000 else 0))
poses = synth.make_trajectory(4, seed=synth.SEED)
kw = dict(elev_deg=synth.hdl64_elev()) if len(sys.argv) > 3 else {}
src = synth.make_scan_n(world, poses[1], ns, seed=synth.SEED + 100, **kw)["xyz"]
v = registration.odometer_vgicp(0)
v.setInputTarget(tgt)
g = np.linalg.inv(poses[0]) @ poses[1]
for rep in range(2):
    v.setInputSource(src); v.align(g.astype(np.float32))
v.profile_enable(True); v.profile_select(["fitness", "linearize"]); v.profile_reset()
for rep in range(6):
    v.setInputSource(src); v.align(g.astype(np.float32)); f = v.getFitnessScore()
p = v.profile()
print(json.dumps({"n_map": nm, "n_scan": ns, "ms": {k: round(x["total_ms"] / max(x["launches"], 1), 4) for k, x in p.items() if x["launches"]}, "fitness": f,
                  "iterations": v.nr_iterations, "voxels": v.stats()["n_voxels"]}))
v.close()
