"""Does a scan's covariance depend on what its context prepared before (the speculative grid, the crowding-driven cell size)?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rgc_slam_amd.synth as synth
from rgc_slam_amd import registration
world, tgt = synth.make_world_and_map(200000, seed=synth.SEED)
poses = synth.make_trajectory(8, seed=synth.SEED + 3)
scans = [synth.make_scan_n(world, poses[i + 1], 15000, seed=synth.SEED + 500 + i)["xyz"] for i in range(6)]
a, b = registration.odometer_vgicp(0), registration.odometer_vgicp(0)
for s in scans[:3]:
    a.setInputSource(s)
b.setInputSource(scans[1]); b.setInputSource(scans[3])
for name, s in (("scan4", scans[4]), ("scan5", scans[5])):
    a.setInputSource(s); b.setInputSource(s)
    ca, cb = a.getSourceCovariances().reshape(len(s), -1), b.getSourceCovariances().reshape(len(s), -1)
    d = np.nonzero(np.any(ca != cb, axis=1))[0]
    sa, sb = a.stats(), b.stats()
    print(name, "differ:", len(d), "max", float(np.abs(ca - cb).max()), {k: (sa[k], sb[k]) for k in ("source_cells", "deferred_source", "source_crowding")})
    c = registration.odometer_vgicp(0); c.setInputSource(s)
    cc = c.getSourceCovariances().reshape(len(s), -1)
    print("   fresh context vs a:", int(np.any(ca != cc, axis=1).sum()), " vs b:", int(np.any(cb != cc, axis=1).sum()), c.stats()["source_cells"], c.stats()["deferred_source"])
