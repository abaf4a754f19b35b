"""Stage times (HIP events) of the c3 configuration (HDL-64 130 k-point scans against a 5 M-point map) or, with the argument c5, of c5
(two fused 64-beam sweeps, 250 k points, against a 20 M-point map), one frame at a time."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rgc_slam_amd.synth as synth
from rgc_slam_amd import registration
world, tile = synth.make_world_and_map(5000000, seed=synth.SEED + 7)
poses = synth.make_trajectory(7, seed=synth.SEED + 7)
e64 = synth.hdl64_elev()
if len(sys.argv) > 1 and sys.argv[1] == "c5":
    L = 2.0 * world.half_extent + 4.0
    tile = np.concatenate([tile, tile + np.float32([L, 0, 0]), tile + np.float32([0, L, 0]), tile + np.float32([L, L, 0])]).astype(np.float32)
    d = 0.5 * float(np.abs(np.diff(np.sort(e64))).min())
    scans = [np.concatenate([synth.make_scan_n(world, poses[i + 1], 125000, elev_deg=e64, seed=synth.SEED + 300 + i)["xyz"],
                             synth.make_scan_n(world, poses[i + 1], 125000, elev_deg=e64 + d, seed=synth.SEED + 400 + i)["xyz"]]).astype(np.float32) for i in range(4)]
else:
    scans = [synth.make_scan_n(world, poses[i + 1], 130000, elev_deg=e64, seed=synth.SEED + 200 + i)["xyz"] for i in range(6)]
v = registration.odometer_vgicp(0)
def to_dev(xyz):
    a = np.zeros((xyz.shape[0], 4), np.float32); a[:, :3] = xyz
    p = v.device_alloc(a.nbytes); v.upload(p, a); return p
d_t = to_dev(tile); d_s = [to_dev(s) for s in scans]
g = poses[0].astype(np.float32)
for rep in range(2):
    if rep == 1:
        v.profile_enable(True); v.profile_reset()
    gg = g
    for i in range(len(scans)):
        v.setInputTargetDevice(d_t, len(tile), 16); v.setInputSourceDevice(d_s[i], len(scans[i]), 16)
        v.align(gg, want_output=False, want_fitness=True); gg = v.getFinalTransformation()
    v.synchronize()
p = v.profile()
print(json.dumps({k: round(x["total_ms"] / len(scans), 4) for k, x in p.items() if x["launches"]}), v.stats())
