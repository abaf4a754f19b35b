"""Small driver for rocprofv3: N dependent steps (bench.DependentSequence, one context), nothing else.  python scripts/prof_dependent.py [frames] [overlap 0|1] [cmain|c3|c5] [lazy margin]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
import rgc_slam_amd.synth as synth
from rgc_slam_amd import registration
K = int(sys.argv[1]) if len(sys.argv) > 1 else 12
overlap = len(sys.argv) > 2 and sys.argv[2] == "1"
cfg = sys.argv[3] if len(sys.argv) > 3 else "cmain"   # cmain | c3 | c5 (bench.py's configurations)
lazy_margin = int(sys.argv[4]) if len(sys.argv) > 4 else 0   # > 0: rgc_set_target_lazy
prior = None
if cfg == "cmain":
    world, tgt = synth.make_world_and_map(1000000, seed=synth.SEED)
    poses = synth.make_trajectory(K + 2, seed=synth.SEED)
    scans = [synth.make_scan_n(world, poses[i + 1], 30000, seed=synth.SEED + 100 + i)["xyz"] for i in range(K + 1)]
else:
    world, tile = synth.make_world_and_map(5_000_000, seed=synth.SEED + 7)
    e64 = synth.hdl64_elev()
    if cfg == "c3":
        tgt = tile
        poses = synth.make_trajectory(K + 2, seed=synth.SEED + 7)
        scans = [synth.make_scan_n(world, poses[i + 1], 130000, elev_deg=e64, seed=synth.SEED + 200 + i)["xyz"] for i in range(K + 1)]
    else:
        L = 2.0 * world.half_extent + 4.0
        tgt = np.concatenate([tile, tile + np.float32([L, 0, 0]), tile + np.float32([0, L, 0]), tile + np.float32([L, L, 0])]).astype(np.float32)
        poses = synth.make_trajectory(K + 2, seed=synth.SEED + 9)
        from rgc_slam_amd import odometry
        imu_prior = odometry.imu_rotation_priors(poses)
        scans, prior = [], {}
        for i in range(K + 1):
            a = synth.make_scan_n(world, poses[i + 1], 125000, elev_deg=e64, seed=synth.SEED + 300 + i)["xyz"]
            b = synth.make_scan_n(world, poses[i + 1], 125000, elev_deg=e64 + 0.5 * float(np.abs(np.diff(np.sort(e64))).min()), seed=synth.SEED + 400 + i)["xyz"]
            scans.append(np.concatenate([a, b]).astype(np.float32))
            prior[i] = imu_prior[i + 1]
pv = registration.PipelinedVGICP(0, depth=2)
v = pv.v[0]
for w in pv.v:
    w.setLazyTarget(lazy_margin)
def to_dev(xyz):
    a = np.zeros((xyz.shape[0], 4), np.float32); a[:, :3] = xyz
    p = v.device_alloc(a.nbytes); v.upload(p, a); return p
d_map, d_scans = to_dev(tgt), [to_dev(s) for s in scans]
seq = bench.DependentSequence(pv.v, d_map, len(tgt), d_scans, [len(s) for s in scans])
I4 = np.eye(4, dtype=np.float32)
for w in pv.v:
    seq.v = [w]; seq.run(0, 1, poses[0], I4, False, prior_world=prior)
seq.v = pv.v
seq.run(1, K, poses[0], I4, overlap, prior_world=prior)
seq.run(1, K, poses[0], I4, overlap, prior_world=prior)
pv.synchronize()
print("done")
