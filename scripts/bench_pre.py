"""B2/B3/B9 timings on the MI355X against the CPU oracle (host arrays in and out)."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401
import rgc_slam_amd.synth as synth
from rgc_slam_amd import odometry
from oracle import oracle
world, tgt = synth.make_world_and_map(300000)
sc = synth.make_scan_n(world, np.eye(4), 30000)
scan = np.concatenate([sc["xyz"], (sc["ring"] + 0.05)[:, None]], axis=1).astype(np.float32)
sub = np.concatenate([tgt, np.zeros((len(tgt), 1), np.float32)], axis=1)
pre = odometry.Preprocessor(0)
q = np.array([0.001, -0.002, 0.01, 0.99994]); q /= np.linalg.norm(q); t = np.array([0.1, 0.02, 0.0])
def tm(f, reps=20):
    for _ in range(3): f()
    t0 = time.perf_counter()
    for _ in range(reps): f()
    return 1e3 * (time.perf_counter() - t0) / reps
res = {"deskew_30k_ms": [tm(lambda: pre.adjustDistortion(scan, q, t)), tm(lambda: oracle.deskew(scan, q, t), 5)],
       "voxelgrid_0.2_30k_ms": [tm(lambda: pre.voxelGridFilter(scan, 0.2)), tm(lambda: oracle.voxelgrid_filter(scan, 0.2), 5)],
       "voxelgrid_0.3_300k_ms": [tm(lambda: pre.voxelGridFilter(sub, 0.3)), tm(lambda: oracle.voxelgrid_filter(sub, 0.3), 3)],
       "transform_300k_ms": [tm(lambda: pre.transformPointCloud(sub, q, t)), tm(lambda: oracle.transform_cloud(sub, q, t), 5)]}
print(json.dumps({k: {"gpu": round(v[0], 3), "cpu_oracle": round(v[1], 3)} for k, v in res.items()}))
