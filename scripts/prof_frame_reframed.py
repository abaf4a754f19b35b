"""Small driver for rocprofv3: N preparations of a 1 M-point map handed over by rgc_set_target_reframed (the first launch of the dominant
kernel is the unseeded k_knn_sp<20, true, true, false>, the others the seeded k_knn_sp<20, true, true, true>), nothing else.
    python scripts/prof_frame_reframed.py [n_target] [frames]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rgc_slam_amd.synth as synth
from rgc_slam_amd import registration
import bench
nt = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 5
world, tgt = synth.make_world_and_map(nt, seed=synth.SEED)
poses = synth.make_trajectory(frames + 1, seed=synth.SEED)
v = registration.odometer_vgicp(0)
a = np.zeros((nt, 4), np.float32); a[:, :3] = tgt
d_map, d_body = v.device_alloc(a.nbytes), v.device_alloc(a.nbytes)
v.upload(d_map, a)
for f in range(frames):
    q, t = bench.world_to_body(np.asarray(poses[f], np.float64))
    v.setInputTargetReframed(d_map, nt, 16, q, t, d_body)
    v.synchronize()
print("done", v.stats())
