"""Developer aid: registration statistics (cloud sizes, deferred queries, crowding, iterations) of the frames of the odometry node's workload."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rgc_slam_amd.synth as synth
from rgc_slam_amd import odometry
world = synth.make_world(half_extent=45.0, seed=synth.SEED)
poses = synth.make_trajectory(25, seed=synth.SEED)
hb = odometry.HipBackend(0)
od = odometry.RollingOdometer(hb)
for k in range(12):
    sc = synth.make_scan(world, poses[k], n_az=1800, seed=synth.SEED + 50 + k, T_ws_end=poses[k + 1])
    od.process(np.concatenate([sc["xyz"], sc["intensity"][:, None]], axis=1).astype(np.float32))
    if k >= 8: print(hb.reg.stats())
