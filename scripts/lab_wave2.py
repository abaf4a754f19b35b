"""Developer aid (-DRGC_LAB): start / end of every workgroup of the scan's bulk kNN launch (100 MHz wall clock)."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rgc_slam_amd.synth as synth
from rgc_slam_amd import registration, _lib
lib = _lib.load()
lib.rgc_lab_wave_ts.argtypes = [C.c_void_p, C.c_void_p]
world, tgt = synth.make_world_and_map(100000, seed=synth.SEED)
poses = synth.make_trajectory(4, seed=synth.SEED)
src = synth.make_scan_n(world, poses[1], 30000, seed=synth.SEED + 100)["xyz"]
v = registration.odometer_vgicp(0)
v.setInputTarget(tgt)
ts = np.zeros(2 * 8192, np.int64)
nb = (4 * len(src) + 255) // 256
for rep in range(4):
    v.setInputSource(src); v.synchronize()
    lib.rgc_lab_wave_ts(v._h, ts.ctypes.data)
    t = (ts.reshape(-1, 2)[:nb] & ((1 << 48) - 1)).astype(np.float64) * 0.01
    t0 = t[:, 0].min(); dur = t[:, 1] - t[:, 0]
    print("blocks", nb, "span", round(t[:, 1].max() - t0, 1), "last start", round(t[:, 0].max() - t0, 1), "dur median", round(float(np.median(dur)), 1),
          "p90", round(float(np.percentile(dur, 90)), 1), "max", round(float(dur.max()), 1), "starts after 5us", int((t[:, 0] - t0 > 5).sum()))
v.close()
