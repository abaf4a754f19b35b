"""Developer aid (library built with RGC_EXTRA_FLAGS=-DRGC_LAB): per deferred query of the scan, when the cooperative kernel started and
finished it (100 MHz wall clock), the final cube radius and the number of search rounds."""
import sys, os, json, ctypes as C
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
ts = np.zeros(2 * 8192, np.int64)
v.setInputTarget(tgt)
for rep in range(4):
    v.setInputSource(src); v.synchronize()
    if rep == 0:
        v.align(np.eye(4, dtype=np.float32)); v.setInputSource(src); v.synchronize()   # the align carries the deferred count home: the next launch is sized from it
    nd = v.stats()["deferred_source"]
    lib.rgc_lab_wave_ts(v._h, ts.ctypes.data)
    t = ts.reshape(-1, 2)[:min(nd, 8192)]
    r = (t[:, 0] >> 56) & 0xff; rounds = (t[:, 0] >> 48) & 0xff
    t0 = (t[:, 0] & ((1 << 48) - 1)).astype(np.float64) * 0.01; t1 = (t[:, 1] & ((1 << 48) - 1)).astype(np.float64) * 0.01
    dur = t1 - t0
    print("deferred", nd, "span us", round(t1.max() - t0.min(), 1), "first start spread", round(t0.max() - t0.min(), 1), "query us: median", round(float(np.median(dur)), 1),
          "p90", round(float(np.percentile(dur, 90)), 1), "max", round(float(dur.max()), 1), "| radius hist", np.bincount(r)[:12].tolist(), "rounds hist", np.bincount(rounds)[:8].tolist())
    o = np.argsort(-dur)[:6]
    print("  slowest:", [(round(float(dur[j]), 1), int(r[j]), int(rounds[j])) for j in o])
v.close()
