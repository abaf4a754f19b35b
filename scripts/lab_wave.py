"""Developer aid (library built with RGC_EXTRA_FLAGS=-DRGC_LAB): when every wave of the scan's bulk kNN launch started and ended
(100 MHz wall clock), standalone and underneath the map's launch."""
import sys, os, json, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rgc_slam_amd.synth as synth
from rgc_slam_amd import registration, _lib
lib = _lib.load()
lib.rgc_lab_wave_ts.argtypes = [C.c_void_p, C.c_void_p]
world, tgt = synth.make_world_and_map(1000000, seed=synth.SEED)
poses = synth.make_trajectory(4, seed=synth.SEED)
src = synth.make_scan_n(world, poses[1], 30000, seed=synth.SEED + 100)["xyz"]
v = registration.odometer_vgicp(0)
ts = np.zeros(2 * 8192, np.int64)
def report(tag):
    lib.rgc_lab_wave_ts(v._h, ts.ctypes.data)
    nw = (4 * len(src) + 63) // 64                             # four lanes per query, one-wave workgroups
    nd = v.stats()["deferred_source"]                          # (the cooperative kernel's waves overwrite the first `deferred` slots)
    m48 = (1 << 48) - 1
    t = (ts.reshape(-1, 2)[nd:min(nw, 8192)] & m48).astype(np.float64) * 0.01       # us
    t0 = t[:, 0].min()
    dur = t[:, 1] - t[:, 0]
    order = np.argsort(-dur)
    print(tag, "waves", nw, "launch span us", round(t[:, 1].max() - t0, 1), "last start", round(t[:, 0].max() - t0, 1),
          "wave us: median", round(float(np.median(dur)), 1), "p90", round(float(np.percentile(dur, 90)), 1), "max", round(float(dur.max()), 1),
          "sum", round(float(dur.sum()), 0), "slowest waves", (order[:8] + nd).tolist(), [round(float(dur[o]), 1) for o in order[:8]],
          "waves over 20 us", int((dur > 20).sum()), "over 30 us", int((dur > 30).sum()), "deferred", nd)
for rep in range(3):
    v.setInputSource(src); v.synchronize()
    report("standalone")
v.setInputTarget(tgt); v.synchronize()
for rep in range(3):
    v.setInputTarget(tgt); v.setInputSource(src); v.synchronize()
    report("under the map launch")
v.close()
