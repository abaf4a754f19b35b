"""Developer aid (library built with RGC_EXTRA_FLAGS=-DRGC_LAB): phase timestamps inside k_lm_step (100 MHz wall clock) --
entry (earliest workgroup), per-point work done (earliest), row stored (earliest), last arriver past the ticket, rows folded,
decision taken, LM try done, state written.  max_iterations = 1: one LIN step (slots 0-7) and one BA step (slots 8-15) per solve."""
import sys, os, json, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rgc_slam_amd.synth as synth
from rgc_slam_amd import registration, _lib
lib = _lib.load()
lib.rgc_lab_lm_ts.argtypes = [C.c_void_p, C.c_void_p]
world, tgt = synth.make_world_and_map(1000000, seed=synth.SEED)
poses = synth.make_trajectory(4, seed=synth.SEED)
src = synth.make_scan_n(world, poses[1], 30000, seed=synth.SEED + 100)["xyz"]
v = registration.odometer_vgicp(0)
v.setMaximumIterations(1)
v.setInputTarget(tgt)
ts = np.zeros(16, np.uint64)
lib.rgc_lab_lm_ts(v._h, ts.ctypes.data)
rows = []
for rep in range(6):
    v.setInputSource(src)
    v.align(np.eye(4, dtype=np.float32))
    lib.rgc_lab_lm_ts(v._h, ts.ctypes.data)
    t = ts.astype(np.int64)
    rows.append([[int(t[b + k] - t[b]) * 10 for k in range(8)] for b in (0, 8)])
for r in rows:
    print("LIN ns since entry", r[0], " BA", r[1])
v.close()
