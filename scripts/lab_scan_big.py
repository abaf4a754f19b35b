"""Developer aid: the preparation of a c3 / c5-sized raw scan by itself: stage times, deferred queries and (with a -DRGC_LAB library) why
the bulk launch deferred them.   python scripts/lab_scan_big.py [c3|c5]"""
import sys, os, json, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rgc_slam_amd.synth as synth
from rgc_slam_amd import registration, _lib
which = sys.argv[1] if len(sys.argv) > 1 else "c5"
world, tgt = synth.make_world_and_map(200000, seed=synth.SEED + 7)
poses = synth.make_trajectory(4, seed=synth.SEED + 9)
e64 = synth.hdl64_elev()
if which == "c3":
    src = synth.make_scan_n(world, poses[1], 130000, elev_deg=e64, seed=synth.SEED + 200)["xyz"]
else:
    a = synth.make_scan_n(world, poses[1], 125000, elev_deg=e64, seed=synth.SEED + 300)["xyz"]
    b = synth.make_scan_n(world, poses[1], 125000, elev_deg=e64 + 0.5 * float(np.abs(np.diff(np.sort(e64))).min()), seed=synth.SEED + 400)["xyz"]
    src = np.concatenate([a, b]).astype(np.float32)
v = registration.odometer_vgicp(0)
v.setInputTarget(tgt)
for rep in range(3):
    v.setInputSource(src); v.align(np.eye(4, dtype=np.float32))
lib = _lib.load()
why = None
if hasattr(lib, "rgc_lab_why"):
    lib.rgc_lab_why.argtypes = [C.c_void_p, C.c_void_p]
    w8 = np.zeros(8, np.int32); lib.rgc_lab_why(v._h, w8.ctypes.data)
v.profile_enable(True); v.profile_select(["grid_build", "knn_cov_source", "knn_coop_source"]); v.profile_reset()
for rep in range(5):
    v.setInputSource(src); v.synchronize()
p = v.profile()
if hasattr(lib, "rgc_lab_why"):
    lib.rgc_lab_why(v._h, w8.ctypes.data); why = (w8 / 5).tolist()
st = v.stats()
print(json.dumps({"scan": which, "n": len(src), "ms": {k: round(x["total_ms"] / max(x["launches"], 1), 4) for k, x in p.items() if x["launches"]},
                  "deferred": st["deferred_source"], "crowding": round(st["source_crowding"], 1), "source_cells": st["source_cells"],
                  "why_per_launch(1 heavy piece,2 ordinals,3 dropped key,4 <k,5 unproven,6 tie)": why}))
v.close()
