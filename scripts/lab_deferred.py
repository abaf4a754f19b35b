"""Developer aid (library built with RGC_EXTRA_FLAGS=-DRGC_LAB): which queries the bulk kNN kernel deferred, for the map and the scan."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rgc_slam_amd.synth as synth
from rgc_slam_amd import registration, _lib
world, tgt = synth.make_world_and_map(int(sys.argv[1]) if len(sys.argv) > 1 else 1000000, seed=synth.SEED)
poses = synth.make_trajectory(4, seed=synth.SEED)
src = synth.make_scan_n(world, poses[1], 30000, seed=synth.SEED + 100)["xyz"]
lib = _lib.load()
lib.rgc_lab_deferred.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int)]
lib.rgc_lab_why.argtypes = [C.c_void_p, C.c_void_p]
for res in [None] + [float(a) for a in sys.argv[2:]]:
    if res is not None: os.environ["RGC_SRC_RES"] = str(res)
    v = registration.odometer_vgicp(0)
    why = np.zeros(8, np.int32)
    v.setInputTarget(tgt); v.synchronize(); lib.rgc_lab_why(v._h, why.ctypes.data)
    v.setInputSource(src); v.synchronize(); lib.rgc_lab_why(v._h, why.ctypes.data)
    print("scan bulk kernel deferred because: piece too long", why[1], "ordinals", why[2], "dropped key", why[3], "< k in block", why[4], "unproven", why[5], "tie", why[6])
    for which, cloud in ((1, tgt), (0, src)):
        if res is not None and which == 1: continue
        cap = len(cloud)
        idx = np.zeros(cap, np.int32); thr = np.zeros(cap, np.float32); cnt = C.c_int(0)
        lib.rgc_lab_deferred(v._h, which, idx.ctypes.data, thr.ctypes.data, cap, C.byref(cnt))
        n = cnt.value; idx, thr = idx[:n], thr[:n]
        neg = idx < 0
        rng = np.linalg.norm(cloud[np.where(neg, ~idx, idx)], axis=1) if n else np.zeros(0)
        print("src_res", res, "target" if which else "source", "n", len(cloud), "deferred", n, "| unproven (~i):", int(neg.sum()), "of which thr=inf", int((neg & np.isinf(thr)).sum()),
              "| heavy/unsafe (i, inf):", int((~neg & np.isinf(thr)).sum()), "| undecided (i, finite):", int((~neg & np.isfinite(thr)).sum()),
              "| range of deferred pts pcts", np.percentile(rng, [10, 50, 90]).round(1) if n else None, flush=True)
    v.close()
