"""Developer aid (library built with RGC_EXTRA_FLAGS=-DRGC_LAB): which queries the bulk kNN kernel deferred, and why they are slow."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rgc_slam_amd.synth as synth
from rgc_slam_amd import registration, _lib
world, tgt = synth.make_world_and_map(1000000, seed=synth.SEED)
v = registration.odometer_vgicp(0)
v.setInputTarget(tgt); v.synchronize()
lib = _lib.load()
cap = 100000
idx = np.zeros(cap, np.int32); thr = np.zeros(cap, np.float32); cnt = C.c_int(0)
lib.rgc_lab_deferred.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int)]
h = v._h if hasattr(v, "_h") else v.ctx
rc = lib.rgc_lab_deferred(h, 1, idx.ctypes.data, thr.ctypes.data, cap, C.byref(cnt))
n = cnt.value
print("rc", rc, "deferred", n)
idx, thr = idx[:n], thr[:n]
neg = idx < 0
print("scanned-but-unproven (~i):", int(neg.sum()), " not scanned / undecided (i):", int((~neg).sum()), " thr=inf:", int(np.isinf(thr).sum()), "nan:", int(np.isnan(thr).sum()))
print("thr (finite) percentiles", np.percentile(thr[np.isfinite(thr)], [0, 50, 90, 100]) if np.isfinite(thr).any() else None)
print(list(zip(idx[:40].tolist(), thr[:40].tolist())))
