"""Developer aid (RGC_LAB build): deferred scan queries against a brute-force k-NN: was deferring them necessary?"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from scipy.spatial import cKDTree
import rgc_slam_amd.synth as synth
from rgc_slam_amd import registration, _lib
world, tgt = synth.make_world_and_map(100000, seed=synth.SEED)
poses = synth.make_trajectory(4, seed=synth.SEED)
src = synth.make_scan_n(world, poses[1], 30000, seed=synth.SEED + 100)["xyz"]
res = float(sys.argv[1]) if len(sys.argv) > 1 else 0.5
os.environ["RGC_SRC_RES"] = str(res)
lib = _lib.load()
lib.rgc_lab_deferred.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int)]
v = registration.odometer_vgicp(0)
v.setInputTarget(tgt); v.setInputSource(src); v.synchronize()
cap = len(src); idx = np.zeros(cap, np.int32); thr = np.zeros(cap, np.float32); cnt = C.c_int(0)
lib.rgc_lab_deferred(v._h, 0, idx.ctypes.data, thr.ctypes.data, cap, C.byref(cnt))
n = cnt.value; idx, thr = idx[:n], thr[:n]
# idx refers to the SORTED array; recover the point through the library's own sorted order is not exposed: use coordinates via normals? no:
# the deferred list holds sorted positions, so rebuild the sort here (same cell function, stable by original index)
cc = np.floor(src.astype(np.float64) / res - 0.5).astype(np.int64)
minc = cc.min(0); dim = cc.max(0) - minc + 1
print("own bbox grid dims", dim, "(the library's speculative grid may be wider: sorted positions can differ!)")
tree = cKDTree(src.astype(np.float64))
d, _ = tree.query(src.astype(np.float64), k=20)
kth = d[:, -1]
c = cc - minc
wall = np.minimum((src / res - 0.5) - np.floor(src / res - 0.5), 1 - ((src / res - 0.5) - np.floor(src / res - 0.5))) * res  # distance to own cell walls per axis
bound = (wall + res).min(axis=1)   # distance to the 3x3x3 block boundary
provable = kth < bound * (1 - 1e-5)
print("points", len(src), "provable inside 3x3x3 at res", res, ":", int(provable.sum()), "not provable:", int((~provable).sum()), "| deferred by the kernel:", n)
lin = (c[:, 2] * dim[1] + c[:, 1]) * dim[0] + c[:, 0]
cnt_cell = np.bincount(lin)
tot = np.zeros(len(src), np.int64)
occ = {}
from collections import Counter
cell_count = Counter(map(tuple, c))
for dx in (-1, 0, 1):
    for dy in (-1, 0, 1):
        for dz in (-1, 0, 1):
            tot += np.array([cell_count.get((a + dx, b + dy, e + dz), 0) for a, b, e in c[:3000]]).sum() * 0
print("kth distance pcts", np.percentile(kth, [10, 50, 90, 99]).round(3), "bound pcts", np.percentile(bound, [10, 50, 90]).round(3))
