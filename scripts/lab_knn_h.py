"""Developer aid: the map's preparation on the voxel grid (RGC_MAP_HALF=0) and on the half-size search grid, stage by stage, with
the deferred counts -- and, when the library was built with RGC_EXTRA_FLAGS=-DRGC_LAB, why the bulk kernel deferred.
    python scripts/lab_knn_h.py [n_target] [reps]"""
import sys, os, time, json, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rgc_slam_amd.synth as synth
from rgc_slam_amd import registration, _lib
nt = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
world, tgt = synth.make_world_and_map(nt, seed=synth.SEED)
lib = _lib.load()
has_lab = hasattr(lib, "rgc_lab_why")
if has_lab:
    lib.rgc_lab_why.argtypes = [C.c_void_p, C.c_void_p]
normals = {}
for half in ("0", "1"):
    os.environ["RGC_MAP_HALF"] = half
    v = registration.odometer_vgicp(0)
    why = np.zeros(8, np.int32)
    v.setInputTarget(tgt); v.synchronize()
    if has_lab:
        lib.rgc_lab_why(v._h, why.ctypes.data)
    deferred = v.stats()["deferred_target"]
    v.setInputTarget(tgt); v.synchronize()
    v.profile_enable(True); v.profile_reset()
    t0 = time.perf_counter()
    for _ in range(reps):
        v.setInputTarget(tgt)
    v.synchronize()
    wall = (time.perf_counter() - t0) / reps * 1e3
    p = v.profile()
    out = {k: round(x["total_ms"] / reps, 4) for k, x in p.items() if x["launches"]}
    out.update(wall_ms=round(wall, 4), deferred=int(deferred), n_voxels=v.stats()["n_voxels"], cells=v.stats()["target_cells"])
    if has_lab:
        out["why"] = dict(row_too_long=int(why[1]), table_full=int(why[2]), fewer_than_k=int(why[4]), unproven=int(why[5]), tie=int(why[6]))
    v.profile_enable(False)
    normals[half] = v.getTargetNormals()
    print("RGC_MAP_HALF=" + half, json.dumps(out), flush=True)
    v.close()
a, b = normals["0"], normals["1"]
s = np.sign(np.sum(a * b, axis=1, keepdims=True))
print("normals of the two layouts differ by at most", float(np.abs(a - s * b).max()))
