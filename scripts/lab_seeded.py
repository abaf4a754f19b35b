"""The map's seeded exact search (knn_point_seeded) on a re-framed 1 M-point map: per frame the dominant launch's time (profiling region) and,
with a developer build (RGC_EXTRA_FLAGS=-DRGC_LAB), the wave-level counts: waves, trips of the unseeded / seeded scan loops, chain insert
rounds, waves that ran the full search after the seeded one declined.
    python scripts/lab_seeded.py [n_target] [frames]"""
import sys, os, json, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rgc_slam_amd.synth as synth
from rgc_slam_amd import registration
import bench
nt = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 6
world, tgt = synth.make_world_and_map(nt, seed=synth.SEED)
poses = synth.make_trajectory(frames + 1, seed=synth.SEED)
v = registration.odometer_vgicp(0)
lib = v._L
lab = hasattr(lib, "rgc_lab_iters")
if lab:
    lib.rgc_lab_iters.argtypes = [C.c_void_p, C.c_void_p]
    lib.rgc_lab_declines.argtypes = [C.c_void_p, C.c_void_p]
decl = np.zeros(16, np.int32)
a = np.zeros((nt, 4), np.float32); a[:, :3] = tgt
d_map, d_body = v.device_alloc(a.nbytes), v.device_alloc(a.nbytes)
v.upload(d_map, a)
it = np.zeros(8, np.uint64)
rows = []
for f in range(frames):
    q, t = bench.world_to_body(np.asarray(poses[f], np.float64))
    v.profile_enable(True); v.profile_reset()
    if lab: lib.rgc_lab_iters(v._h, it.ctypes.data); lib.rgc_lab_declines(v._h, decl.ctypes.data)
    v.setInputTargetReframed(d_map, nt, 16, q, t, d_body)
    v.synchronize()
    p = v.profile()
    row = {"frame": f, "ms": {k: round(x["total_ms"], 4) for k, x in p.items() if x["launches"]}, "deferred": v.stats()["deferred_target"], "searched": v.stats()["searched_target"]}
    if lab:
        lib.rgc_lab_iters(v._h, it.ctypes.data)
        lib.rgc_lab_declines(v._h, decl.ctypes.data)
        row["declined_lanes"] = dict(zip(["no seed", "crowded row", "fewer than k", "more than k+1", "k keys undecided", "k+2 may contend", "three contenders", "exact tie"], decl[1:9].tolist()))
        w = float(it[0]) if it[0] else 1.0
        row["lab"] = {"waves_full_search": int(it[0]), "trips_full": round(float(it[1]) / w, 2) if it[0] else 0, "insert_rounds": round(float(it[2]) / w, 2) if it[0] else 0,
                      "tie_breaks": int(it[5]), "waves_after_seeded_declined": int(it[6]), "seeded_trips_total": int(it[7]),
                      "seeded_trips_per_wave": round(float(it[7]) / (nt / 64.0), 2)}
    rows.append(row)
    print(json.dumps(row), flush=True)
v.close()
