"""Experiment: the scan's kNN grid at another cell size than the voxel grid (RGC_SRC_RES), scan preparation alone and whole frame."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rgc_slam_amd.synth as synth
from rgc_slam_amd import registration
kind = sys.argv[1] if len(sys.argv) > 1 else "vlp"
world, tgt = synth.make_world_and_map(1000000, seed=synth.SEED)
poses = synth.make_trajectory(8, seed=synth.SEED)
if kind == "vlp":
    scans = [synth.make_scan_n(world, poses[i + 1], 30000, seed=synth.SEED + 100 + i)["xyz"] for i in range(6)]
else:
    scans = [synth.make_scan_n(world, poses[i + 1], 130000, elev_deg=synth.hdl64_elev(), seed=synth.SEED + 100 + i)["xyz"] for i in range(6)]
for res in ("0", "0.5", "0.33", "0.25"):
    if res == "0": os.environ.pop("RGC_SRC_RES", None)
    else: os.environ["RGC_SRC_RES"] = res
    v = registration.odometer_vgicp(0)
    v.setInputTarget(tgt)
    for s in scans: v.setInputSource(s)
    v.synchronize(); v.profile_enable(True); v.profile_reset()
    for rep in range(3):
        for s in scans: v.setInputSource(s)
    v.synchronize()
    p = v.profile()
    src = {k: round(x["total_ms"] / 18, 4) for k, x in p.items() if x["launches"] and k in ("grid_build", "knn_cov_source", "knn_coop_source")}
    v.profile_enable(False)
    g = poses[0].astype(np.float32); fin = []
    for rep in range(3):
        g = poses[0].astype(np.float32)
        v.synchronize(); t0 = time.perf_counter()
        for s in scans:
            v.setInputTarget(tgt); v.setInputSource(s); v.align(g, want_output=False, want_fitness=True); g = v.getFinalTransformation()
        v.synchronize(); el = (time.perf_counter() - t0) / len(scans)
    print(kind, "src_res", res, src, "deferred", v.stats()["deferred_source"], "crowding", round(v.stats()["source_crowding"], 1), "frame ms (host arrays)", round(1e3 * el, 3), "t", g[:3, 3])
    v.close()
