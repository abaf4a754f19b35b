"""A/B of the bulk kNN implementations (RGC_KNN_IMPL) on the map and the scan of the c-main workload: same covariances, time per stage.
    python scripts/lab_knn.py [n_target] [reps] [impl ...]"""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rgc_slam_amd.synth as synth
from rgc_slam_amd import registration
nt = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
impls = sys.argv[3:] or ["rows", "sp"]
world, tgt = synth.make_world_and_map(nt, seed=synth.SEED)
poses = synth.make_trajectory(4, seed=synth.SEED)
src = synth.make_scan_n(world, poses[1], 30000, seed=synth.SEED + 100)["xyz"]
ref = {}
for impl in impls:
    os.environ["RGC_KNN_IMPL"] = impl
    v = registration.odometer_vgicp(0)
    out = {"impl": impl}
    for name, cloud, setter, getter in (("target", tgt, v.setInputTarget, v.getTargetCovariances), ("source", src, v.setInputSource, v.getSourceCovariances)):
        setter(cloud); v.synchronize()
        v.profile_enable(True); v.profile_reset()
        t0 = time.perf_counter()
        for _ in range(reps):
            setter(cloud)
        v.synchronize()
        wall = (time.perf_counter() - t0) / reps * 1e3
        p = v.profile()
        out[name] = {k: round(x["total_ms"] / reps, 4) for k, x in p.items() if x["launches"]}
        out[name]["wall_ms"] = round(wall, 4)
        v.profile_enable(False)
        cov = getter()
        st = v.stats()
        out[name]["deferred"] = st["deferred_target" if name == "target" else "deferred_source"]
        if name in ref:
            d = np.abs(cov - ref[name]).reshape(len(cov), -1).max(axis=1)
            out[name]["vs_first_impl"] = {"max": float(d.max()), "n_gt_1e-9": int((d > 1e-9).sum()), "n_gt_1e-12": int((d > 1e-12).sum())}
        else:
            ref[name] = cov
    print(json.dumps(out), flush=True)
    v.close()
