"""Experiment: kNN kernel time by density regime of a raw VLP-16 scan (near = dense cells, far = sparse)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rgc_slam_amd.synth as synth
from rgc_slam_amd import registration

world, tgt = synth.make_world_and_map(int(sys.argv[1]) if len(sys.argv) > 1 else 200000)
scan = synth.make_scan_n(world, np.eye(4), 30000)["xyz"]
r = np.linalg.norm(scan, axis=1)
v = registration.odometer_vgicp(0)
v.profile_enable(True)
def run(name, pts, target=False, reps=3):
    v.profile_reset()
    for _ in range(reps):
        (v.setInputTarget if target else v.setInputSource)(pts)
    v.synchronize()
    p = v.profile()
    key = "knn_cov_target" if target else "knn_cov_source"
    print(f"{name:28s} n={len(pts):8d} knn={p[key]['total_ms']/reps:8.3f} ms  grid={p['grid_build']['total_ms']/reps:7.3f} ms  voxel={p['voxel_build']['total_ms']/reps:7.3f}")
run("full scan", scan)
run("near r<6", scan[r < 6])
run("mid 6<r<25", scan[(r >= 6) & (r < 25)])
run("far r>25", scan[r >= 25])
run("far r>40", scan[r >= 40])
run("map as target", tgt, target=True)
run("map as source", tgt)
# cell occupancy of the scan
c = np.floor(scan - 0.5).astype(np.int64)
_, cnt = np.unique(c, axis=0, return_counts=True)
print("scan cells:", len(cnt), "max", cnt.max(), "mean", cnt.mean(), "pts in cells>64:", cnt[cnt > 64].sum(), ">256:", cnt[cnt > 256].sum())
