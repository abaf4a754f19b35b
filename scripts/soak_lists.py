"""Soak of the neighbour lists: a 1 M-point map handed over by rgc_set_target_reframed under random poses (any rotation, up to +-400 m)
with an occasional edit of the buffer, every covariance compared bit for bit with a context that keeps neither seeds nor lists.
    python scripts/soak_lists.py [frames] [n_target]"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rgc_slam_amd.synth as synth
from rgc_slam_amd import registration
import bench
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 60
nt = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
world, tgt = synth.make_world_and_map(nt, seed=synth.SEED)
a = np.zeros((nt, 4), np.float32); a[:, :3] = tgt
v = registration.odometer_vgicp(0)
os.environ["RGC_KNN_SEEDS"] = "0"
w = registration.odometer_vgicp(0)
del os.environ["RGC_KNN_SEEDS"]
dm, db = v.device_alloc(a.nbytes), v.device_alloc(a.nbytes)
dmw, dbw = w.device_alloc(a.nbytes), w.device_alloc(a.nbytes)
v.upload(dm, a); w.upload(dmw, a)
rng = np.random.default_rng(5)
bad, searched = 0, []
for f in range(frames):
    if f and f % 13 == 0:                       # an edit: a handful of points moved
        j = rng.integers(0, nt, 5)
        a[j, :3] += rng.normal(0, 0.2, (5, 3)).astype(np.float32)
        v.upload(dm, a); w.upload(dmw, a)
    ang = rng.uniform(-np.pi, np.pi, 3) * np.array([1.0, 0.02, 0.02])
    scale = 400.0 if f % 7 == 6 else 30.0
    Tw = synth.se3(synth.rot_zyx(*ang), rng.uniform(-scale, scale, 3) * np.array([1, 1, 0.05]))
    q, t = bench.world_to_body(Tw)
    v.setInputTargetReframed(dm, nt, 16, q, t, db)
    w.setInputTargetReframed(dmw, nt, 16, q, t, dbw)
    cv, cw = v.getTargetCovariances(), w.getTargetCovariances()
    same = bool(np.array_equal(cv, cw))
    xv, xw = v.getVoxels(), w.getVoxels()
    same = same and np.array_equal(xv["coords"], xw["coords"]) and np.array_equal(xv["cov"], xw["cov"]) and np.array_equal(xv["mean"], xw["mean"])
    bad += 0 if same else 1
    searched.append(int(v.stats()["searched_target"]))
    if not same:
        d = np.nonzero(np.any(cv.reshape(nt, -1) != cw.reshape(nt, -1), axis=1))[0]
        print("frame", f, "DIFFERS in", len(d), "points", d[:8].tolist(), flush=True)
print(json.dumps({"frames": frames, "n_target": nt, "frames_that_differ": bad, "searched_per_frame": searched}))
