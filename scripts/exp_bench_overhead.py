"""Why does bench.py's timed loop (first pass over its scans, stats + dominant-kernel events on) run slower per frame than a plain
warm loop over the same frames?  Times pass 1, 2, 3 over the 20 timed frames with and without the per-frame extras."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rgc_slam_amd.synth as synth
from rgc_slam_amd import registration
world, tgt = synth.make_world_and_map(1000000, seed=synth.SEED)
poses = synth.make_trajectory(24, seed=synth.SEED)
scans = [synth.make_scan_n(world, poses[i + 1], 30000, seed=synth.SEED + 100 + i)["xyz"] for i in range(23)]
v = registration.odometer_vgicp(0)
def to_dev(xyz):
    a = np.zeros((xyz.shape[0], 4), np.float32); a[:, :3] = xyz
    p = v.device_alloc(a.nbytes); v.upload(p, a); return p
d_tgt = to_dev(tgt); d_s = [to_dev(s) for s in scans]
def step(i, g):
    v.setInputTargetDevice(d_tgt, len(tgt), 16); v.setInputSourceDevice(d_s[i], 30000, 16)
    v.align(g, want_output=False, want_fitness=True); return v.getFinalTransformation()
g0 = poses[0].astype(np.float32)
for j in range(96): step(j % 3, g0)
v.synchronize()
def timed(extras):
    g = g0
    if extras:   # events on during the warm-up frames: the first event regions of a context are slow (pool creation, lazy runtime set-up)
        v.profile_enable(True); v.profile_select(["knn_cov_target"])
    for i in range(3): g = step(i, g)
    v.synchronize()
    if extras:
        v.profile_reset()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fin, gi, pf = [], [], []
    per = []
    for i in range(3, 23):
        tf = time.perf_counter()
        gi.append(g); g = step(i, g); fin.append(g)
        per.append(round(1e3 * (time.perf_counter() - tf), 3))
        if per[-1] > 0.9:
            st0 = v.stats()
            print("slow frame", i, per[-1], {k: st0[k] for k in ("outer_iterations", "n_linearize", "deferred_source", "deferred_target", "source_cells", "target_cells")}, file=sys.stderr, flush=True)
        if extras:
            st = v.stats(); pf.append((st["outer_iterations"], st["n_linearize"], st["n_error"], st["n_corr"], st["n_voxels"]))
    v.synchronize(); torch.cuda.synchronize()
    el = time.perf_counter() - t0
    if extras: v.profile_enable(False)
    print(per, file=sys.stderr)
    return round(1e3 * el / 20, 4)
print("pass1 extras", timed(True), "pass2 extras", timed(True), "pass3 plain", timed(False), "pass4 plain", timed(False), "pass5 extras", timed(True))
