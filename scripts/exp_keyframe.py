"""Why is a keyframe frame slow?  The bench's keyframe_every_3rd_frame sequence (RGC_REUSE_NONE), per-frame host times and counters.
    RGC_TRACE_ALLOC=1 python scripts/exp_keyframe.py [variant: rows|noupload|selfcopy]"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
import rgc_slam_amd.synth as synth
from rgc_slam_amd import registration
variant = sys.argv[1] if len(sys.argv) > 1 else "rows"
K, W, seed = 24, 0, synth.SEED
world, tgt = synth.make_world_and_map(1000000, seed=seed)
poses = synth.make_trajectory(K + 2, seed=seed)
scans = [synth.make_scan_n(world, poses[i + 1], 30000, seed=seed + 100 + i)["xyz"] for i in range(K + 1)]
pv = registration.PipelinedVGICP(0, depth=2); v = pv.v[0]
for w in pv.v: w.setNeighbourReuse(0)
n_map = len(tgt)
map_host = np.zeros((n_map, 4), np.float32); map_host[:, :3] = tgt
perm = np.random.default_rng(seed + 499).permutation(n_map)
map_perm = np.ascontiguousarray(map_host[perm])
def to_dev(a):
    p = v.device_alloc(a.nbytes); v.upload(p, a); return p
d_map = to_dev(map_perm)
def s4(x):
    a = np.zeros((len(x), 4), np.float32); a[:, :3] = x; return a
d_scans = [to_dev(s4(s)) for s in scans]
seq = bench.DependentSequence(pv.v, d_map, n_map, d_scans, [len(s) for s in scans])
kf_n = n_map // 100
lo, hi = tgt.min(axis=0) + np.float32(0.5), tgt.max(axis=0) - np.float32(0.5)
kf = torch.zeros((K // 3 + 2, kf_n, 4), dtype=torch.float32).pin_memory().numpy()
for kk in range(len(kf)):
    i = min(3 * kk, len(scans) - 1)
    P = np.asarray(poses[i + 1], np.float64)
    wall = synth.leaf_centroids(scans[i].astype(np.float64) @ P[:3, :3].T + P[:3, 3], 0.3).astype(np.float32)
    wp = wall[np.random.default_rng(seed + 500 + kk).permutation(len(wall))][:kf_n]
    wp = wp[np.all((wp > lo) & (wp < hi), axis=1)]
    slot = kk * kf_n
    kf[kk] = map_perm[slot:slot + kf_n]
    if variant != "selfcopy":
        kf[kk, :len(wp), :3] = wp
    print("keyframe", kk, "points", len(wp), file=sys.stderr)
def edit(i, w):
    if i % 3 == 0 and variant != "noupload":
        w.upload_async(d_map + (i // 3) * kf_n * 16, kf[i // 3])
I4 = np.eye(4, dtype=np.float32); Tw0 = np.asarray(poses[0], np.float64)
for w in pv.v:
    seq.v = [w]; seq.run(0, 1, Tw0, I4, False)
seq.v = pv.v
for rep in range(2):
    v.upload(d_map, map_perm)
    st = [time.perf_counter()]
    rows = []
    def on(i, w):
        s_ = w.stats(); rows.append((i, s_["deferred_target"], s_["deferred_source"], s_["outer_iterations"], s_["n_voxels"]))
    seq.run(0, K, Tw0, I4, True, edit_map=edit, stamps=st, on_result=on if rep == 1 else None)
    pv.synchronize()
    ms = np.diff(st) * 1e3
    print(json.dumps({"variant": variant, "rep": rep, "ms": [round(float(x), 3) for x in ms], "rows": rows}))
