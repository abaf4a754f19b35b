"""keyframe_every_3rd_frame as bench.py builds it (a block of rows re-sampled in place), default reuse mode: per frame, was the change seen?"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
import rgc_slam_amd.synth as synth
from rgc_slam_amd import registration
K, seed = 12, synth.SEED
world, tgt = synth.make_world_and_map(1000000, seed=seed)
poses = synth.make_trajectory(K + 2, seed=seed)
scans = [synth.make_scan_n(world, poses[i + 1], 30000, seed=seed + 100 + i)["xyz"] for i in range(K + 1)]
alt = synth.make_map(world, None, seed=seed + 77)
pv = registration.PipelinedVGICP(0, depth=2); v = pv.v[0]
n_map = len(tgt)
map_host = np.zeros((n_map, 4), np.float32); map_host[:, :3] = tgt
def to_dev(a):
    p = v.device_alloc(a.nbytes); v.upload(p, a); return p
d_map = to_dev(map_host)
def s4(x):
    a = np.zeros((len(x), 4), np.float32); a[:, :3] = x; return a
d_scans = [to_dev(s4(s)) for s in scans]
seq = bench.DependentSequence(pv.v, d_map, n_map, d_scans, [len(s) for s in scans])
kf_n = n_map // 100
kf = torch.zeros((K // 3 + 2, kf_n, 4), dtype=torch.float32).pin_memory().numpy()
for kk in range(len(kf)):
    slot = kk * kf_n
    a_lo = int(slot * (len(alt) / float(n_map)))
    kf[kk, :, :3] = alt[a_lo:a_lo + kf_n]
    print("kf", kk, "rows differ:", int(np.any(kf[kk, :, :3] != map_host[slot:slot + kf_n, :3], axis=1).sum()), "z range", kf[kk, :, 2].min(), kf[kk, :, 2].max(), file=sys.stderr)
def edit(i, w):
    if i % 3 == 0:
        w.upload_async(d_map + (i // 3) * kf_n * 16, kf[i // 3])
I4 = np.eye(4, dtype=np.float32); Tw0 = np.asarray(poses[0], np.float64)
rows = []
def on(i, w):
    rows.append((i, w.stats()["searched_target"]))
st = [time.perf_counter()]
m_e, _, _ = seq.run(0, K, Tw0, I4, True, edit_map=edit, stamps=st, on_result=on)
pv.synchronize()
v.upload(d_map, map_host)
m_0, _, _ = seq.run(0, K, Tw0, I4, True)
back = v.download(d_map, (n_map, 4))
print(json.dumps({"searched": rows, "ms": [round(float(x) * 1e3, 3) for x in np.diff(st)], "poses_differ_from_unedited": [bool(not np.array_equal(a, b)) for a, b in zip(m_e, m_0)]}))
