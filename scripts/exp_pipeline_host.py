"""Where the host's time goes in the pipelined loop: seconds inside set_clouds / align_begin (enqueueing) and inside align_end (waiting)."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rgc_slam_amd.synth as synth
from rgc_slam_amd import registration
nt, frames, reps = 1000000, 8, 8
D = int(sys.argv[1]) if len(sys.argv) > 1 else 2
world, tgt = synth.make_world_and_map(nt)
poses = synth.make_trajectory(frames + 1)
scans = [synth.make_scan_n(world, poses[i + 1], 30000, seed=synth.SEED + 100 + i)["xyz"] for i in range(frames)]
pv = registration.PipelinedVGICP(0, depth=D)
v = pv.v[0]
def to_dev(xyz):
    a = np.zeros((xyz.shape[0], 4), np.float32); a[:, :3] = xyz
    p = v.device_alloc(a.nbytes); v.upload(p, a); return p
d_tgt = to_dev(tgt); d_s = [to_dev(s) for s in scans]
acc = {"set_target": 0.0, "set_source": 0.0, "begin": 0.0, "end": 0.0}
def setc(j, w):
    t0 = time.perf_counter(); w.setInputTargetDevice(d_tgt, len(tgt), 16)
    t1 = time.perf_counter(); w.setInputSourceDevice(d_s[j % frames], 30000, 16)
    t2 = time.perf_counter(); acc["set_target"] += t1 - t0; acc["set_source"] += t2 - t1
def run(n):
    for j in range(min(D - 1, n)): setc(j, pv.v[j % D])
    g = poses[0].astype(np.float32)
    for i in range(n):
        cur = pv.v[i % D]
        t0 = time.perf_counter(); cur.align_begin(g, True); acc["begin"] += time.perf_counter() - t0
        j = i + D - 1
        if j < n: setc(j, pv.v[j % D])
        t0 = time.perf_counter(); g = cur.align_end(); acc["end"] += time.perf_counter() - t0
    pv.synchronize()
run(16)
for k in acc: acc[k] = 0.0
n = reps * frames
t0 = time.perf_counter(); run(n); el = time.perf_counter() - t0
print(json.dumps({"depth": D, "us_per_frame": round(el / n * 1e6, 1), **{k + "_us": round(x / n * 1e6, 1) for k, x in acc.items()}}))
