"""One sequence, frames one at a time (FastVGICP.align) against the two-context pipeline (PipelinedVGICP): scans/s and that the
poses are the same."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rgc_slam_amd.synth as synth
from rgc_slam_amd import registration
nt, frames, reps = 1000000, 8, 8
world, tgt = synth.make_world_and_map(nt)
poses = synth.make_trajectory(frames + 1)
scans = [synth.make_scan_n(world, poses[i + 1], 30000, seed=synth.SEED + 100 + i)["xyz"] for i in range(frames)]
v = registration.odometer_vgicp(0)
def to_dev(xyz):
    a = np.zeros((xyz.shape[0], 4), np.float32); a[:, :3] = xyz
    p = v.device_alloc(a.nbytes); v.upload(p, a); return p
d_tgt = to_dev(tgt); d_s = [to_dev(s) for s in scans]
def seq(n_rep):
    g = poses[0].astype(np.float32); out = []
    for rep in range(n_rep):
        for i in range(frames):
            v.setInputTargetDevice(d_tgt, len(tgt), 16)
            v.setInputSourceDevice(d_s[i], 30000, 16)
            v.align(g, want_output=False, want_fitness=True)
            g = v.getFinalTransformation(); out.append(g)
    v.synchronize()
    return out
pv = registration.PipelinedVGICP(0, depth=int(sys.argv[1]) if len(sys.argv) > 1 else 3)
def setc(j, w):
    w.setInputTargetDevice(d_tgt, len(tgt), 16)
    w.setInputSourceDevice(d_s[j % frames], 30000, 16)
def pipe(n_rep):
    out = pv.run(n_rep * frames, setc, poses[0].astype(np.float32), want_fitness=True)
    pv.synchronize()
    return out
seq(2); pipe(2)
t0 = time.perf_counter(); a = seq(reps); ta = time.perf_counter() - t0
t0 = time.perf_counter(); b = pipe(reps); tb = time.perf_counter() - t0
print(json.dumps({"one_at_a_time_scans_per_s": round(reps * frames / ta, 1), "pipelined_scans_per_s": round(reps * frames / tb, 1),
                  "max_abs_pose_difference": float(max(np.abs(x - y).max() for x, y in zip(a, b)))}))
