"""How much throughput is left on the table by running a frame's stages strictly one after another: two independent sequences on two
contexts of ONE GPU, each driven by its own host thread, against one sequence alone."""
import sys, os, time, threading, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rgc_slam_amd.synth as synth
from rgc_slam_amd import registration
nt, frames, reps = 1000000, 8, 12
world, tgt = synth.make_world_and_map(nt)
poses = synth.make_trajectory(frames + 1)
scans = [synth.make_scan_n(world, poses[i + 1], 30000, seed=synth.SEED + 100 + i)["xyz"] for i in range(frames)]
def make():
    v = registration.odometer_vgicp(0)
    def to_dev(xyz):
        a = np.zeros((xyz.shape[0], 4), np.float32); a[:, :3] = xyz
        p = v.device_alloc(a.nbytes); v.upload(p, a); return p
    return v, to_dev(tgt), [to_dev(s) for s in scans]
def run(ctx, n_rep, out, key):
    v, d_tgt, d_s = ctx
    g = np.eye(4, dtype=np.float32)
    t0 = time.perf_counter()
    for rep in range(n_rep):
        for i in range(frames):
            v.setInputTargetDevice(d_tgt, len(tgt), 16)
            v.setInputSourceDevice(d_s[i], 30000, 16)
            v.align(g, want_output=False, want_fitness=True)
            g = v.getFinalTransformation()
    v.synchronize()
    out[key] = (time.perf_counter() - t0, n_rep * frames)
a, b = make(), make()
out = {}
run(a, 2, out, "warm_a"); run(b, 2, out, "warm_b")
run(a, reps, out, "alone")
th = [threading.Thread(target=run, args=(c, reps, out, k)) for c, k in ((a, "pair_a"), (b, "pair_b"))]
t0 = time.perf_counter()
for t in th: t.start()
for t in th: t.join()
wall = time.perf_counter() - t0
print(json.dumps({"alone_scans_per_s": round(out["alone"][1] / out["alone"][0], 1),
                  "two_contexts_aggregate_scans_per_s": round(2 * reps * frames / wall, 1)}))
