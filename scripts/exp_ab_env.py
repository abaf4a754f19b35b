"""A/B of an environment knob read at context creation, on the c-main workload, alternating within one process:
    python scripts/exp_ab_env.py RGC_SPEC_GRID 0 1"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rgc_slam_amd.synth as synth
from rgc_slam_amd import registration
knob, vals = sys.argv[1], sys.argv[2:]
world, tgt = synth.make_world_and_map(1000000, seed=synth.SEED)
poses = synth.make_trajectory(24, seed=synth.SEED)
scans = [synth.make_scan_n(world, poses[i + 1], 30000, seed=synth.SEED + 100 + i)["xyz"] for i in range(23)]
ctx = {}
for val in vals:
    os.environ[knob] = val
    v = registration.odometer_vgicp(0)
    def to_dev(xyz):
        a = np.zeros((xyz.shape[0], 4), np.float32); a[:, :3] = xyz
        p = v.device_alloc(a.nbytes); v.upload(p, a); return p
    ctx[val] = (v, to_dev(tgt), [to_dev(s) for s in scans])
def run(val, reps):
    v, d_tgt, d_s = ctx[val]
    g = poses[0].astype(np.float32)
    out = []
    for r in range(reps):
        g = poses[0].astype(np.float32)
        for i in range(23):
            if i == 3: v.synchronize(); t0 = time.perf_counter()
            v.setInputTargetDevice(d_tgt, len(tgt), 16); v.setInputSourceDevice(d_s[i], 30000, 16)
            v.align(g, want_output=False, want_fitness=True); g = v.getFinalTransformation()
        v.synchronize()
        out.append(1e3 * (time.perf_counter() - t0) / 20)
    return out, g
for val in vals: run(val, 5)   # spin-up past the runtime's one-time stall
res = {val: [] for val in vals}
for rnd in range(6):
    for val in vals:
        t, g = run(val, 3)
        res[val] += t
for val in vals:
    a = np.array(res[val])
    print(knob, "=", val, "ms/frame median %.4f min %.4f mean %.4f" % (np.median(a), a.min(), a.mean()))
