"""Steady state of the DEPENDENT c-main sequence (bench.DependentSequence: every frame's target is the map re-expressed on the device in the
previous pose's body frame): the same 40 frames run REPS times from the same start, on one context and on two taking turns -- every
repetition must give the first one's poses bit for bit (hinted grids, pre-sized cell arrays, posted state, held scan preparation change no
result) -- with per-frame wall time statistics and device memory sampled after the second repetition and at the end."""
import sys, json, os, time, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
import rgc_slam_amd.synth as synth
from rgc_slam_amd import registration
REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 50
K = 40
world, tgt = synth.make_world_and_map(1000000, seed=synth.SEED)
poses = synth.make_trajectory(K + 2, seed=synth.SEED)
scans = [synth.make_scan_n(world, poses[i + 1], 30000, seed=synth.SEED + 100 + i)["xyz"] for i in range(K + 1)]
pv = registration.PipelinedVGICP(0, depth=2)
v = pv.v[0]
def to_dev(xyz):
    a = np.zeros((xyz.shape[0], 4), np.float32); a[:, :3] = xyz
    p = v.device_alloc(a.nbytes); v.upload(p, a); return p
d_map, d_scans = to_dev(tgt), [to_dev(s) for s in scans]
seq = bench.DependentSequence(pv.v, d_map, len(tgt), d_scans, [len(s) for s in scans])
I4 = np.eye(4, dtype=np.float32)
for w in pv.v:
    seq.v = [w]; _, w0, _ = seq.run(0, 1, poses[0], I4, False)
seq.v = pv.v
Tw1 = w0[0]
out = {"frames_per_repetition": K, "repetitions": REPS}
for mode, overlap in (("one_context", False), ("two_contexts", True)):
    gc.collect(); gc.freeze()
    ref, per, free2 = None, [], None
    for r in range(REPS):
        pv.synchronize(); t0 = time.perf_counter()
        m, _, _ = seq.run(1, K, Tw1, I4, overlap)
        pv.synchronize(); per.append(1e3 * (time.perf_counter() - t0) / K)
        if ref is None: ref = m
        else: assert all(np.array_equal(a, b) for a, b in zip(ref, m)), f"{mode}: repetition {r} differs from the first"
        if r == 1: free2 = torch.cuda.mem_get_info()[0]
    free_end = torch.cuda.mem_get_info()[0]
    per = np.array(per[2:])
    out[mode] = {"ms_per_frame_median": round(float(np.median(per)), 4), "ms_per_frame_min_max": [round(float(per.min()), 4), round(float(per.max()), 4)],
                 "scans_per_s_median": round(1e3 / float(np.median(per)), 1), "growth_MiB_after_the_second_repetition": round((free2 - free_end) / 2**20, 2),
                 "identical_poses_every_repetition": True}
    if mode == "one_context": first = ref
    else: out["two_contexts_equal_one_context"] = bool(all(np.array_equal(a, b) for a, b in zip(first, ref)))
print(json.dumps(out))
