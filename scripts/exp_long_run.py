"""Steady state over a long sequence: 3000 c-main frames (23 scans cycled forward and backward so that consecutive frames are consecutive
poses), per-frame wall time statistics, drift of the rate, and device memory sampled before the first frame, at frame 200 (the working set
is allocated by then) and at the end: working set and steady-state growth are separate numbers."""
import sys, json, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gc
import numpy as np, torch
import rgc_slam_amd.synth as synth
from rgc_slam_amd import registration
N = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
world, tgt = synth.make_world_and_map(1000000, seed=synth.SEED)
poses = synth.make_trajectory(24, seed=synth.SEED)
scans = [synth.make_scan_n(world, poses[i + 1], 30000, seed=synth.SEED + 100 + i)["xyz"] for i in range(23)]
v = registration.odometer_vgicp(0)
def to_dev(xyz):
    a = np.zeros((xyz.shape[0], 4), np.float32); a[:, :3] = xyz
    p = v.device_alloc(a.nbytes); v.upload(p, a); return p
d_tgt = to_dev(tgt); d_s = [to_dev(s) for s in scans]
order = list(range(23)) + list(range(21, 0, -1))     # 0..22..1, repeated: neighbours in the order are neighbours in space
free0 = torch.cuda.mem_get_info()[0]
free200 = None
gc.collect(); gc.freeze()   # CPython's full collections over torch's object graph are 40 ms pauses of this harness (scripts/exp_stall.py)
g = poses[0].astype(np.float32)
per = np.empty(N)
ref = {}
for f in range(N):
    i = order[f % len(order)]
    t0 = time.perf_counter()
    v.setInputTargetDevice(d_tgt, len(tgt), 16); v.setInputSourceDevice(d_s[i], 30000, 16)
    v.align(poses[i].astype(np.float32), want_output=False, want_fitness=True)
    T = v.getFinalTransformation()
    per[f] = 1e3 * (time.perf_counter() - t0)
    if f == 199: v.synchronize(); free200 = torch.cuda.mem_get_info()[0]
    if i in ref: assert np.array_equal(ref[i], T), f"frame {f}: scan {i} gave a different pose than the first time"   # same inputs, same guess
    else: ref[i] = T
v.synchronize()
free1 = torch.cuda.mem_get_info()[0]
q = np.percentile(per[200:], [50, 90, 99, 100])
print(json.dumps({"frames": N, "ms_median": round(float(q[0]), 4), "ms_p90": round(float(q[1]), 4), "ms_p99": round(float(q[2]), 4), "ms_max": round(float(q[3]), 3),
       "first_500_median": round(float(np.median(per[200:700])), 4), "last_500_median": round(float(np.median(per[-500:])), 4),
       "frames_over_1ms": int((per[200:] > 1.0).sum()), "working_set_MiB": round((free0 - free200) / 2**20, 1),
       "steady_state_growth_MiB_frames_200_to_end": round((free200 - free1) / 2**20, 2), "identical_results": True}))
