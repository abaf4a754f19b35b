"""Steady state of the two-context pipeline over a long sequence: N c-main frames (23 scans cycled forward and backward, each with its
own fixed guess so that a repeated scan must give the bit-identical pose), throughput per block of 500 frames, device memory at frame 400
and at the end."""
import sys, os, time, gc, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rgc_slam_amd.synth as synth
from rgc_slam_amd import registration
N = int(sys.argv[1]) if len(sys.argv) > 1 else 6000
world, tgt = synth.make_world_and_map(1000000, seed=synth.SEED)
poses = synth.make_trajectory(24, seed=synth.SEED)
scans = [synth.make_scan_n(world, poses[i + 1], 30000, seed=synth.SEED + 100 + i)["xyz"] for i in range(23)]
pv = registration.PipelinedVGICP(0, depth=2)
v = pv.v[0]
def to_dev(xyz):
    a = np.zeros((xyz.shape[0], 4), np.float32); a[:, :3] = xyz
    p = v.device_alloc(a.nbytes); v.upload(p, a); return p
d_tgt = to_dev(tgt); d_s = [to_dev(s) for s in scans]
order = list(range(23)) + list(range(21, 0, -1))
def setc(f, w):
    w.setInputTargetDevice(d_tgt, len(tgt), 16); w.setInputSourceDevice(d_s[order[f % len(order)]], 30000, 16)
guess_of = lambda f: poses[order[f % len(order)]].astype(np.float32)
gc.collect(); gc.freeze()
ref, marks, mem = {}, [], {}
free0 = torch.cuda.mem_get_info()[0]
def done(f, w):
    i = order[f % len(order)]
    T = w.getFinalTransformation()
    if i in ref: assert np.array_equal(ref[i], T), f"frame {f}: scan {i} gave a different pose than the first time"
    else: ref[i] = T
    if f % 500 == 499: marks.append(time.perf_counter())
    if f == 399: mem["a"] = torch.cuda.mem_get_info()[0]
t0 = time.perf_counter()
pv.run(N, setc, guess_of(0), want_fitness=True, next_guess=lambda f, T: guess_of(f + 1), on_result=done)
pv.synchronize()
mem["b"] = torch.cuda.mem_get_info()[0]
rates = [round(500 / (b - a), 1) for a, b in zip(marks[:-1], marks[1:])]
print(json.dumps({"frames": N, "scans_per_s_overall": round(N / (time.perf_counter() - t0), 1), "scans_per_s_per_500_frames_min_max": [min(rates), max(rates)],
                  "working_set_MiB": round((free0 - mem["a"]) / 2**20, 1), "growth_MiB_frames_400_to_end": round((mem["a"] - mem["b"]) / 2**20, 2),
                  "identical_results": True}))
