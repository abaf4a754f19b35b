"""Where is the one-off ~40 ms stall of a fresh process (bench.py used to hide it behind 96 untimed frames)?  Per-frame wall time of
the first 300 frames of the c-main loop; frames slower than 3x the median are printed with their index.
    python scripts/exp_stall.py [frames]         env: RGC_PREWARM=<n empty launches per stream at rgc_create>"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rgc_slam_amd.synth as synth
from rgc_slam_amd import registration
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 300
nt = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
pause_at = int(sys.argv[3]) if len(sys.argv) > 3 else -1
world, tgt = synth.make_world_and_map(nt, seed=synth.SEED)
poses = synth.make_trajectory(8, seed=synth.SEED)
scans = [synth.make_scan_n(world, poses[i + 1], 30000, seed=synth.SEED + 100 + i)["xyz"] for i in range(4)]
t0 = time.perf_counter()
v = registration.odometer_vgicp(0)
t_create = time.perf_counter() - t0
def to_dev(xyz):
    a = np.zeros((xyz.shape[0], 4), np.float32); a[:, :3] = xyz
    p = v.device_alloc(a.nbytes); v.upload(p, a); return p
d_tgt = to_dev(tgt); d_s = [to_dev(s) for s in scans]
g0 = poses[0].astype(np.float32)
import gc
if os.environ.get("EXP_GC") == "freeze":
    gc.collect(); gc.freeze()
if os.environ.get("EXP_GC") == "off":
    gc.disable()
per = []
parts = []
for j in range(frames):
    if j == pause_at: time.sleep(0.2)
    tf = time.perf_counter()
    v.setInputTargetDevice(d_tgt, len(tgt), 16); ta = time.perf_counter()
    v.setInputSourceDevice(d_s[j % 4], 30000, 16); tb = time.perf_counter()
    v.align(g0, want_output=False, want_fitness=True)
    te = time.perf_counter()
    per.append(1e3 * (te - tf))
    parts.append((round(1e3 * (ta - tf), 2), round(1e3 * (tb - ta), 2), round(1e3 * (te - tb), 2)))
per = np.array(per)
med = float(np.median(per[20:]))
slow = [(int(i), round(float(t), 2)) for i, t in enumerate(per) if t > 3 * med and i > 0]
t_loop = float(per.sum())
print({"gc": os.environ.get("EXP_GC"), "n_target": nt, "pause_at": pause_at, "ms_to_first_slow": round(float(per[:slow[0][0]].sum()), 1) if slow else None, "prewarm": os.environ.get("RGC_PREWARM"), "create_ms": round(1e3 * t_create, 1), "median_ms": round(med, 3), "slow_frames": slow, "slow_parts_ms(set_target,set_source,align)": [parts[i] for i, _ in slow], "first_frames_ms": [round(float(t), 2) for t in per[:4]]})
