"""The host's turn-around inside a dependent sequence, measured on the device (a -DRGC_LAB_TURN build named by RGC_HIP_LIB): 100 MHz wall-clock
ticks from the deciding LM launch's post of the final pose to the first wave of the next map's counting pass.
    RGC_EXTRA_FLAGS=-DRGC_LAB_TURN RGC_LIB_OUT=/tmp/librgc_turn.so python3 rgc-slam_amd/build.py; RGC_HIP_LIB=/tmp/librgc_turn.so python3 scripts/lab_turn.py"""
import os, sys, time, json, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
import rgc_slam_amd.synth as synth
from rgc_slam_amd import registration, _lib
K, W = 40, 4
world, tgt = synth.make_world_and_map(1000000, seed=synth.SEED)
poses = synth.make_trajectory(K + W + 1, seed=synth.SEED)
scans = [synth.make_scan_n(world, poses[i + 1], 30000, seed=synth.SEED + 100 + i)["xyz"] for i in range(K + W)]
pv = registration.PipelinedVGICP(0, depth=2)
v = pv.v[0]
for w in pv.v:
    w.setNeighbourReuse(0)
def to_dev(xyz):
    a = np.zeros((xyz.shape[0], 4), np.float32); a[:, :3] = xyz
    p = v.device_alloc(a.nbytes); v.upload(p, a); return p
d_map, d_scans = to_dev(tgt), [to_dev(s) for s in scans]
seq = bench.DependentSequence(pv.v, d_map, len(tgt), d_scans, [len(s) for s in scans])
I4 = np.eye(4, dtype=np.float32); Tw0 = np.asarray(poses[0], np.float64)
for w in pv.v:
    seq.v = [w]; seq.run(0, 1, Tw0, I4, False)
seq.v = pv.v
m, wd, _ = seq.run(0, W, Tw0, I4, True)
L = _lib.load()
L.rgc_lab_turn.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
out8 = (C.c_ulonglong * 8)()
out = {}
for name, overlap in (("two_contexts", True), ("one_frame_at_a_time", False)):
    for r in range(3):
        L.rgc_lab_turn(None, out8)
        pv.synchronize(); t0 = time.perf_counter()
        seq.run(W, K, wd[-1], m[-1], overlap)
        pv.synchronize(); dt = time.perf_counter() - t0
        L.rgc_lab_turn(None, out8)
        t = list(out8)
    out[name] = {"scans_per_s": round(K / dt, 1), "frame_us": round(1e6 * dt / K, 1), "turnarounds": t[1], "post_to_next_count_us_mean": round(t[0] / max(t[1], 1) / 100, 2),
                 "post_to_next_count_us_max": t[2] / 100, "deciding_launch_entry_to_post_us_mean": round(t[5] / max(t[1], 1) / 100, 2)}
print(json.dumps(out))
