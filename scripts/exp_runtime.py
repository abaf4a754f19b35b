"""Which HIP runtime drives the library matters: PyTorch 2.10+rocm7.0 bundles its own libamdhip64 / libhsa-runtime64 (ROCm 7.0) and a process
that imports torch first binds librgc_hip.so to THAT copy (same soname); a process that never imports torch -- any C++ caller -- gets the
image's /opt/rocm 7.2 runtime.  The same dependent c-main sequence through the C++ frame loop (librgc_seq.so) either way.
    python scripts/exp_runtime.py torch|system [frames]"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mode = sys.argv[1] if len(sys.argv) > 1 else "system"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 40
if mode == "torch":
    import torch  # noqa: F401
import numpy as np
import bench
import rgc_slam_amd.synth as synth
from rgc_slam_amd import registration
W = 4
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
m, wd, _ = seq.run_cpp(0, W, Tw0, I4, True)
out = {"mode": mode, "runtime": [l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l and "r-xp" in l]}
for name, overlap in (("two_contexts", True), ("one_frame_at_a_time", False)):
    per = []
    for r in range(8):
        pv.synchronize(); t0 = time.perf_counter()
        seq.run_cpp(W, K, wd[-1], m[-1], overlap)
        pv.synchronize(); per.append(time.perf_counter() - t0)
    out[name] = round(K / float(np.median(per[2:])), 1)
out["pose_checksum"] = float(np.asarray(wd).sum())
print(json.dumps(out))
