"""Front-end (A1-A8) timing on the MI355X against the CPU oracle, one synthetic VLP-16 sweep (~28 k points)."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401
import rgc_slam_amd.synth as synth
from rgc_slam_amd import frontend
from oracle import oracle
w = synth.make_world(seed=synth.SEED)
sc = synth.make_scan(w, np.eye(4), n_az=1800, seed=synth.SEED + 3)
xyzi = np.concatenate([sc["xyz"], sc["intensity"][:, None]], axis=1).astype(np.float32)
fe = frontend.ScanRegistration(device=0)
for _ in range(3):
    out = fe.laserCloudHandler(xyzi, diagnostics=False)
t0 = time.perf_counter(); reps = 50
for _ in range(reps):
    out = fe.laserCloudHandler(xyzi, diagnostics=False)
t_gpu = (time.perf_counter() - t0) / reps
t0 = time.perf_counter()
for _ in range(5):
    ref = oracle.frontend(xyzi)
t_cpu = (time.perf_counter() - t0) / 5
print(json.dumps({"workload": f"front-end A1-A8, one VLP-16 sweep of {len(xyzi)} points", "gpu_ms": round(1e3 * t_gpu, 3), "cpu_oracle_ms_1_thread": round(1e3 * t_cpu, 3),
                  "n_sharp": int(len(out["sharp"])), "n_flat": int(len(out["flat"]))}))
