"""f4 measurement: loop-closure ICP (rgc_icp_align) on the MI355X against the CPU oracle on the same clouds."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401
import rgc_slam_amd.synth as synth
from rgc_slam_amd import loop_closure
from oracle import oracle
nt = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
ns = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
world, tgt = synth.make_world_and_map(nt)
T_true = synth.se3(synth.rot_zyx(0.05, 0.01, -0.008), [0.4, -0.25, 0.05])
src = synth.make_scan_n(world, np.eye(4), ns)["xyz"]
Ti = np.linalg.inv(T_true)
src = (src @ Ti[:3, :3].T + Ti[:3, 3]).astype(np.float32)
icp = loop_closure.IterativeClosestPoint(0)
icp.setMaxCorrespondenceDistance(10.0)
icp.setInputSource(src); icp.setInputTarget(tgt)
for _ in range(2):
    T = icp.align().copy()
t0 = time.perf_counter(); reps = 10
for _ in range(reps):
    T = icp.align().copy()
t_gpu = (time.perf_counter() - t0) / reps
res = {}
for th in (14, 0):
    t0 = time.perf_counter()
    To, ro = oracle.icp_align(src, tgt, max_corr_dist=10.0, threads=th)
    res[th] = time.perf_counter() - t0
print(json.dumps({"workload": f"f4 loop-closure ICP: {ns}-point key frame vs {nt}-point history sub-map, max correspondence distance 10 m",
                  "gpu_ms": round(1e3 * t_gpu, 3), "iterations": icp.nr_iterations, "state": icp.convergence_state,
                  "cpu_oracle_ms_14_threads": round(1e3 * res[14], 1), "cpu_oracle_ms_all_cores": round(1e3 * res[0], 1), "cores": os.cpu_count(),
                  "max_T_diff_vs_oracle": float(np.abs(T - To).max()), "fitness": icp.getFitnessScore(), "oracle_fitness": ro["fitness"]}))
