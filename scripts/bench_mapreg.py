"""f1 measurement: the mapping node's feature registration (rgc_mapreg_optimize) on the MI355X against the CPU oracle on the
same inputs (front-end features of synthetic VLP-16 scans against feature maps accumulated from earlier frames)."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch  # noqa: F401  (first: its HIP runtime is the one the library must bind to)
import rgc_slam_amd.synth as synth
from rgc_slam_amd import mapping
from oracle import oracle
import mapreg_data as md

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 40
case = md.make_case(synth, oracle.frontend, n_map_frames=frames, n_az=1800, voxelgrid=oracle.voxelgrid_filter)
rng = np.random.default_rng(3)
x0 = md.poses14(md.perturb(case["T_cur"], rng), md.perturb(case["T_last"], rng))
reg = mapping.MapFeatureRegistration(0)
def gpu_once():
    reg.setInputMaps(case["corner_map"], case["surf_map"])
    return reg.optimize(case["corner_cur"], case["surf_cur"], case["corner_last"], case["surf_last"], x0[0:4], x0[4:7], x0[7:11], x0[11:14])
for _ in range(3):
    out = gpu_once()
t0 = time.perf_counter(); reps = 20
for _ in range(reps):
    out = gpu_once()
t_gpu = (time.perf_counter() - t0) / reps
res = {}
for th in (14, 0):
    t0 = time.perf_counter()
    xo, rc, tr = oracle.mapreg_optimize(case["corner_cur"], case["surf_cur"], case["corner_last"], case["surf_last"], case["corner_map"], case["surf_map"], x0, threads=th)
    res[th] = time.perf_counter() - t0
x = np.concatenate(out[:4])
print(json.dumps({"workload": f"f1 feature registration: {len(case['corner_cur'])}+{len(case['surf_cur'])} / {len(case['corner_last'])}+{len(case['surf_last'])} features vs "
                              f"{len(case['corner_map'])} corner / {len(case['surf_map'])} surf map points (maps re-uploaded and re-gridded per frame)",
                  "gpu_ms_per_frame": round(1e3 * t_gpu, 3), "cpu_oracle_ms_14_threads": round(1e3 * res[14], 2), "cpu_oracle_ms_all_cores": round(1e3 * res[0], 2),
                  "cores": os.cpu_count(), "max_pose_diff_vs_oracle": float(np.abs(x - xo).max()), "report": out[4]}))
