"""What the general covariance route costs: c1-sized clouds (30 k-point scan, 100 k-point map), every RegularizationMethod /
VoxelAccumulationMode against the tuned route, ms per full registration (set both clouds + align), median of 5.
    python scripts/bench_general.py > profiles/r06_general_route.json"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rgc_slam_amd.synth as synth
from rgc_slam_amd import registration
F = registration.FastVGICP
world, tgt = synth.make_world_and_map(100000, seed=synth.SEED)
T_true = synth.se3(synth.rot_zyx(0.02, 0.002, -0.001), [0.12, 0.02, 0.001])
src = synth.make_scan_n(world, T_true, 30000, seed=synth.SEED)["xyz"]
I4 = np.eye(4, dtype=np.float32)
names_r = {F.REG_NONE: "NONE", F.REG_MIN_EIG: "MIN_EIG", F.REG_NORMALIZED_MIN_EIG: "NORMALIZED_MIN_EIG", F.REG_PLANE: "PLANE", F.REG_FROBENIUS: "FROBENIUS"}
names_v = {F.VOXEL_ADDITIVE: "ADDITIVE", F.VOXEL_MULTIPLICATIVE: "MULTIPLICATIVE"}
rows = []
for reg in (F.REG_PLANE, F.REG_MIN_EIG, F.REG_NORMALIZED_MIN_EIG, F.REG_FROBENIUS, F.REG_NONE):
    for vox in (F.VOXEL_ADDITIVE, F.VOXEL_MULTIPLICATIVE):
        if reg == F.REG_NONE and vox == F.VOXEL_MULTIPLICATIVE:
            continue
        v = registration.odometer_vgicp(0)
        v.setRegularizationMethod(reg); v.setVoxelAccumulationMode(vox)
        per = []
        for r in range(6):
            v.synchronize(); t0 = time.perf_counter()
            v.setInputTarget(tgt); v.setInputSource(src); v.align(I4, want_output=False)
            per.append(time.perf_counter() - t0)
        T = v.getFinalTransformation()
        rows.append({"regularization": names_r[reg], "voxel_mode": names_v[vox], "route": "tuned" if (reg == F.REG_PLANE and vox == F.VOXEL_ADDITIVE) else "general",
                     "ms_per_registration": round(1e3 * float(np.median(per[1:])), 3), "iterations": int(v.nr_iterations),
                     "max_abs_translation_error_vs_truth_m": float(np.abs(T[:3, 3] - T_true[:3, 3]).max())})
        v.close()
hc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), ".head_commit")
print(json.dumps({"what": "30 k-point scan vs 100 k-point map (host clouds in, pose out), set both clouds + align, median of 5", "commit": open(hc).read().strip() if os.path.exists(hc) else None, "rows": rows}, indent=1))
