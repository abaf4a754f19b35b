"""One trial of tests/fuzz/fuzz_oracle.py in detail: python tests/fuzz/fuzz_oracle_repro.py <trial> <seed> <nmax>"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import rgc_slam_amd.synth as synth
from rgc_slam_amd import registration as reg
import oracle as orc
trial, seed0, nmax = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
src_txt = open(os.path.join(ROOT, "tests", "fuzz", "fuzz_oracle.py")).read()
ns_ = {"np": np, "synth": synth}
exec("def problem" + src_txt.split("def problem")[1].split("KINDS = ")[0], ns_)
KINDS = ["synth", "synth", "synth", "uniform", "sheets", "clump"]
rng = np.random.default_rng(seed0 * 7919 + trial)
kind = KINDS[int(rng.integers(0, len(KINDS)))]
n = int(np.exp(rng.uniform(np.log(3000), np.log(nmax))))
res = float(rng.choice([0.5, 1.0, 1.0, 2.0])); k = int(rng.choice([20, 20, 20, 10, 25]))
method, mode = 3, 0
if rng.random() < 0.25: method, mode = int(rng.integers(0, 5)), int(rng.integers(0, 3))
tgt, src, d = ns_["problem"](kind, n, rng)
guess = np.eye(4) if rng.random() < 0.5 else synth.se3(synth.rot_zyx(*(rng.normal(0, 0.01, 3))), rng.normal(0, 0.05, 3))
print(kind, len(tgt), len(src), res, k, method, mode)
v = reg.odometer_vgicp(0); v.setResolution(res); v.setCorrespondenceRandomness(k); v.setRegularizationMethod(method); v.setVoxelAccumulationMode(mode)
o = orc.Registration(voxel_res=res, max_iterations=25, translation_eps=1e-6, num_threads=14, k_correspondences=k, regularization=method, voxel_mode=mode)
v.setInputTarget(tgt); v.setInputSource(src); o.set_target(tgt); o.set_source(src); o.prepare()
v.align(guess.astype(np.float32), want_output=False, want_fitness=True)
To = o.align(guess.astype(np.float32)); T = v.getFinalTransformation()
print("hip converged", v.hasConverged(), v.nr_iterations, "oracle", o.converged, o.iterations, "lm_failed", v.lm_failed, o.lm_failed)
print("dt", np.abs(T[:3, 3] - To[:3, 3]).max(), "true motion", d[:3, 3], "hip", T[:3, 3], "oracle", To[:3, 3])
for nm, X in (("hip", T), ("oracle", To)):
    print(" cost at", nm, "pose: oracle says", o.linearize(X.astype(np.float64), want_H=False)[0], " library says", v.compute_error(X.astype(np.float64)), "corr", o.num_correspondences)
for t in o.trace[-6:]:
    print("  oracle trace", t["outer"], t["inner"], t["n_corr"], t["y0"], t["yi"], t["rho"], t["accepted"])
