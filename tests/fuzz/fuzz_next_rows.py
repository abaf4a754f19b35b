"""Randomised campaign of two of SURVEY 8f's "next" rows against the ORACLE: f4, the loop-closure ICP (random maps, drifts up to 1 m / 5 degrees,
source sizes, correspondence distances), and f1, the mapping node's feature registration (random worlds / trajectories / perturbations:
association flags and factors, the two-pass solve's iteration counts, costs and poses).
    python tests/fuzz/fuzz_next_rows.py [icp trials] [mapreg trials] [seed]"""
import sys, os, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import rgc_slam_amd.synth as synth
from rgc_slam_amd import loop_closure, mapping
from oracle import oracle
import mapreg_data as md

n_icp = int(sys.argv[1]) if len(sys.argv) > 1 else 40
n_mr = int(sys.argv[2]) if len(sys.argv) > 2 else 6
seed0 = int(sys.argv[3]) if len(sys.argv) > 3 else 1
rep = {"icp_trials": 0, "mapreg_trials": 0, "failures": [], "max": {"icp_T": 0.0, "icp_fitness_rel": 0.0, "mapreg_x": 0.0, "mapreg_cost_rel": 0.0}}
t0 = time.time()
icp = loop_closure.IterativeClosestPoint(0)
for trial in range(n_icp):
    rng = np.random.default_rng(seed0 * 49979687 + trial)
    nt, ns = int(np.exp(rng.uniform(np.log(5000), np.log(120000)))), int(rng.integers(500, 15000))
    mcd = float(rng.choice([2.0, 5.0, 10.0]))
    tag = {"stage": "icp", "trial": trial, "n_target": nt, "n_source": ns, "max_corr_dist": mcd}
    try:
        world, tgt = synth.make_world_and_map(nt, seed=int(rng.integers(1, 1 << 30)))
        T_true = synth.se3(synth.rot_zyx(*(rng.normal(0, 0.03, 3) * np.array([1, 0.3, 0.3]))), rng.normal(0, 0.3, 3) * np.array([1, 1, 0.2]))
        src = synth.make_scan_n(world, np.eye(4), ns, seed=int(rng.integers(1, 1 << 30)))["xyz"]
        Ti = np.linalg.inv(T_true)
        src = (src @ Ti[:3, :3].T + Ti[:3, 3]).astype(np.float32)
        icp.setMaxCorrespondenceDistance(mcd); icp.setMaximumIterations(100); icp.setTransformationEpsilon(1e-6); icp.setEuclideanFitnessEpsilon(1e-6)
        icp.setInputSource(src); icp.setInputTarget(tgt.astype(np.float32))
        T = icp.align().copy()
        To, ro = oracle.icp_align(src, tgt.astype(np.float32), max_corr_dist=mcd)
        dT = float(np.abs(T - To).max())
        rep["max"]["icp_T"] = max(rep["max"]["icp_T"], dT)
        same_it = icp.nr_iterations == ro["iterations"]
        # (an iteration count one apart: the stopping test sits on a step of ~1e-6; the two ends then differ by about that step)
        if not (bool(icp.hasConverged()) == bool(ro["converged"]) and abs(icp.nr_iterations - ro["iterations"]) <= 1 and dT < (1e-5 if same_it else 1e-4)):
            rep["failures"].append(dict(tag, error="icp", dT=dT, it=[int(icp.nr_iterations), int(ro["iterations"])], converged=[bool(icp.hasConverged()), bool(ro["converged"])]))
        elif ro["converged"] and ro["fitness"] > 0:
            fr = abs(icp.getFitnessScore() - ro["fitness"]) / ro["fitness"]
            rep["max"]["icp_fitness_rel"] = max(rep["max"]["icp_fitness_rel"], fr)
            if not fr <= (1e-5 if same_it else 1e-3):
                rep["failures"].append(dict(tag, error="icp fitness", rel=fr))
    except Exception as e:
        rep["failures"].append(dict(tag, error="exception: %r" % (e,)))
    rep["icp_trials"] += 1
icp.close()
for trial in range(n_mr):
    rng = np.random.default_rng(seed0 * 67867967 + trial)
    tag = {"stage": "mapreg", "trial": trial}
    try:
        c = md.make_case(synth, oracle.frontend, n_map_frames=int(rng.integers(5, 14)), seed=int(rng.integers(1, 1 << 20)), n_az=int(rng.choice([900, 1200, 1800])),
                         voxelgrid=oracle.voxelgrid_filter)
        x0 = md.poses14(md.perturb(c["T_cur"], rng, ang=float(rng.uniform(0.002, 0.02)), trans=float(rng.uniform(0.01, 0.1))), md.perturb(c["T_last"], rng))
        r = mapping.MapFeatureRegistration(0)
        r.setInputMaps(c["corner_map"], c["surf_map"])
        for kind in ("edge", "plane"):
            feat, mp = (c["corner_cur"], c["corner_map"]) if kind == "edge" else (c["surf_cur"], c["surf_map"])
            a = r.associate(feat, x0[0:4], x0[4:7], kind)
            b = oracle.mapreg_associate(feat, x0[0:4], x0[4:7], mp, kind)
            if (a["valid"] != b["valid"]).sum() > 3:
                rep["failures"].append(dict(tag, error="association flags (%s)" % kind, differ=int((a["valid"] != b["valid"]).sum()), n=len(feat)))
            both = a["valid"] & b["valid"]
            if both.any():
                if kind == "edge":
                    d = np.minimum(np.abs(a["a"][both] - b["a"][both]).max(axis=1), np.abs(a["a"][both] - b["b"][both]).max(axis=1)).max()
                else:
                    d = max(np.abs(a["n"][both] - b["n"][both]).max(), np.abs(a["d"][both] - b["d"][both]).max() * 0.1)
                if not d < 1e-8:
                    rep["failures"].append(dict(tag, error="association factors (%s)" % kind, err=float(d)))
        qc, tc, ql, tl, rp = r.optimize(c["corner_cur"], c["surf_cur"], c["corner_last"], c["surf_last"], x0[0:4], x0[4:7], x0[7:11], x0[11:14])
        xo, rc, tr = oracle.mapreg_optimize(c["corner_cur"], c["surf_cur"], c["corner_last"], c["surf_last"], c["corner_map"], c["surf_map"], x0)
        x = np.concatenate([qc, tc, ql, tl])
        dx = float(np.abs(x - xo).max())
        rep["max"]["mapreg_x"] = max(rep["max"]["mapreg_x"], dx)
        its = [(rp[i]["iterations"], tr[i]["iterations"]) for i in range(2)]
        same = all(a_ == b_ for a_, b_ in its)
        if same:
            for i in range(2):
                cr = abs(rp[i]["final_cost"] - tr[i]["final_cost"]) / max(tr[i]["final_cost"], 1e-300)
                rep["max"]["mapreg_cost_rel"] = max(rep["max"]["mapreg_cost_rel"], cr)
        if not (rc == 0 and dx < (1e-6 if same else 1e-3)):
            rep["failures"].append(dict(tag, error="mapreg solve", dx=dx, iterations=its, rc=int(rc)))
        r.close()
    except Exception as e:
        import traceback
        rep["failures"].append(dict(tag, error="exception: %r" % (e,), where=traceback.format_exc()[-400:]))
    rep["mapreg_trials"] += 1
rep["wall_s"] = round(time.time() - t0, 1)
print(json.dumps(rep))
