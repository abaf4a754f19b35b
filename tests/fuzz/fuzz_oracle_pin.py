"""Randomised pinning of the C ORACLE itself (oracle/rgc_oracle*.c) against the literal numpy / scipy restatements it was first checked with on
fixed fixtures (oracle/py_oracle.py: cKDTree kNN, numpy SVD, 4x4 homogeneous matrices as the reference writes them): random small clouds --
the synthetic world, uniform noise, sheets -- k, leaf sizes, every regularisation and accumulation mode, DIRECT1 / DIRECT7 / DIRECT27:
neighbour sets, covariances, voxel tables, a linearisation, the LM trajectory's end and the fitness; the leaf filter; de-skew and re-framing against scipy.  No GPU.
    python tests/fuzz/fuzz_oracle_pin.py [trials] [seed]"""
import sys, os, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import rgc_slam_amd.synth as synth
from oracle import oracle as orc, py_oracle as po

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 30
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
REG = ["NONE", "MIN_EIG", "NORMALIZED_MIN_EIG", "PLANE", "FROBENIUS"]
MODE = ["ADDITIVE", "ADDITIVE_WEIGHTED", "MULTIPLICATIVE"]
METH = ["DIRECT27", "DIRECT7", "DIRECT1"]     # enum order of gicp_settings.hpp:8
rep = {"trials": 0, "failures": [], "max": {"cov": 0.0, "vox_mean": 0.0, "vox_cov": 0.0, "H_rel": 0.0, "b_rel": 0.0, "cost_rel": 0.0, "pose": 0.0, "fitness_rel": 0.0, "leaf_filter": 0.0, "deskew": 0.0, "transform": 0.0}}


def note(k, v):
    rep["max"][k] = max(rep["max"][k], float(v))


t0 = time.time()
for trial in range(trials):
    rng = np.random.default_rng(seed0 * 32416190071 % (1 << 32) + trial)
    kind = str(rng.choice(["synth", "synth", "uniform", "sheets"]))
    n = int(rng.integers(300, 2500))
    k = int(rng.choice([20, 20, 10, 25]))
    res = float(rng.choice([0.5, 1.0, 2.0]))
    method, mode = (3, 0) if rng.random() < 0.6 else (int(rng.integers(0, 5)), int(rng.integers(0, 3)))
    nmeth = int(rng.integers(0, 3))
    tag = {"trial": trial, "kind": kind, "n": n, "k": k, "res": res, "method": REG[method], "mode": MODE[mode], "neighbours": METH[nmeth]}
    try:
        if kind == "synth":
            world, tgt = synth.make_world_and_map(n, seed=int(rng.integers(1, 1 << 30)))
        elif kind == "uniform":
            tgt = rng.uniform(-6, 6, (n, 3))
        else:
            tgt = np.vstack([np.c_[rng.uniform(-8, 8, (n // 2, 2)), np.zeros(n // 2)], np.c_[rng.uniform(-8, 8, n - n // 2), np.full(n - n // 2, 3.0), rng.uniform(0, 5, n - n // 2)]]) + rng.normal(0, 5e-3, (n, 3))
        tgt = np.ascontiguousarray(tgt, np.float32)
        d = synth.se3(synth.rot_zyx(*(rng.normal(0, 0.01, 3))), rng.normal(0, 0.06, 3))
        sel = rng.choice(n, max(k + 5, n // 3), replace=False)
        src = np.ascontiguousarray(((tgt[sel].astype(np.float64) - d[:3, 3]) @ d[:3, :3]).astype(np.float32) + rng.normal(0, 0.01, (len(sel), 3)).astype(np.float32))
        # ---- neighbour sets and covariances ----
        pc, pidx = po.covariances(tgt, k, REG[method])
        oc = orc.covariances_m(tgt, method, k=k, threads=4)
        oidx, _ = orc.knn(tgt, k=k, threads=4)
        same_sets = np.array_equal(np.sort(np.asarray(oidx), axis=1), np.sort(pidx, axis=1))
        if not same_sets:
            bad = int(np.any(np.sort(np.asarray(oidx), axis=1) != np.sort(pidx, axis=1), axis=1).sum())
            # (a tie on the k-th distance in fp64 vs fp32 arithmetic may pick another point: cKDTree works in double)
            if bad > max(2, n // 200):
                rep["failures"].append(dict(tag, error="neighbour sets differ", queries=bad))
        e = np.abs(np.asarray(oc).reshape(n, 3, 3) - pc[:, :3, :3]).reshape(n, -1).max(axis=1)
        ok_rows = np.all(np.sort(np.asarray(oidx), axis=1) == np.sort(pidx, axis=1), axis=1)
        ce = float(e[ok_rows].max()) if ok_rows.any() else 0.0
        note("cov", ce)
        if not ce <= (1e-7 if REG[method] == "FROBENIUS" else 1e-9):
            rep["failures"].append(dict(tag, error="covariances", err=ce))
        # ---- the whole registration object: voxel table, linearisation, solve, fitness ----
        o = orc.Registration(voxel_res=res, max_iterations=25, translation_eps=1e-6, num_threads=4, k_correspondences=k, regularization=method, voxel_mode=mode, neighbor_method=nmeth)
        o.set_target(tgt); o.set_source(src); o.prepare()
        p = po.VGICP(res=res, max_iterations=25, method=METH[nmeth], regularization=REG[method], voxel_mode=MODE[mode])
        if k != 20:      # (the restatement's class takes the reference's k = 20: compare the registration object only there)
            rep["trials"] += 1
            continue
        p.set_target(tgt); p.set_source(src)
        # ---- the voxel table: same cells, same counts, means and covariances ----
        ov = o.voxelmap()
        okeys = {tuple(int(x) for x in c): j for j, c in enumerate(ov["coords"])}
        if set(okeys) != set(p.vox) or any(int(ov["num"][okeys[c]]) != v["n"] for c, v in p.vox.items()):
            rep["failures"].append(dict(tag, error="voxel table: cells or counts differ", c=len(okeys), py=len(p.vox)))
        elif ok_rows.all():
            vm = max(float(np.abs(ov["mean"][okeys[c]] - v["mean"][:3]).max()) for c, v in p.vox.items())
            vc = max(float(np.abs(ov["cov"][okeys[c]] - v["cov"][:3, :3]).max() / max(1.0, float(np.abs(v["cov"][:3, :3]).max()))) for c, v in p.vox.items())
            note("vox_mean", vm); note("vox_cov", vc)
            if not (vm <= 1e-9 and vc <= (1e-6 if REG[method] == "FROBENIUS" or MODE[mode] == "MULTIPLICATIVE" else 1e-9)):
                rep["failures"].append(dict(tag, error="voxel table", mean=vm, cov=vc))
        g = np.eye(4)
        ocost, oH, ob = o.linearize(g)
        pcost, pH, pb = p.linearize(g)
        if abs(pcost) > 0 and np.abs(pH).max() > 0:
            hr, br, cr = np.abs(oH - pH).max() / np.abs(pH).max(), np.abs(ob - pb).max() / max(np.abs(pb).max(), 1e-300), abs(ocost - pcost) / abs(pcost)
            note("H_rel", hr); note("b_rel", br); note("cost_rel", cr)
            if ok_rows.all() and not (hr <= 1e-7 and br <= 1e-6 and cr <= 1e-7):
                rep["failures"].append(dict(tag, error="linearisation", H=float(hr), b=float(br), cost=float(cr)))
        To = o.align(g.astype(np.float32))
        Tp = p.align(g)
        if np.all(np.isfinite(To)) and np.all(np.isfinite(Tp)) and ok_rows.all():
            dT = float(np.abs(np.asarray(To, float) - np.asarray(Tp, float)).max())
            note("pose", dT)
            if o.converged and o.iterations < 25 and not dT <= 1e-4:
                rep["failures"].append(dict(tag, error="pose", dT=dT, iterations=int(o.iterations)))
            if dT <= 1e-6:
                fo, fp_ = o.fitness(), p.fitness()
                fr = abs(fo - fp_) / max(abs(fp_), 1e-300)
                note("fitness_rel", fr)
                if not fr <= 1e-4:
                    rep["failures"].append(dict(tag, error="fitness", c=float(fo), py=float(fp_)))
        # ---- B2 adjustDistortion (RGC_odometer.cpp:1441-1481) and B9 transformPointCloud (:1495-1514) against scipy's Rotation / Slerp ----
        from scipy.spatial.transform import Rotation as Rot, Slerp
        m_pts = min(n, 600)
        pts = tgt[:m_pts]
        ring, rel = rng.integers(0, 16, m_pts), rng.uniform(0, 1, m_pts)
        inten = (ring + 0.1 * rel).astype(np.float32)
        cloud = np.ascontiguousarray(np.c_[pts, inten], np.float32)
        qd = Rot.from_rotvec(rng.normal(0, 0.03, 3))
        td = rng.normal(0, 0.1, 3)
        out = orc.deskew(cloud, qd.as_quat(), td)
        sfrac = 1.0 - (inten.astype(np.float64) - np.floor(inten.astype(np.float64))) / 0.1
        qs = Slerp([0.0, 1.0], Rot.from_quat(np.stack([[0, 0, 0, 1.0], qd.inv().as_quat()])))(np.clip(sfrac, 0.0, 1.0))
        e_ds = float(np.abs(out[:, :3] - qs.apply(pts.astype(np.float64) - sfrac[:, None] * td).astype(np.float32)).max())
        tw = orc.transform_cloud(cloud, qd.as_quat(), td)
        e_tf = float(np.abs(tw[:, :3] - (qd.apply(pts.astype(np.float64)) + td).astype(np.float32)).max())
        note("deskew", e_ds); note("transform", e_tf)
        if not (e_ds < 4e-6 and e_tf < 4e-6 and np.array_equal(out[:, 3], inten) and np.array_equal(tw[:, 3], inten)):     # (fp32 results of points up to ~20 m: an ulp is 2e-6)
            rep["failures"].append(dict(tag, error="de-skew / transform", deskew=e_ds, transform=e_tf))
        # ---- the leaf filter ----
        xyzi = np.c_[tgt, rng.uniform(0, 16, n)].astype(np.float32)
        leaf = float(rng.choice([0.2, 0.3, 0.5, 1.0]))
        a, b = orc.voxelgrid_filter(xyzi, leaf), po.voxelgrid_filter(xyzi, leaf)
        if a.shape != b.shape:
            rep["failures"].append(dict(tag, error="leaf filter: sizes", c=list(a.shape), py=list(b.shape)))
        else:
            e = float(np.abs(a - b).max()) if len(a) else 0.0
            note("leaf_filter", e)
            if not e <= 1e-5:
                rep["failures"].append(dict(tag, error="leaf filter", err=e))
    except Exception as e:
        import traceback
        rep["failures"].append(dict(tag, error="exception: %r" % (e,), where=traceback.format_exc()[-500:]))
    rep["trials"] += 1
    if len(rep["failures"]) > 12:
        break
rep["wall_s"] = round(time.time() - t0, 1)
print(json.dumps(rep))
