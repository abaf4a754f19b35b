"""Randomised differential campaign over the routes one target can take through the library: rgc_set_target_reframed with nothing kept /
seeds / neighbour lists / a lazy target, and rgc_transform_cloud + rgc_set_target_device -- on clouds the fixed tests do not hold (lattices
whose distances tie exactly, repeated points, sheets and lines, a dense clump in a sparse field, uniform noise), random sizes, leaf sizes, k,
poses (any yaw, +-60 m) and edits of the buffer between frames.  Per frame: every covariance and the voxel table bit for bit across the routes,
the oracle's covariances (1e-9) on the smaller clouds, then one solve per route from the same guess: final transformation, iteration count and
fitness bit for bit (where the routes' contexts chose the same grid for the scan).      python tests/fuzz/fuzz_modes.py [trials] [seed] [max_points]"""
import sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import rgc_slam_amd.synth as synth
from rgc_slam_amd import registration
import bench
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "oracle"))
import oracle as orc

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
nmax = int(sys.argv[3]) if len(sys.argv) > 3 else 200000
ORACLE_MAX = 40000


def cloud(kind, n, rng):
    if kind == "synth":
        return synth.make_world_and_map(n, seed=int(rng.integers(1, 1 << 30)))[1].astype(np.float32)
    if kind == "uniform":
        side = (n / rng.uniform(3.0, 30.0)) ** (1.0 / 3.0)
        return rng.uniform(-side / 2, side / 2, (n, 3)).astype(np.float32)
    if kind == "lattice":          # exact ties: every point has 6 neighbours at one distance, 12 at the next, ...
        m = int(round(n ** (1.0 / 3.0))) + 1
        s = rng.choice([0.25, 0.3, 0.5])
        g = np.stack(np.meshgrid(np.arange(m), np.arange(m), np.arange(max(2, m // 4)), indexing="ij"), -1).reshape(-1, 3).astype(np.float32) * np.float32(s)
        g = g[rng.permutation(len(g))[:n]]
        j = rng.random(len(g)) < 0.2
        g[j] += rng.normal(0, 0.01, (int(j.sum()), 3)).astype(np.float32)
        return g
    if kind == "sheets":           # two planes and a pole: degenerate neighbourhoods
        a = np.c_[rng.uniform(-20, 20, (n // 2, 2)), np.zeros(n // 2)]
        b = np.c_[rng.uniform(-20, 20, n // 3), np.full(n // 3, 3.0), rng.uniform(0, 6, n // 3)]
        c = np.c_[np.full(n - n // 2 - n // 3, 1.5), np.full(n - n // 2 - n // 3, -2.0), rng.uniform(0, 8, n - n // 2 - n // 3)]
        p = np.vstack([a, b, c]) + rng.normal(0, 1e-3, (n, 3))
        return p.astype(np.float32)[rng.permutation(n)]
    if kind == "clump":            # a dense clump in a sparse field (crowded rows beside empty blocks)
        a = rng.normal(0, 0.4, (n // 2, 3))
        b = rng.uniform(-40, 40, (n - n // 2, 3)) * np.array([1, 1, 0.1])
        return np.vstack([a, b]).astype(np.float32)[rng.permutation(n)]
    if kind == "repeats":          # exact duplicates: zero distances, ties decided by the original index
        base = synth.make_world_and_map(max(64, n // 3), seed=int(rng.integers(1, 1 << 30)))[1].astype(np.float32)
        return base[rng.integers(0, len(base), n)]
    raise ValueError(kind)


KINDS = ["synth", "synth", "uniform", "lattice", "sheets", "clump", "repeats"]
report = {"trials": 0, "frames": 0, "solves": 0, "oracle_checks": 0, "failures": [], "by_kind": {}, "max_cov_err_vs_oracle": 0.0, "searched_fraction_lists": []}
t_start = time.time()
for trial in range(trials):
    rng = np.random.default_rng(seed0 * 100003 + trial)
    kind = KINDS[int(rng.integers(0, len(KINDS)))]
    n = int(np.exp(rng.uniform(np.log(2000), np.log(nmax))))
    res = float(rng.choice([0.5, 1.0, 1.0, 2.0]))
    k = int(rng.choice([20, 20, 20, 20, 10, 25]))
    pts = cloud(kind, n, rng)
    n = len(pts)
    a = np.zeros((n, 4), np.float32); a[:, :3] = pts
    reach = float(rng.choice([3.0, 20.0, 60.0]))     # how far the poses of a trial's frames lie apart (the lists' certificates follow the box's exponent)
    tag = {"trial": trial, "kind": kind, "n": n, "res": res, "k": k, "reach": reach}
    ctx = {}
    for name, mode in (("none", 0), ("seeds", 1), ("lists", 2), ("lazy", 2), ("device", 0)):
        v = registration.odometer_vgicp(0)
        v.setResolution(res); v.setCorrespondenceRandomness(k); v.setNeighbourReuse(mode)
        if name == "lazy":
            v.setLazyTarget(2)
        ctx[name] = v
    v0 = ctx["none"]
    dm = {nm: v.device_alloc(a.nbytes) for nm, v in ctx.items()}
    db = {nm: v.device_alloc(a.nbytes) for nm, v in ctx.items()}
    for nm, v in ctx.items():
        v.upload(dm[nm], a)
    try:
        Tw = None
        for f in range(5 if reach <= 3.0 else 3):
            if f and rng.random() < 0.4:     # an edit between frames
                j = rng.integers(0, n, max(1, n // 500))
                a[j, :3] += rng.normal(0, 0.05, (len(j), 3)).astype(np.float32)
                for nm, v in ctx.items():
                    v.upload(dm[nm], a)
            if reach <= 3.0 and Tw is not None:   # a vehicle's motion from the previous pose: what the lists are for (their certificates hold while
                Tw = Tw @ synth.se3(synth.rot_zyx(*(rng.normal(0, 0.03, 3) * np.array([1, 0.1, 0.1]))), rng.normal(0, 0.4, 3) * np.array([1, 1, 0.05]))  # the box keeps its exponent)
            else:
                ang = rng.uniform(-np.pi, np.pi, 3) * np.array([1.0, 0.03, 0.03])
                Tw = synth.se3(synth.rot_zyx(*ang), rng.uniform(-reach, reach, 3) * np.array([1, 1, 0.05]))
            q, t = bench.world_to_body(Tw)
            for nm in ("none", "seeds", "lists", "lazy"):
                ctx[nm].setInputTargetReframed(dm[nm], n, 16, q, t, db[nm])
            ctx["device"].transformCloudDevice(dm["device"], n, 16, q, t, db["device"])
            ctx["device"].setInputTargetDevice(db["device"], n, 16)
            report["frames"] += 1
            c0 = v0.getTargetCovariances()
            def voxels_sorted(v):
                x = v.getVoxels()
                o = np.lexsort(x["coords"].T[::-1])   # (the routes' grids need not have the same box: voxel ids may be numbered differently)
                return {kk: x[kk][o] for kk in ("coords", "num", "mean", "cov")}
            x0 = voxels_sorted(v0)
            for nm in ("seeds", "lists", "device"):
                c1 = ctx[nm].getTargetCovariances()
                x1 = voxels_sorted(ctx[nm])
                ok = np.array_equal(c0, c1) and all(np.array_equal(x0[kk], x1[kk]) for kk in ("coords", "num", "mean", "cov"))
                if not ok:
                    report["failures"].append(dict(tag, frame=f, what="covariances / voxels of route '%s' differ from 'none'" % nm,
                                                   points=int(np.any(c0.reshape(n, -1) != c1.reshape(n, -1), axis=1).sum())))
            if os.environ.get("FUZZ_TRACE") and k == 20:
                print(tag, f, {nm: (ctx[nm].stats()["searched_target"], ctx[nm].stats()["deferred_target"], ctx[nm].stats()["target_cells"]) for nm in ("none", "seeds", "lists")}, file=sys.stderr)
            if k == 20 and f >= 1:
                report["searched_fraction_lists"].append(round(ctx["lists"].stats()["searched_target"] / n, 4))
            body = v0.download(db["none"], (n, 4))
            # (repeated points: neighbourhoods of two or three distinct positions, whose smallest eigenvalue is 0 twice over -- ANY normal in that
            # plane is an eigenvector, the reference's SVD returns one, the oracle another, the library a third; a lattice's symmetric
            # neighbourhoods have nearly equal eigenvalues, which amplify the sums' rounding: 1e-6 there)
            if n <= ORACLE_MAX and f == 0 and kind != "repeats":
                oc, _ = orc.covariances(body[:, :3].copy(), k=k, threads=14)
                err = float(np.abs(c0 - oc).max())
                report["oracle_checks"] += 1
                report["max_cov_err_vs_oracle"] = max(report["max_cov_err_vs_oracle"], err)
                if not err <= (1e-6 if kind == "lattice" else 1e-9):
                    report["failures"].append(dict(tag, frame=f, what="covariances differ from the oracle's", err=err,
                                                   points=int((np.abs(c0 - oc).reshape(n, -1).max(1) > 1e-9).sum())))
            # one solve per route: a scan = a subset of the body-frame cloud, moved a little, with noise
            ns = int(min(n, max(k + 1, rng.integers(300, 20000))))
            sel = rng.choice(n, ns, replace=False)
            d = synth.se3(synth.rot_zyx(*(rng.normal(0, 0.02, 3))), rng.normal(0, 0.15, 3))
            src = ((body[sel, :3].astype(np.float64) - d[:3, 3]) @ d[:3, :3]).astype(np.float32) + rng.normal(0, 0.01, (ns, 3)).astype(np.float32)
            guess = np.eye(4, dtype=np.float32)
            res_ = {}
            for nm, v in ctx.items():
                v.setInputSource(src)
                v.align(guess, want_output=False, want_fitness=True)
                res_[nm] = (v.getFinalTransformation().copy(), v.nr_iterations, v.getFitnessScore(), v.stats()["source_cells"])
            report["solves"] += 1
            for nm in ("seeds", "lists", "lazy", "device"):
                # (a solve's sums run over the scan in the order of ITS grid, whose cell size follows the crowding of the context's previous scan: a
                # context with another history -- the lazy one after a repeated solve -- may differ in the last bits; same grid: same bits)
                if res_[nm][3] == res_["none"][3]:
                    ok = np.array_equal(res_[nm][0], res_["none"][0], equal_nan=True) and res_[nm][1] == res_["none"][1] and res_[nm][2] == res_["none"][2]
                else:
                    report["solves_on_another_scan_grid"] = report.get("solves_on_another_scan_grid", 0) + 1
                    ok = np.allclose(res_[nm][0], res_["none"][0], rtol=0, atol=1e-5, equal_nan=True)
                if not ok:
                    report["failures"].append(dict(tag, frame=f, what="solve of route '%s' differs from 'none'" % nm,
                                                   dT=float(np.abs(res_[nm][0] - res_["none"][0]).max()), it=[res_[nm][1], res_["none"][1]]))
    except Exception as e:  # a refused cloud is a finding too
        report["failures"].append(dict(tag, what="exception: %r" % (e,)))
    for nm, v in ctx.items():
        v.device_free(dm[nm]); v.device_free(db[nm]); v.close()
    report["trials"] += 1
    report["by_kind"][kind] = report["by_kind"].get(kind, 0) + 1
    if len(report["failures"]) > 20:
        break
report["wall_s"] = round(time.time() - t_start, 1)
sf = report.pop("searched_fraction_lists")
report["lists_later_frames"] = {"frames": len(sf), "frames_where_lists_served_most_queries": int(sum(1 for x in sf if x < 0.5)), "smallest_searched_fraction": float(np.min(sf)) if sf else None}
print(json.dumps(report))
