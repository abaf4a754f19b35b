"""Degenerate clouds through the registration's calls: every point the same, points on a line, an exact lattice sheet, two clumps a hundred
kilometres apart, coordinates of 1e6 m, exactly k points, k + 1, one NaN / inf among good points, a single far outlier -- as the target, as
the scan, registered to themselves.  What must hold: a clean refusal or a result, never a crash or a hang; where the oracle's answer is
defined (covariances of non-degenerate neighbourhoods), it is the library's.      python tests/fuzz/fuzz_degenerate.py [seed]"""
import sys, os, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import rgc_slam_amd.synth as synth
from rgc_slam_amd import registration as reg, _lib
import oracle as orc

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
rng = np.random.default_rng(seed)
world, base = synth.make_world_and_map(20000, seed=5)
base = base.astype(np.float32)
cases = {
    "all the same point": np.repeat(np.float32([[1.5, -2.0, 0.3]]), 500, axis=0),
    "two distinct points": np.repeat(np.float32([[0, 0, 0], [0.1, 0, 0]]), 300, axis=0),
    "a line": np.c_[np.linspace(-30, 30, 4000), np.zeros(4000), np.zeros(4000)].astype(np.float32),
    "a lattice sheet": np.stack(np.meshgrid(np.arange(-40, 40) * 0.25, np.arange(-40, 40) * 0.25, [0.0], indexing="ij"), -1).reshape(-1, 3).astype(np.float32),
    "two clumps 100 km apart": np.vstack([rng.normal(0, 1, (2000, 3)), rng.normal(0, 1, (2000, 3)) + [1e5, 0, 0]]).astype(np.float32),
    "coordinates of 1e6 m": (base[:8000] + np.float32([1e6, -1e6, 100.0])),
    "exactly k points": base[:20].copy(),
    "k + 1 points": base[:21].copy(),
    "k - 1 points": base[:19].copy(),
    "one NaN": np.vstack([base[:3000], [[np.nan, 0, 0]]]).astype(np.float32),
    "one inf": np.vstack([base[:3000], [[np.inf, 0, 0]]]).astype(np.float32),
    "one outlier 5 km off": np.vstack([base[:6000], [[5000.0, 0, 0]]]).astype(np.float32),
    "an empty cloud": np.zeros((0, 3), np.float32),
    "three points": base[:3].copy(),
    "dense clump of 20 000 in a decimetre": (rng.normal(0, 0.03, (20000, 3))).astype(np.float32),
}
rep = {"cases": {}, "failures": []}
for name, cloud in cases.items():
    out = {}
    for role in ("target", "source"):
        v = reg.odometer_vgicp(0)
        try:
            (v.setInputTarget if role == "target" else v.setInputSource)(cloud)
            c = (v.getTargetCovariances if role == "target" else v.getSourceCovariances)()
            ok = bool(np.all(np.isfinite(c)))
            out[role] = "covariances: %d, finite: %s" % (len(c), ok)
            if not ok:
                rep["failures"].append(dict(case=name, role=role, error="non-finite covariances"))
            if name in ("coordinates of 1e6 m", "one outlier 5 km off", "k + 1 points", "exactly k points") and ok:
                oc, _ = orc.covariances(cloud.copy(), k=20, threads=14)
                e = float(np.abs(c - oc).max())
                out[role] += ", vs oracle %.1e" % e
                if not e <= 1e-9:
                    rep["failures"].append(dict(case=name, role=role, error="covariances differ from the oracle's", err=e))
        except _lib.RgcError as e:
            out[role] = "refused: " + str(e)[:90]
        except Exception as e:
            out[role] = "EXCEPTION %r" % (e,)
            rep["failures"].append(dict(case=name, role=role, error=out[role]))
        v.close()
    v = reg.odometer_vgicp(0)
    try:
        v.setInputTarget(cloud); v.setInputSource(cloud)
        v.align(np.eye(4, dtype=np.float32), want_output=False, want_fitness=True)
        T = v.getFinalTransformation()
        out["to itself"] = "iterations %d, converged %s, |T - I| %.1e, fitness %.1e" % (v.nr_iterations, v.hasConverged(), float(np.abs(T - np.eye(4)).max()), v.getFitnessScore())
    except _lib.RgcError as e:
        out["to itself"] = "refused: " + str(e)[:90]
    except Exception as e:
        out["to itself"] = "EXCEPTION %r" % (e,)
        rep["failures"].append(dict(case=name, role="solve", error=out["to itself"]))
    v.close()
    rep["cases"][name] = out
print(json.dumps(rep, indent=1))
