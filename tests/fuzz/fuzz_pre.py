"""Randomised differential campaign of the stages in front of the registration against the ORACLE (test infrastructure): the front-end A1-A8
(ring-major sweep, stencils, ground marking + plane, occlusion mask, feature selection), the leaf filter B3, de-skew B2 and re-framing B9.
Sweeps of 16 / 32 / 64 beams from random worlds, poses (tilted, moving), azimuth counts, with points dropped, NaNs, shuffled firing order,
truncated; leaf sizes 0.1-1 m on sweeps, noise, lattices whose points sit on leaf boundaries, through ONE filter object (its kept box).
    python tests/fuzz/fuzz_pre.py [trials] [seed]"""
import sys, os, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import rgc_slam_amd.synth as synth
from rgc_slam_amd import frontend, odometry, wire
import oracle as orc
from test_gpu_frontend import _compare      # the front-end's stage-by-stage comparison (tests/test_gpu_frontend.py)

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 50
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rep = {"trials": 0, "frontend": 0, "voxelgrid": 0, "deskew": 0, "transform": 0, "failures": [], "max": {"deskew": 0.0, "transform": 0.0}}
fe = {b: frontend.ScanRegistration(b) for b in (16, 32, 64)}
spec = {b: frontend.ScanRegistration(b) for b in (16, 32, 64)}   # the sweep left on the device: launches sized from the PREVIOUS sweep (whose size is random here)
msgfe = {b: frontend.ScanRegistration(b) for b in (16, 32, 64)}  # the callback on the message's bytes
pre = odometry.Preprocessor(0)
t0 = time.time()


def sweep(rng, beams):
    w = synth.make_world(half_extent=float(rng.choice([25.0, 40.0, 60.0])), seed=int(rng.integers(1, 1 << 30)))
    elev = synth.VLP16_ELEV if beams == 16 else (synth.hdl32_elev() if beams == 32 else synth.hdl64_elev())
    T = synth.se3(synth.rot_zyx(rng.uniform(-np.pi, np.pi), rng.normal(0, 0.02), rng.normal(0, 0.02)), rng.uniform(-10, 10, 3) * np.array([1, 1, 0.01]))
    n_az = int(rng.integers(150, 2200 if beams < 64 else 1200))
    sc = synth.make_scan(w, T, elev_deg=elev, n_az=n_az, seed=int(rng.integers(1, 1 << 30)))
    return sc


only = os.environ.get("FUZZ_ONLY")
for trial in range(trials):
    if only is not None and str(trial) not in only.split(","):
        continue
    rng = np.random.default_rng(seed0 * 104729 + trial)
    tag = {"trial": trial}
    try:
        # ---- front-end ----
        beams = int(rng.choice([16, 16, 32, 64]))
        sc = sweep(rng, beams)
        raw = np.concatenate([sc["xyz"], sc["intensity"][:, None]], axis=1).astype(np.float32)
        what = str(rng.choice(["plain", "plain", "dropped", "nan", "shuffled", "short", "high", "scaled"]))
        if what == "dropped":
            raw = raw[rng.random(len(raw)) < rng.uniform(0.2, 0.9)]
        elif what == "nan":
            raw = raw.copy(); raw[rng.random(len(raw)) < 0.02, int(rng.integers(0, 3))] = np.nan
        elif what == "shuffled":
            raw = raw[rng.permutation(len(raw))]
        elif what == "short":
            raw = raw[: int(rng.integers(1, 400))]
        elif what == "high":
            raw = raw[raw[:, 2] > rng.uniform(-0.3, 0.5)]
        elif what == "scaled":      # ranges below min_range / beyond max_range
            raw = raw.copy(); raw[:, :3] *= np.float32(rng.choice([0.01, 3.0]))
        tag.update(stage="frontend", beams=beams, n=len(raw), what=what)
        if len(raw) > 0:
            g, _o = _compare(fe[beams], orc, raw, beams, time_outliers=3 if what == "shuffled" else 0)
            rep["frontend"] += 1
            keys = ("curvature", "curvature2", "inten_curvature", "ground_marked", "picked", "label", "inten_label", "sharp", "flat", "inten", "ground_pts", "ring_count", "groundparam")
            a = spec[beams].laserCloudHandler(raw, cloud=False)
            bad = [k for k in keys if not np.array_equal(a[k], g[k])] + [k for k in ("n_cloud", "n_sharp_own", "n_ground", "ground_valid") if a[k] != g[k]]
            if bad:
                rep["failures"].append(dict(tag, error="the sweep left on the device differs from the synchronous path", keys=bad))
            m = np.zeros((len(raw), 8), np.float32); m[:, :3] = raw[:, :3]; m[:, 4] = raw[:, 3]
            lay = wire.layout(32, dict(x=(0, 7), y=(4, 7), z=(8, 7), intensity=(16, 7)))
            b = msgfe[beams].laserCloudHandlerMsg(m.tobytes(), len(raw), lay)
            bad = [k for k in keys + ("cloud",) if not np.array_equal(b[k], g[k])] + [k for k in ("n_cloud", "n_sharp_own", "n_ground", "ground_valid") if b[k] != g[k]]
            if bad:
                rep["failures"].append(dict(tag, error="the callback on the message bytes differs from the host path", keys=bad))
            rep["frontend_device_and_message"] = rep.get("frontend_device_and_message", 0) + 1
        # ---- leaf filter, on one object (the box kept from the previous cloud of the same leaf) ----
        leaf = float(rng.choice([0.1, 0.2, 0.3, 0.5, 1.0]))
        kind = str(rng.choice(["sweep", "sweep", "noise", "lattice", "shifted", "nan"]))
        base = np.concatenate([sc["xyz"], (sc["ring"] + 0.1 * sc["rel_time"])[:, None].astype(np.float32)], axis=1).astype(np.float32)
        if kind == "noise":
            m = int(rng.integers(10, 60000))
            base = np.c_[rng.uniform(-20, 20, (m, 3)), rng.uniform(0, 16, m)].astype(np.float32)
        elif kind == "lattice":     # points exactly on leaf boundaries (multiples of the leaf), negative coordinates too
            m = int(rng.integers(5, 40))
            g = np.stack(np.meshgrid(np.arange(-m, m), np.arange(-m, m), np.arange(-2, 3), indexing="ij"), -1).reshape(-1, 3).astype(np.float32) * np.float32(leaf * rng.choice([0.5, 1.0, 2.0]))
            base = np.c_[g, rng.uniform(0, 16, len(g))].astype(np.float32)[rng.permutation(len(g))]
        elif kind == "shifted":
            base = base.copy(); base[:, :3] += rng.uniform(-30, 30, 3).astype(np.float32)
        elif kind == "nan":
            base = base.copy(); base[rng.random(len(base)) < 0.01, int(rng.integers(0, 3))] = np.nan
        tag.update(stage="voxelgrid", leaf=leaf, kind=kind, n=len(base))
        if kind == "nan":           # a cloud with a non-finite coordinate is refused (RGC_ERR_NONFINITE: the reference's cloud went through removeNaN in
            try:                    # the front-end), the refusal leaves nothing behind, and the finite points filter like any cloud
                pre.voxelGridFilter(base, leaf)
                rep["failures"].append(dict(tag, error="a cloud with NaNs was not refused"))
            except Exception:
                pass
            base = base[np.all(np.isfinite(base), axis=1)]
        got, exp = pre.voxelGridFilter(base, leaf), orc.voxelgrid_filter(base, leaf)
        if not (got.shape == exp.shape and np.array_equal(got, exp)):
            rep["failures"].append(dict(tag, error="leaf filter differs", shapes=[list(got.shape), list(exp.shape)],
                                        err=float(np.abs(got - exp).max()) if got.shape == exp.shape else None))
        rep["voxelgrid"] += 1
        # ---- de-skew and re-framing ----
        fin = base[np.all(np.isfinite(base), axis=1)]
        if len(fin):
            R = synth.rot_zyx(*(rng.normal(0, 0.03, 3)))
            qw = np.sqrt(1 + np.trace(R)) / 2
            q = np.array([(R[2, 1] - R[1, 2]) / (4 * qw), (R[0, 2] - R[2, 0]) / (4 * qw), (R[1, 0] - R[0, 1]) / (4 * qw), qw])
            t = rng.normal(0, 0.2, 3)
            tag.update(stage="deskew")
            e = float(np.abs(pre.adjustDistortion(fin, q, t)[:, :3] - orc.deskew(fin, q, t)[:, :3]).max())
            rep["max"]["deskew"] = max(rep["max"]["deskew"], e); rep["deskew"] += 1
            if not e <= 4e-6 * max(1.0, float(np.abs(fin[:, :3]).max()) / 50.0):
                rep["failures"].append(dict(tag, error="de-skew", err=e))
            q2 = rng.normal(0, 1, 4); q2 /= np.linalg.norm(q2)
            t2 = rng.uniform(-50, 50, 3)
            tag.update(stage="transform")
            g2, e2 = pre.transformPointCloud(fin, q2, t2), orc.transform_cloud(fin, q2, t2)
            rep["transform"] += 1
            if not np.array_equal(g2, e2):       # (B9 is the oracle's fp64 expression stored as fp32: bit for bit, tests/test_gpu_parity.py)
                d2 = float(np.abs(g2 - e2).max())
                rep["max"]["transform"] = max(rep["max"]["transform"], d2)
                if not d2 <= 8e-6:
                    rep["failures"].append(dict(tag, error="re-framing", err=d2))
    except AssertionError as e:
        import traceback
        tb = traceback.extract_tb(e.__traceback__)[-1]
        rep["failures"].append(dict(tag, error="assertion at %s:%d `%s` %s" % (os.path.basename(tb.filename), tb.lineno, tb.line, str(e)[:200])))
    except Exception as e:
        rep["failures"].append(dict(tag, error="exception: %r" % (e,)))
    rep["trials"] += 1
    if len(rep["failures"]) > 25:
        break
rep["wall_s"] = round(time.time() - t0, 1)
print(json.dumps(rep))
