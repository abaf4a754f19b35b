"""Trial of tests/fuzz/fuzz_modes.py where the lazy route's solve differs: which inputs of the solve differ?   FUZZ_ONLY-style: python tests/fuzz/fuzz_repro_lazy.py <trial> <seed> <nmax> <frame>"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import rgc_slam_amd.synth as synth
from rgc_slam_amd import registration
import bench
trial, seed0, nmax, fstop = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
src_txt = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "fuzz_modes.py")).read()
ns_ = {}
exec("import numpy as np\nimport rgc_slam_amd.synth as synth\n" + "def cloud" + src_txt.split("def cloud")[1].split("KINDS = ")[0], ns_)
KINDS = ["synth", "synth", "uniform", "lattice", "sheets", "clump", "repeats"]
rng = np.random.default_rng(seed0 * 100003 + trial)
kind = KINDS[int(rng.integers(0, len(KINDS)))]
n = int(np.exp(rng.uniform(np.log(2000), np.log(nmax))))
res = float(rng.choice([0.5, 1.0, 1.0, 2.0])); k = int(rng.choice([20, 20, 20, 20, 10, 25]))
pts = ns_["cloud"](kind, n, rng); n = len(pts)
a = np.zeros((n, 4), np.float32); a[:, :3] = pts
reach = float(rng.choice([3.0, 20.0, 60.0]))
print(kind, n, res, k, reach)
ctx = {}
for name, mode in (("none", 0), ("lazy", 2)):
    v = registration.odometer_vgicp(0); v.setResolution(res); v.setCorrespondenceRandomness(k); v.setNeighbourReuse(mode)
    if name == "lazy": v.setLazyTarget(2)
    ctx[name] = v
# the other three contexts of the campaign draw nothing from rng: the stream is the same
dm = {nm: v.device_alloc(a.nbytes) for nm, v in ctx.items()}; db = {nm: v.device_alloc(a.nbytes) for nm, v in ctx.items()}
for nm, v in ctx.items(): v.upload(dm[nm], a)
Tw = None
for f in range(5 if reach <= 3.0 else 3):
    if f and rng.random() < 0.4:
        j = rng.integers(0, n, max(1, n // 500)); a[j, :3] += rng.normal(0, 0.05, (len(j), 3)).astype(np.float32)
        for nm, v in ctx.items(): v.upload(dm[nm], a)
    if reach <= 3.0 and Tw is not None:
        Tw = Tw @ synth.se3(synth.rot_zyx(*(rng.normal(0, 0.03, 3) * np.array([1, 0.1, 0.1]))), rng.normal(0, 0.4, 3) * np.array([1, 1, 0.05]))
    else:
        ang = rng.uniform(-np.pi, np.pi, 3) * np.array([1.0, 0.03, 0.03]); Tw = synth.se3(synth.rot_zyx(*ang), rng.uniform(-reach, reach, 3) * np.array([1, 1, 0.05]))
    q, t = bench.world_to_body(Tw)
    for nm in ctx: ctx[nm].setInputTargetReframed(dm[nm], n, 16, q, t, db[nm])
    body = ctx["none"].download(db["none"], (n, 4))
    if f < fstop:
        ctx["none"].getTargetCovariances(); ctx["none"].getVoxels()   # (what the campaign does on 'none' every frame)
    ns = int(min(n, max(k + 1, rng.integers(300, 20000)))); sel = rng.choice(n, ns, replace=False)
    d = synth.se3(synth.rot_zyx(*(rng.normal(0, 0.02, 3))), rng.normal(0, 0.15, 3))
    src = ((body[sel, :3].astype(np.float64) - d[:3, 3]) @ d[:3, :3]).astype(np.float32) + rng.normal(0, 0.01, (ns, 3)).astype(np.float32)
    R = {}
    for nm, v in ctx.items():
        v.setInputSource(src); v.align(np.eye(4, dtype=np.float32), want_output=False, want_fitness=True)
        R[nm] = (v.getFinalTransformation().copy(), v.nr_iterations, v.getFitnessScore(), v.stats())
    same = np.array_equal(R["none"][0], R["lazy"][0])
    print("frame", f, "pose equal", same, "iters", R["none"][1], R["lazy"][1], "lazy_misses", R["lazy"][3]["lazy_misses"], "deferred", R["none"][3]["deferred_target"], R["lazy"][3]["deferred_target"], "source_cells", R["none"][3]["source_cells"], R["lazy"][3]["source_cells"], "target_cells", R["none"][3]["target_cells"], R["lazy"][3]["target_cells"], "dT", float(np.abs(R["none"][0] - R["lazy"][0]).max()))
    if f == fstop:
        cs0, cs1 = ctx["none"].getSourceCovariances(), ctx["lazy"].getSourceCovariances()
        print("  source covariances equal:", np.array_equal(cs0, cs1))
        l0, l1 = ctx["none"].linearize(np.eye(4)), ctx["lazy"].linearize(np.eye(4))
        print("  linearize at identity:", [float(np.abs(np.asarray(x, float) - np.asarray(y, float)).max()) for x, y in zip(l0, l1)] if isinstance(l0, (tuple, list)) else (l0, l1))
        c0, c1 = ctx["none"].getTargetCovariances().reshape(n, -1), ctx["lazy"].getTargetCovariances().reshape(n, -1)
        dd = np.nonzero(np.any(c0 != c1, axis=1))[0]
        print("  target covariances (after the lazy context completed its map): differ in", len(dd), float(np.abs(c0 - c1).max()))
        x0, x1 = ctx["none"].getVoxels(), ctx["lazy"].getVoxels()
        o0, o1 = np.lexsort(x0["coords"].T[::-1]), np.lexsort(x1["coords"].T[::-1])
        print("  voxels equal:", {kk: bool(np.array_equal(x0[kk][o0], x1[kk][o1])) for kk in ("coords", "num", "mean", "cov")})
        break
