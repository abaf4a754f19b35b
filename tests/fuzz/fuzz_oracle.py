"""Randomised differential campaign against the ORACLE (test infrastructure, oracle/): whole registrations on clouds and settings the fixed tests
do not hold -- both covariance sets, the voxel table, a linearisation at the guess, the solve's pose, the fitness.  Clouds: the synthetic world
with a raw sweep as the scan, uniform noise, sheets and a pole, a clump in a sparse field; k = 10 / 20 / 25, leaf 0.5 / 1 / 2 m, every
RegularizationMethod and VoxelAccumulationMode of the reference's interface, guesses on and off the truth.
    python tests/fuzz/fuzz_oracle.py [trials] [seed] [max target points]"""
import sys, os, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import rgc_slam_amd.synth as synth
from rgc_slam_amd import registration as reg
import oracle as orc

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 50
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
nmax = int(sys.argv[3]) if len(sys.argv) > 3 else 60000
THREADS = 14


def rot_angle(Ra, Rb):
    R = Ra.astype(np.float64) @ Rb.astype(np.float64).T
    w = 0.5 * np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    return float(np.arcsin(min(1.0, np.linalg.norm(w))))


def problem(kind, n, rng):
    """target cloud, source cloud (both float32 (m, 3)), the motion between them"""
    d = synth.se3(synth.rot_zyx(*(rng.normal(0, 0.02, 3))), rng.normal(0, 0.12, 3) * np.array([1, 1, 0.2]))
    if kind == "synth":
        world, tgt = synth.make_world_and_map(n, seed=int(rng.integers(1, 1 << 30)))
        ns = int(rng.choice([4000, 15000, 30000]))
        src = synth.make_scan_n(world, d, ns, seed=int(rng.integers(1, 1 << 30)))["xyz"]
        return tgt.astype(np.float32), src.astype(np.float32), d
    if kind == "uniform":
        side = (n / rng.uniform(5.0, 40.0)) ** (1.0 / 3.0)
        tgt = rng.uniform(-side / 2, side / 2, (n, 3))
    elif kind == "sheets":
        a = np.c_[rng.uniform(-15, 15, (n // 2, 2)), np.zeros(n // 2)]
        b = np.c_[rng.uniform(-15, 15, n // 3), np.full(n // 3, 3.0), rng.uniform(0, 6, n // 3)]
        m = n - n // 2 - n // 3
        c = np.c_[np.full(m, 1.5), np.full(m, -2.0), rng.uniform(0, 8, m)]
        tgt = np.vstack([a, b, c]) + rng.normal(0, 2e-3, (n, 3))
    else:  # clump
        a = rng.normal(0, 0.6, (n // 2, 3))
        b = rng.uniform(-30, 30, (n - n // 2, 3)) * np.array([1, 1, 0.1])
        tgt = np.vstack([a, b])
    tgt = tgt.astype(np.float32)[rng.permutation(n)]
    ns = int(min(n, rng.integers(300, 12000)))
    sel = rng.choice(n, ns, replace=False)
    src = ((tgt[sel].astype(np.float64) - d[:3, 3]) @ d[:3, :3]).astype(np.float32) + rng.normal(0, 0.01, (ns, 3)).astype(np.float32)
    return tgt, src, d


KINDS = ["synth", "synth", "synth", "uniform", "sheets", "clump"]
rep = {"trials": 0, "failures": [], "by_kind": {}, "general_route_trials": 0, "max": {"cov_src": 0.0, "cov_tgt": 0.0, "vox_mean": 0.0, "vox_cov": 0.0, "H_rel": 0.0, "b_rel": 0.0,
                                                                                         "cost_rel": 0.0, "dt": 0.0, "dtheta": 0.0, "fitness_rel": 0.0}}
t0 = time.time()
for trial in range(trials):
    rng = np.random.default_rng(seed0 * 7919 + trial)
    kind = KINDS[int(rng.integers(0, len(KINDS)))]
    n = int(np.exp(rng.uniform(np.log(3000), np.log(nmax))))
    res = float(rng.choice([0.5, 1.0, 1.0, 2.0]))
    k = int(rng.choice([20, 20, 20, 10, 25]))
    method, mode = reg.FastVGICP.REG_PLANE, reg.FastVGICP.VOXEL_ADDITIVE
    if rng.random() < 0.25:
        method, mode = int(rng.integers(0, 5)), int(rng.integers(0, 3))
    tgt, src, d = problem(kind, n, rng)
    guess = np.eye(4) if rng.random() < 0.5 else synth.se3(synth.rot_zyx(*(rng.normal(0, 0.01, 3))), rng.normal(0, 0.05, 3))
    tag = {"trial": trial, "kind": kind, "n_target": len(tgt), "n_source": len(src), "res": res, "k": k, "method": method, "mode": mode}
    general = not (method == reg.FastVGICP.REG_PLANE and mode != reg.FastVGICP.VOXEL_MULTIPLICATIVE)
    rep["general_route_trials"] += int(general)

    def bad(what, **kw):
        rep["failures"].append(dict(tag, what=what, **kw))

    def note(key, val):
        rep["max"][key] = max(rep["max"][key], float(val))
    try:
        v = reg.odometer_vgicp(0)
        v.setResolution(res); v.setCorrespondenceRandomness(k); v.setRegularizationMethod(method); v.setVoxelAccumulationMode(mode)
        o = orc.Registration(voxel_res=res, max_iterations=25, translation_eps=1e-6, num_threads=THREADS, k_correspondences=k, regularization=method, voxel_mode=mode)
        v.setInputTarget(tgt); v.setInputSource(src)
        o.set_target(tgt); o.set_source(src); o.prepare()
        cs, ct = v.getSourceCovariances(), v.getTargetCovariances()
        es, et = np.abs(cs - o.source_cov(len(cs))).max(), np.abs(ct - o.target_cov(len(ct))).max()
        note("cov_src", es); note("cov_tgt", et)
        # (NONE / FROBENIUS keep the raw neighbourhood covariance: absolute 1e-9 on entries of order 0.1-1)
        # (FROBENIUS inverts C + 1e-3 I, normalises, inverts again, fast_gicp_impl.hpp:283-288: a collinear neighbourhood -- a far ring of a sweep --
        # has condition 1e3-1e4, and two summation orders end 1e-9..1e-8 apart)
        ctol = 1e-7 if method == reg.FastVGICP.REG_FROBENIUS else 1e-9
        if not (es <= ctol and et <= ctol):
            bad("covariances", src=float(es), tgt=float(et), n_bad=[int((np.abs(cs - o.source_cov(len(cs))).reshape(len(cs), -1).max(1) > 1e-9).sum()),
                                                                     int((np.abs(ct - o.target_cov(len(ct))).reshape(len(ct), -1).max(1) > 1e-9).sum())])
        vm, om = v.getVoxels(), o.voxelmap()
        ko, kv = np.lexsort(om["coords"].T[::-1]), np.lexsort(vm["coords"].T[::-1])
        if not (np.array_equal(vm["coords"][kv], om["coords"][ko]) and np.array_equal(vm["num"][kv], om["num"][ko])):
            bad("voxel table: coordinates / counts")
        else:
            em, ec = np.abs(vm["mean"][kv] - om["mean"][ko]).max(), np.abs(vm["cov"][kv] - om["cov"][ko]).max()
            note("vox_mean", em); note("vox_cov", ec)
            scale = max(1.0, float(np.abs(om["cov"]).max()))
            if not (em <= 1e-9 and ec <= 1e-8 * scale):
                bad("voxel table: values", mean=float(em), cov=float(ec), scale=scale)
        cost, H, b = v.linearize(guess)
        ocost, oH, ob = o.linearize(guess)
        if v.num_correspondences != o.num_correspondences:
            bad("correspondence count", hip=int(v.num_correspondences), oracle=int(o.num_correspondences))
        elif o.num_correspondences > 0:
            hr, br, cr = np.abs(H - oH).max() / max(np.abs(oH).max(), 1e-300), np.abs(b - ob).max() / max(np.abs(ob).max(), 1e-300), abs(cost - ocost) / max(abs(ocost), 1e-300)
            note("H_rel", hr); note("b_rel", br); note("cost_rel", cr)
            if not (hr <= 1e-8 and br <= 1e-7 and cr <= 1e-8):
                bad("linearisation", H=float(hr), b=float(br), cost=float(cr))
        v.align(guess.astype(np.float32), want_output=False, want_fitness=True)
        To = o.align(guess.astype(np.float32))
        T = v.getFinalTransformation()
        if np.all(np.isfinite(To)) and np.all(np.isfinite(T)):
            dt, dth = float(np.abs(T[:3, 3] - To[:3, 3]).max()), rot_angle(T[:3, :3], To[:3, :3])
            note("dt", dt); note("dtheta", dth)
            # (the pose bar of the path: 1e-4 m / 1e-4 rad; a solve that did not converge on either side ends wherever its last accepted step was)
            # a solve that used up max_iterations was still moving (a flat valley: a sparse map): the two paths' rounding noise has had 25 steps to grow
            ptol = 1e-3 if (v.nr_iterations >= 25 or o.iterations >= 25) else 1e-4
            if v.hasConverged() and o.converged and not (dt <= ptol and dth <= ptol):
                bad("pose", dt=dt, dtheta=dth, it=[int(v.nr_iterations), int(o.iterations)])
            if v.hasConverged() != o.converged and not (dt <= ptol and dth <= ptol):
                bad("convergence flag and pose", hip=bool(v.hasConverged()), oracle=bool(o.converged), dt=dt, dtheta=dth, it=[int(v.nr_iterations), int(o.iterations)])
            fo = o.fitness()
            fr = abs(v.getFitnessScore() - fo) / max(abs(fo), 1e-300)
            if dt <= 1e-6 and dth <= 1e-6:
                note("fitness_rel", fr)
                # (the score is a mean of squared nearest-neighbour distances d^2: a pose difference e moves it by ~ 2 e / d relative)
                if not fr <= 1e-5 + 4.0 * (dt + dth * 30.0) / max(np.sqrt(abs(fo)), 1e-12):
                    bad("fitness", hip=float(v.getFitnessScore()), oracle=float(fo))
        elif np.all(np.isfinite(To)) != np.all(np.isfinite(T)):
            bad("one side ends in NaN", hip=bool(np.all(np.isfinite(T))), oracle=bool(np.all(np.isfinite(To))))
        v.close()
    except Exception as e:
        bad("exception: %r" % (e,))
    rep["trials"] += 1
    rep["by_kind"][kind] = rep["by_kind"].get(kind, 0) + 1
    if len(rep["failures"]) > 25:
        break
rep["wall_s"] = round(time.time() - t0, 1)
print(json.dumps(rep))
