"""Contexts are independent: T host threads, each with contexts of its own, run random registrations, leaf filters and front-ends at the same
time (ctypes releases the interpreter lock inside a call); every result against the same call made alone beforehand.
    python tests/fuzz/fuzz_threads.py [threads] [jobs per thread] [seed]"""
import sys, os, json, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import rgc_slam_amd.synth as synth
from rgc_slam_amd import registration as reg, odometry, frontend

T = int(sys.argv[1]) if len(sys.argv) > 1 else 6
J = int(sys.argv[2]) if len(sys.argv) > 2 else 25
seed0 = int(sys.argv[3]) if len(sys.argv) > 3 else 1
world, base = synth.make_world_and_map(60000, seed=5)
base = base.astype(np.float32)
rep = {"threads": T, "jobs": 0, "failures": []}


def jobs_of(t):
    rng = np.random.default_rng(seed0 * 1000 + t)
    out = []
    for j in range(J):
        kind = str(rng.choice(["register", "register", "filter", "frontend"]))
        if kind == "register":
            nt, ns = int(rng.integers(5000, 60000)), int(rng.integers(1000, 15000))
            tgt = base[rng.choice(len(base), nt, replace=False)]
            d = synth.se3(synth.rot_zyx(*(rng.normal(0, 0.01, 3))), rng.normal(0, 0.08, 3))
            sel = rng.choice(nt, min(nt, ns), replace=False)
            src = ((tgt[sel].astype(np.float64) - d[:3, 3]) @ d[:3, :3]).astype(np.float32)
            out.append((kind, tgt, src))
        elif kind == "filter":
            c = base[rng.choice(len(base), int(rng.integers(2000, 60000)), replace=False)]
            a = np.zeros((len(c), 4), np.float32); a[:, :3] = c
            out.append((kind, a, float(rng.choice([0.2, 0.3, 0.5]))))
        else:
            sc = synth.make_scan(world, np.eye(4), n_az=int(rng.integers(200, 1200)), seed=int(rng.integers(1, 1 << 30)))
            out.append((kind, np.concatenate([sc["xyz"], sc["intensity"][:, None]], axis=1).astype(np.float32)))
    return out


def run(job, ctx):
    v, p, f = ctx
    if job[0] == "register":
        v.setInputTarget(job[1]); v.setInputSource(job[2])
        v.align(np.eye(4, dtype=np.float32), want_output=False, want_fitness=True)
        return (v.getFinalTransformation().copy(), v.nr_iterations, v.getFitnessScore())
    if job[0] == "filter":
        return p.voxelGridFilter(job[1], job[2])
    g = f.laserCloudHandler(job[1])
    return (g["cloud"].copy(), g["sharp"].copy(), g["flat"].copy(), g["groundparam"].copy())


def same(a, b):
    if isinstance(a, tuple):
        return all(same(x, y) for x, y in zip(a, b))
    return np.array_equal(a, b, equal_nan=True) if isinstance(a, np.ndarray) else a == b


all_jobs = [jobs_of(t) for t in range(T)]
# alone first: every thread's jobs in order on contexts of their own (a scan's grid follows its context's previous scan: the same order)
alone = []
for t in range(T):
    ctx = (reg.odometer_vgicp(0), odometry.Preprocessor(0), frontend.ScanRegistration(16))
    alone.append([run(j, ctx) for j in all_jobs[t]])
    for o in ctx:
        o.close()
results = [None] * T
errors = []


def worker(t):
    try:
        ctx = (reg.odometer_vgicp(0), odometry.Preprocessor(0), frontend.ScanRegistration(16))
        results[t] = [run(j, ctx) for j in all_jobs[t]]
        for o in ctx:
            o.close()
    except Exception as e:
        errors.append("thread %d: %r" % (t, e))


t0 = time.time()
th = [threading.Thread(target=worker, args=(t,)) for t in range(T)]
for x in th:
    x.start()
for x in th:
    x.join()
rep["wall_s"] = round(time.time() - t0, 1)
for e in errors:
    rep["failures"].append(dict(error=e))
for t in range(T):
    if results[t] is None:
        continue
    for j, (a, b) in enumerate(zip(results[t], alone[t])):
        rep["jobs"] += 1
        if not same(a, b):
            rep["failures"].append(dict(thread=t, job=j, kind=all_jobs[t][j][0], error="differs from the same call made alone"))
print(json.dumps(rep))
