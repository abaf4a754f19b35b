"""Randomised campaign of the local map resident on the device (rgc_map_*, f2) against the ORACLE's composition of transformPointCloud and
VoxelGrid: random sequences of insert (random keyframe poses and sizes, sweeps and noise clouds) / evict by count / evict by distance /
rebase / commit with a random leaf / an unrelated setInputTarget in between -- the stored points and the committed target bit for bit
after every operation.      python tests/fuzz/fuzz_map.py [trials] [seed] [operations per trial]"""
import sys, os, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import rgc_slam_amd.synth as synth
from rgc_slam_amd import registration, local_map, _lib
from oracle_backend import OracleBackend

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 20
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
n_ops = int(sys.argv[3]) if len(sys.argv) > 3 else 25


def quat(rng, ang):
    a = rng.normal(0, 1, 3); a *= ang / np.linalg.norm(a)
    th = np.linalg.norm(a)
    return np.concatenate([np.sin(th / 2) * a / th, [np.cos(th / 2)]])


rep = {"trials": 0, "operations": {}, "commits_compared": 0, "failures": []}
t0 = time.time()
for trial in range(trials):
    rng = np.random.default_rng(seed0 * 32452843 + trial)
    tag = {"trial": trial}
    reg = registration.odometer_vgicp(0)
    try:
        world = synth.make_world(half_extent=float(rng.choice([30.0, 45.0])), seed=int(rng.integers(1, 1 << 30)))
        m = local_map.RollingLocalMap(reg)
        ob = OracleBackend()
        origin = rng.uniform(-200, 200, 3) * np.array([1, 1, 0.02])
        m.reset(origin); ob.map_reset(origin)
        pos = origin.copy()
        n_kf = 0
        for op_i in range(n_ops):
            op = str(rng.choice(["insert", "insert", "insert", "commit", "commit", "evict_n", "evict_r", "rebase", "plain_target"]))
            tag.update(op=op, op_i=op_i)
            rep["operations"][op] = rep["operations"].get(op, 0) + 1
            if op == "insert":
                if rng.random() < 0.8:
                    T = synth.se3(synth.rot_zyx(rng.uniform(-np.pi, np.pi), rng.normal(0, 0.02), rng.normal(0, 0.02)), rng.uniform(-8, 8, 3) * np.array([1, 1, 0.01]))
                    sc = synth.make_scan(world, T, n_az=int(rng.integers(100, 1200)), seed=int(rng.integers(1, 1 << 30)))
                    cloud = np.concatenate([sc["xyz"], sc["intensity"][:, None]], axis=1).astype(np.float32)
                else:
                    k = int(rng.integers(1, 5000))
                    cloud = np.c_[rng.normal(0, 10, (k, 3)), rng.uniform(0, 100, k)].astype(np.float32)
                pos = pos + rng.normal(0, 1.5, 3) * np.array([1, 1, 0.05])
                q = quat(rng, rng.uniform(1e-3, 3.0))
                a, b = m.insert(cloud, q, pos), ob.map_insert(cloud, q, pos)
                if a != b:
                    rep["failures"].append(dict(tag, error="insert returned another id", hip=int(a), oracle=int(b)))
                n_kf += 1
            elif op == "evict_n":
                keep = int(rng.integers(1, 6))
                a, b = m.evict(keep), ob.map_evict(keep)
                if a != b:
                    rep["failures"].append(dict(tag, error="evict by count", hip=int(a), oracle=int(b)))
            elif op == "evict_r":
                r = float(rng.uniform(0.5, 6.0))
                a, b = m.evict(0, pos, r), ob.map_evict(0, pos, r)
                if a != b:
                    rep["failures"].append(dict(tag, error="evict by distance", hip=int(a), oracle=int(b)))
            elif op == "rebase":
                o2 = pos + rng.normal(0, 20, 3) * np.array([1, 1, 0.02])
                m.rebase(o2); ob.map_rebase(o2)
            elif op == "plain_target":
                reg.setInputTarget(rng.uniform(-5, 5, (200, 3)).astype(np.float32))
            if op == "commit":
                leaf = float(rng.choice([0.2, 0.3, 0.5]))
                empty = len(ob.map_points()) == 0
                try:
                    n = m.commit(leaf)
                    if empty:
                        rep["failures"].append(dict(tag, error="commit of an empty map did not fail"))
                    else:
                        exp = ob.map_target(leaf)
                        rep["commits_compared"] += 1
                        if not (n == len(exp) and np.array_equal(m.target(), exp)):
                            rep["failures"].append(dict(tag, error="committed target differs", n=[int(n), len(exp)]))
                except _lib.RgcError as e:
                    # (a target of fewer points than k is refused like any target: not a difference)
                    if not empty and len(ob.map_target(leaf)) >= 20:
                        rep["failures"].append(dict(tag, error="commit failed: %s" % (e,)))
            pts_h, pts_o = m.points(), ob.map_points()
            if not (pts_h.shape == pts_o.shape and np.array_equal(pts_h, pts_o)):
                rep["failures"].append(dict(tag, error="stored points differ", shapes=[list(pts_h.shape), list(pts_o.shape)]))
                break
            info = m.info()
            if info["n_keyframes"] != len(ob._kf):
                rep["failures"].append(dict(tag, error="keyframe count", hip=int(info["n_keyframes"]), oracle=len(ob._kf)))
                break
    except Exception as e:
        import traceback
        rep["failures"].append(dict(tag, error="exception: %r" % (e,), where=traceback.format_exc()[-500:]))
    reg.close()
    rep["trials"] += 1
    if len(rep["failures"]) > 10:
        break
rep["wall_s"] = round(time.time() - t0, 1)
print(json.dumps(rep))
