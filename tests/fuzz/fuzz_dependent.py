"""Randomised campaign of the headline's path: a dependent sequence (the map re-framed by the previous pose, rgc_set_target_reframed /
rgc_align_end_reframe) on random maps, scans, lengths, reuse modes and lazy margins -- one context, two contexts, the Python frame loop, the C++
frame loop (librgc_seq.so) -- all against the plain calls a caller without any of it would make on one context: rgc_transform_cloud,
rgc_set_target_device, rgc_set_source_device, rgc_align, the world pose composed in numpy.  Motions and world poses bit for bit.
    python tests/fuzz/fuzz_dependent.py [trials] [seed] [max map points]"""
import sys, os, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import bench
import rgc_slam_amd.synth as synth
from rgc_slam_amd import registration

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 20
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
nmax = int(sys.argv[3]) if len(sys.argv) > 3 else 250000
rep = {"trials": 0, "frames": 0, "variants_compared": 0, "failures": [], "max_d": 0.0}
t0 = time.time()
for trial in range(trials):
    rng = np.random.default_rng(seed0 * 122949829 + trial)
    nt, ns, K = int(np.exp(rng.uniform(np.log(20000), np.log(nmax)))), int(rng.integers(2000, 30000)), int(rng.integers(5, 12))
    reuse, lazy = int(rng.integers(0, 3)), int(rng.choice([0, 0, 2]))
    tag = {"trial": trial, "n_target": nt, "n_source": ns, "frames": K, "reuse": reuse, "lazy": lazy}
    try:
        world, tgt = synth.make_world_and_map(nt, seed=int(rng.integers(1, 1 << 30)))
        poses = synth.make_trajectory(K + 1, seed=int(rng.integers(1, 1 << 30)))
        scans = [synth.make_scan_n(world, poses[i + 1], ns, seed=int(rng.integers(1, 1 << 30)))["xyz"] for i in range(K)]
        Tw0 = np.asarray(poses[0], np.float64)
        I4 = np.eye(4, dtype=np.float32)
        # ---- the plain calls, nothing kept: on one context, and on two taking turns (a scan's grid follows the previous scan of ITS context --
        #      DESIGN.md 5.1 -- so the two-context variants below are compared with the plain calls made on two contexts the same way) ----
        p = registration.odometer_vgicp(0)
        p.setNeighbourReuse(0)
        p2 = registration.odometer_vgicp(0)
        p2.setNeighbourReuse(0)

        def to_dev(v, xyz):
            a = np.zeros((xyz.shape[0], 4), np.float32); a[:, :3] = xyz
            d = v.device_alloc(a.nbytes); v.upload(d, a); return d
        dm, db = to_dev(p, tgt), p.device_alloc(16 * len(tgt))
        ds = [to_dev(p, s) for s in scans]
        refs = {}
        for two in (False, True):
            ref_m, ref_w = [], []
            Tw, g = Tw0.copy(), I4.copy()
            for i in range(K):
                c_ = p2 if (two and i % 2 == 1) else p
                q, t = bench.world_to_body(Tw)
                c_.transformCloudDevice(dm, len(tgt), 16, q, t, db)
                c_.setInputTargetDevice(db, len(tgt), 16)
                c_.setInputSourceDevice(ds[i], len(scans[i]), 16)
                c_.align(g, want_output=False, want_fitness=True)
                T = c_.getFinalTransformation()
                Tw = bench.compose_world(Tw, T)   # world_T * T in fp64, rows in ascending k (rgc_align_end_reframe's composition)
                g = T
                ref_m.append(T.copy()); ref_w.append(Tw.copy())
            refs[two] = (ref_m, ref_w)
            if two:
                p2.close()
                p.close()
                p = None
            else:   # (the one-context reference is done: a fresh pair for the two-context one)
                p.close()
                p = registration.odometer_vgicp(0); p.setNeighbourReuse(0)
                dm, db = to_dev(p, tgt), p.device_alloc(16 * len(tgt))
                ds = [to_dev(p, s_) for s_ in scans]
        rep["frames"] += K
        # ---- the sequence's own entry points, each variant on contexts of its own (a scan's grid follows its context's previous scan: the
        #      plain calls above start from a fresh context, so must these) ----
        for name, overlap, cpp in (("python loop, one context", False, False), ("python loop, two contexts", True, False),
                                   ("c++ loop, one context", False, True), ("c++ loop, two contexts", True, True)):
            pv = registration.PipelinedVGICP(0, depth=2)
            for w in pv.v:
                w.setNeighbourReuse(reuse)
                if lazy:
                    w.setLazyTarget(lazy)
            d_map = to_dev(pv.v[0], tgt)
            d_scans = [to_dev(pv.v[0], s) for s in scans]
            seq = bench.DependentSequence(pv.v, d_map, len(tgt), d_scans, [len(s) for s in scans])
            m, wd, _ = (seq.run_cpp if cpp else seq.run)(0, K, Tw0, I4, overlap)
            rep["variants_compared"] += 1
            ref_m, ref_w = refs[overlap]
            same = all(np.array_equal(a_, b_) for a_, b_ in zip(m, ref_m)) and all(np.array_equal(a_, b_) for a_, b_ in zip(wd, ref_w))
            if not same:
                d = max(float(np.abs(a_ - b_).max()) for a_, b_ in zip(m, ref_m))
                rep["max_d"] = max(rep["max_d"], d)
                first = next((i for i, (a_, b_) in enumerate(zip(m, ref_m)) if not np.array_equal(a_, b_)), -1)
                rep["failures"].append(dict(tag, error="differs from the plain calls: " + name, max_motion_diff=d, first_frame=first))
            seq.close()
            for w in pv.v:
                w.close()
    except Exception as e:
        import traceback
        rep["failures"].append(dict(tag, error="exception: %r" % (e,), where=traceback.format_exc()[-500:]))
    rep["trials"] += 1
    if len(rep["failures"]) > 10:
        break
rep["wall_s"] = round(time.time() - t0, 1)
print(json.dumps(rep))
