"""Stateful randomised campaign of the registration's C-ABI through its Python mirror: random sequences of the calls a caller can make -- clouds
from the host / the device / re-framed on the device, settings, the lazy target, the reuse modes, begin / end halves, getters, swap, clear,
a shared target -- against a small model of what must succeed and what must be refused, and every successful solve against a FRESH context
given the same clouds and settings through the plain calls (pose to 1e-6: the routes' bits agree, a scan's grid may not).
A call that should be refused and is not, one that should work and is refused, any crash: a failure.
    python tests/fuzz/fuzz_api.py [trials] [seed] [operations per trial]"""
import sys, os, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import rgc_slam_amd.synth as synth
from rgc_slam_amd import registration as reg, local_map, _lib
import bench

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 20
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
n_ops = int(sys.argv[3]) if len(sys.argv) > 3 else 40
rep = {"trials": 0, "operations": {}, "refusals_expected": 0, "solves_compared": 0, "solves_bit_equal": 0, "failures": [], "max_dT": 0.0}
t0 = time.time()


class Model:
    def __init__(self):
        self.tgt = None; self.src = None; self.pending = False
        self.res, self.k, self.method, self.mode, self.lazy, self.reuse = 1.0, 20, 3, 0, 0, 2
        self.stale_settings = False   # a setting changed after the clouds were set: the campaign sets both again before it solves


def fresh_solve(m, guess):
    f = reg.odometer_vgicp(0)
    f.setResolution(m.res); f.setCorrespondenceRandomness(m.k); f.setRegularizationMethod(m.method); f.setVoxelAccumulationMode(m.mode)
    f.setInputTarget(m.tgt); f.setInputSource(m.src)
    f.align(guess, want_output=False, want_fitness=True)
    out = (f.getFinalTransformation().copy(), f.nr_iterations, f.getFitnessScore())
    f.close()
    return out


only = os.environ.get("FUZZ_ONLY")
for trial in range(trials):
    if only is not None and trial != int(only):
        continue
    if os.environ.get("FUZZ_PROGRESS"):
        print("trial", trial, file=sys.stderr, flush=True)
    rng = np.random.default_rng(seed0 * 86028121 + trial)
    world, base = synth.make_world_and_map(int(rng.integers(3000, 40000)), seed=int(rng.integers(1, 1 << 30)))
    base = base.astype(np.float32)
    v = reg.odometer_vgicp(0)
    other = reg.odometer_vgicp(0)
    m = Model()
    lm = local_map.RollingLocalMap(v)   # f2: the map resident on the device commits its target INTO this context (rgc_map_commit)
    lm.reset(None)
    n_in_map = 0
    nmax = len(base)
    bufs = []   # a device cloud belongs to the caller and must stay as it is while it is set (the library re-reads it when a speculative grid did not
                # hold, on a swap ...): every device cloud of the campaign gets a buffer of its own, freed at the trial's end

    def dev(a=None, nbytes=0):
        p_ = v.device_alloc(a.nbytes if a is not None else nbytes)
        bufs.append(p_)
        if a is not None:
            v.upload(p_, a)
        return p_
    tag = {"trial": trial}

    def new_cloud(lo=300):
        n = int(rng.integers(lo, nmax))
        sel = rng.choice(nmax, n, replace=False)
        T = synth.se3(synth.rot_zyx(*(rng.normal(0, 0.02, 3))), rng.normal(0, 0.1, 3))
        return (base[sel].astype(np.float64) @ T[:3, :3].T + T[:3, 3]).astype(np.float32)

    def expect(ok_expected, fn, what):
        """run fn; RgcError is the library's refusal"""
        try:
            r = fn()
        except _lib.RgcError as e:
            if ok_expected:
                rep["failures"].append(dict(tag, error="refused, should have worked: %s: %s" % (what, str(e)[:160])))
            else:
                rep["refusals_expected"] += 1
            return None, False
        if not ok_expected:
            rep["failures"].append(dict(tag, error="worked, should have been refused: %s" % what))
        return r, True

    try:
        for op_i in range(n_ops):
            op = str(rng.choice(["tgt_host", "tgt_dev", "tgt_reframed", "src_host", "src_dev", "align", "align", "begin", "end", "setting", "lazy", "reuse",
                                 "getters", "swap", "clear_src", "clear_tgt", "share", "set_cov", "map_insert", "map_commit", "map_evict", "end_reframe"]))
            tag.update(op=op, op_i=op_i)
            rep["operations"][op] = rep["operations"].get(op, 0) + 1
            if only is not None:
                print(op_i, op, dict(res=m.res, k=m.k, method=m.method, mode=m.mode, lazy=m.lazy, reuse=m.reuse, pending=m.pending, stale=m.stale_settings, tgt=None if m.tgt is None else len(m.tgt), src=None if m.src is None else len(m.src)), file=sys.stderr)
            if op in ("tgt_host", "tgt_dev", "tgt_reframed"):
                c = new_cloud(lo=max(300, m.k + 1))
                a = np.zeros((len(c), 4), np.float32); a[:, :3] = c
                if op == "tgt_host":
                    _, ok = expect(not m.pending, lambda: v.setInputTarget(c), op)
                    if ok: m.tgt = c
                elif op == "tgt_dev":
                    d_map = dev(a)
                    _, ok = expect(not m.pending, lambda: v.setInputTargetDevice(d_map, len(c), 16), op)
                    if ok: m.tgt = c
                else:
                    d_map, d_scr = dev(a), dev(nbytes=a.nbytes)
                    Tw = synth.se3(synth.rot_zyx(rng.uniform(-np.pi, np.pi), rng.normal(0, 0.02), rng.normal(0, 0.02)), rng.uniform(-20, 20, 3) * np.array([1, 1, 0.02]))
                    q, t = bench.world_to_body(Tw)
                    _, ok = expect(not m.pending, lambda: v.setInputTargetReframed(d_map, len(c), 16, q, t, d_scr), op)
                    if ok: m.tgt = v.download(d_scr, (len(c), 4))[:, :3].copy()
                if ok: m.stale_settings = False if m.src is None else m.stale_settings
            elif op in ("src_host", "src_dev"):
                c = new_cloud(lo=max(200, m.k + 1))
                if m.tgt is not None:   # near the target: a sub-sample of it, moved a little
                    sel = rng.choice(len(m.tgt), min(len(m.tgt), int(rng.integers(max(200, m.k + 1), 12000))), replace=False)
                    d = synth.se3(synth.rot_zyx(*(rng.normal(0, 0.01, 3))), rng.normal(0, 0.08, 3))
                    c = ((m.tgt[sel].astype(np.float64) - d[:3, 3]) @ d[:3, :3]).astype(np.float32) + rng.normal(0, 0.01, (len(sel), 3)).astype(np.float32)
                if op == "src_host":
                    _, ok = expect(not m.pending, lambda: v.setInputSource(c), op)
                else:
                    a = np.zeros((len(c), 4), np.float32); a[:, :3] = c
                    d_src = dev(a)
                    _, ok = expect(not m.pending, lambda: v.setInputSourceDevice(d_src, len(c), 16), op)
                if ok: m.src = c
            elif op in ("align", "begin"):
                ready = m.tgt is not None and m.src is not None and not m.pending
                if ready and m.stale_settings:     # (what a setting means for clouds set BEFORE it is not what this campaign is about: set them again)
                    v.setInputTarget(m.tgt); v.setInputSource(m.src); m.stale_settings = False
                guess = np.eye(4, dtype=np.float32) if rng.random() < 0.6 else synth.se3(synth.rot_zyx(*(rng.normal(0, 0.005, 3))), rng.normal(0, 0.03, 3)).astype(np.float32)
                if op == "align":
                    _, ok = expect(ready, lambda: v.align(guess, want_output=False, want_fitness=True), op)
                    if ok:
                        got = (v.getFinalTransformation().copy(), v.nr_iterations, v.getFitnessScore())
                        exp = fresh_solve(m, guess)
                        rep["solves_compared"] += 1
                        if np.all(np.isfinite(exp[0])) and np.all(np.isfinite(got[0])):
                            dT = float(np.abs(got[0] - exp[0]).max())
                            rep["max_dT"] = max(rep["max_dT"], dT)
                            rep["solves_bit_equal"] += int(np.array_equal(got[0], exp[0]) and got[1] == exp[1] and got[2] == exp[2])
                            if not dT <= (1e-6 if got[1] == exp[1] else 2e-4):
                                rep["failures"].append(dict(tag, error="solve differs from a fresh context's", dT=dT, it=[int(got[1]), int(exp[1])],
                                                            state=dict(res=m.res, k=m.k, method=m.method, mode=m.mode, lazy=m.lazy, reuse=m.reuse)))
                else:
                    _, ok = expect(ready, lambda: v.align_begin(guess, want_fitness=bool(rng.random() < 0.5)), op)
                    if ok: m.pending = True; m.pending_guess = guess
            elif op == "end":
                _, ok = expect(m.pending, lambda: v.align_end(), op)
                if ok:
                    m.pending = False
                    got = v.getFinalTransformation().copy()
                    exp = fresh_solve(m, m.pending_guess)
                    rep["solves_compared"] += 1
                    if np.all(np.isfinite(exp[0])) and np.all(np.isfinite(got)):
                        dT = float(np.abs(got - exp[0]).max())
                        rep["max_dT"] = max(rep["max_dT"], dT)
                        rep["solves_bit_equal"] += int(np.array_equal(got, exp[0]))
                        if not dT <= (1e-6 if v.nr_iterations == exp[1] else 2e-4):
                            rep["failures"].append(dict(tag, error="begin / end solve differs from a fresh context's", dT=dT, it=[int(v.nr_iterations), int(exp[1])]))
                            if only is not None:
                                f = reg.odometer_vgicp(0); f.setResolution(m.res); f.setCorrespondenceRandomness(m.k); f.setInputTarget(m.tgt); f.setInputSource(m.src)
                                print("converged", v.hasConverged(), "fitness at the context's pose", f.fitnessAt(got), "at the fresh one's", f.fitnessAt(exp[0]),
                                      "cost", f.linearize(got.astype(np.float64))[0], f.linearize(exp[0].astype(np.float64))[0], "guess cost", f.linearize(m.pending_guess.astype(np.float64))[0],
                                      "\nsource covariances equal", np.array_equal(f.getSourceCovariances(), v.getSourceCovariances()), "target covariances equal", np.array_equal(f.getTargetCovariances(), v.getTargetCovariances()),
                                      "\nstats", v.stats(), "\nfresh", f.stats(), file=sys.stderr)
                                v.align(m.pending_guess, want_output=False); print("the context's blocking solve now:", float(np.abs(v.getFinalTransformation() - exp[0]).max()), v.nr_iterations, file=sys.stderr)
            elif op == "setting":
                if m.pending:
                    continue
                which = int(rng.integers(0, 4))
                if which == 0: m.res = float(rng.choice([0.5, 1.0, 2.0])); v.setResolution(m.res)
                elif which == 1: m.k = int(rng.choice([10, 20, 25])); v.setCorrespondenceRandomness(m.k)
                elif which == 2: m.method = int(rng.integers(0, 5)); v.setRegularizationMethod(m.method)
                else: m.mode = int(rng.integers(0, 3)); v.setVoxelAccumulationMode(m.mode)
                m.stale_settings = True
                other.setResolution(m.res); other.setCorrespondenceRandomness(m.k); other.setRegularizationMethod(m.method); other.setVoxelAccumulationMode(m.mode)
            elif op == "lazy":
                if m.pending:
                    continue
                m.lazy = int(rng.choice([0, 2, 3])); v.setLazyTarget(m.lazy); m.stale_settings = True
            elif op == "reuse":
                if m.pending:
                    continue
                m.reuse = int(rng.integers(0, 3)); v.setNeighbourReuse(m.reuse)
            elif op == "getters":
                have_t, have_s = m.tgt is not None and not m.stale_settings, m.src is not None and not m.stale_settings
                if m.pending or m.stale_settings:
                    continue
                r, ok = expect(have_t, lambda: v.getTargetCovariances(), "getTargetCovariances")
                if ok and len(r) != len(m.tgt): rep["failures"].append(dict(tag, error="target covariances: wrong count"))
                r, ok = expect(have_s, lambda: v.getSourceCovariances(), "getSourceCovariances")
                if ok and len(r) != len(m.src): rep["failures"].append(dict(tag, error="source covariances: wrong count"))
                expect(have_t, lambda: v.getVoxels(), "getVoxels")
                expect(have_t and have_s, lambda: v.linearize(np.eye(4)), "linearize")
                expect(have_t and have_s, lambda: v.fitnessAt(np.eye(4, dtype=np.float32)), "fitnessAt")
                v.stats()
            elif op == "swap":
                if m.pending or m.stale_settings:
                    continue
                both = m.tgt is not None and m.src is not None
                _, ok = expect(both, lambda: v.swapSourceAndTarget(), op)
                if ok: m.tgt, m.src = m.src, m.tgt
            elif op == "clear_src":
                if m.pending:
                    continue
                v.clearSource(); m.src = None
            elif op == "clear_tgt":
                if m.pending:
                    continue
                v.clearTarget(); m.tgt = None
            elif op == "share":
                if m.pending or m.tgt is None or m.stale_settings or m.src is None:
                    continue
                other.setResolution(m.res); other.setCorrespondenceRandomness(m.k); other.setRegularizationMethod(m.method); other.setVoxelAccumulationMode(m.mode)
                _, ok = expect(True, lambda: other.shareTargetFrom(v), op)
                if ok:
                    other.setInputSource(m.src)
                    g = np.eye(4, dtype=np.float32)
                    _, ok2 = expect(True, lambda: other.align(g, want_output=False, want_fitness=True), "align on a shared target")
                    if ok2:
                        exp = fresh_solve(m, g)
                        rep["solves_compared"] += 1
                        if np.all(np.isfinite(exp[0])) and np.all(np.isfinite(other.getFinalTransformation())):
                            dT = float(np.abs(other.getFinalTransformation() - exp[0]).max())
                            rep["max_dT"] = max(rep["max_dT"], dT)
                            rep["solves_bit_equal"] += int(np.array_equal(other.getFinalTransformation(), exp[0]))
                            if not dT <= (1e-6 if other.nr_iterations == exp[1] else 2e-4):
                                rep["failures"].append(dict(tag, error="solve on a shared target differs", dT=dT))
                                if only is not None:
                                    f = reg.odometer_vgicp(0); f.setResolution(m.res); f.setCorrespondenceRandomness(m.k); f.setInputTarget(m.tgt); f.setInputSource(m.src)
                                    ct, cs = f.getTargetCovariances(), f.getSourceCovariances()
                                    vt, vs = v.getTargetCovariances(), v.getSourceCovariances()
                                    print("owner: target covariances equal a fresh context's:", np.array_equal(ct, vt), float(np.abs(ct - vt).max()), " source:", np.array_equal(cs, vs), float(np.abs(cs - vs).max()),
                                          "\n voxels:", v.stats()["n_voxels"], f.stats()["n_voxels"], file=sys.stderr)
                                    v.align(g, want_output=False); print(" owner's own solve vs fresh:", float(np.abs(v.getFinalTransformation() - exp[0]).max()), file=sys.stderr)
                                    f.close()
            elif op == "map_insert":
                c = new_cloud(lo=300)[: int(rng.integers(200, 6000))]
                a = np.zeros((len(c), 4), np.float32); a[:, :3] = c
                th = rng.normal(0, 0.05)
                q = np.array([0, 0, np.sin(th / 2), np.cos(th / 2)])
                _, ok = expect(True, lambda: lm.insert(a, q, rng.normal(0, 0.5, 3) * np.array([1, 1, 0.05])), op)   # (the store is not the target: allowed with a solve in flight)
                n_in_map += int(ok)
            elif op == "map_evict":
                _, ok = expect(True, lambda: lm.evict(int(rng.integers(1, 4))), op)
                if ok: n_in_map = min(n_in_map, 3)
            elif op == "map_commit":
                if n_in_map == 0 or m.stale_settings or m.lazy:
                    continue
                leaf = float(rng.choice([0.3, 0.5]))
                if m.pending:   # (a commit of an unchanged, still bound map is a no-op and may pass; anything that would touch the target must be refused)
                    try:
                        lm.commit(leaf)
                        if not np.array_equal(lm.target()[:, :3], m.tgt):
                            rep["failures"].append(dict(tag, error="a commit changed the target under a solve in flight"))
                            if only is not None:
                                t_now = lm.target()
                                print("commit in flight: leaf", leaf, "target now", t_now.shape, "model", m.tgt.shape, "same prefix", np.array_equal(t_now[:10, :3], m.tgt[:10]), file=sys.stderr)
                                print("info", lm.info(), file=sys.stderr)
                    except _lib.RgcError:
                        rep["refusals_expected"] += 1
                    continue
                r, ok = expect(True, lambda: lm.commit(leaf), op)
                if ok:
                    m.tgt = lm.target()[:, :3].copy()
                    if r != len(m.tgt): rep["failures"].append(dict(tag, error="commit's count and the target's differ"))
            elif op == "end_reframe":
                # rgc_align_end_reframe: collect v's solve, compose the world pose, and enqueue the NEXT target -- a world map re-expressed in the new
                # body frame -- on `other`.  All or nothing for the caller's arguments: refused without a solve in flight (nothing happens).
                wm = new_cloud(lo=2000)
                aw = np.zeros((len(wm), 4), np.float32); aw[:, :3] = wm
                d_w, d_s = dev(aw), dev(nbytes=aw.nbytes)
                Tw = synth.se3(synth.rot_zyx(rng.uniform(-np.pi, np.pi), 0.01, -0.01), rng.uniform(-10, 10, 3) * np.array([1, 1, 0.02])).astype(np.float64).copy()
                Tw0 = Tw.copy()
                was_pending = m.pending
                try:
                    r = v.align_end_reframe(other, Tw, d_w, len(wm), 16, d_s); ok = True
                    if not m.pending:
                        rep["failures"].append(dict(tag, error="worked, should have been refused: end_reframe"))
                except _lib.RgcError as e:
                    ok = False
                    if m.pending and "not finite" in str(e):   # the solve ended in NaN (a degenerate problem): its result has been handed over, the next
                        m.pending = False                      # target -- a map re-framed by a pose that is none -- is refused
                        rep["refusals_expected"] += 1
                        rep["nan_poses_refused"] = rep.get("nan_poses_refused", 0) + 1
                        if np.all(np.isfinite(v.getFinalTransformation() if hasattr(v, "_final") else np.zeros(1))):
                            pass
                    elif m.pending:
                        rep["failures"].append(dict(tag, error="refused, should have worked: end_reframe: %s" % (str(e)[:160],)))
                    else:
                        rep["refusals_expected"] += 1
                if ok:
                    m.pending = False
                    exp = fresh_solve(m, m.pending_guess)
                    rep["solves_compared"] += 1
                    if np.all(np.isfinite(exp[0])) and np.all(np.isfinite(r)):
                        dT = float(np.abs(r - exp[0]).max())
                        rep["max_dT"] = max(rep["max_dT"], dT)
                        rep["solves_bit_equal"] += int(np.array_equal(r, exp[0]))
                        if not dT <= (1e-6 if v.nr_iterations == exp[1] else 2e-4):
                            rep["failures"].append(dict(tag, error="end_reframe's solve differs from a fresh context's", dT=dT))
                        if not np.allclose(Tw, Tw0 @ r.astype(np.float64), rtol=0, atol=1e-9):
                            rep["failures"].append(dict(tag, error="end_reframe: the composed world pose"))
                        q, t = bench.world_to_body(Tw)
                        body = other.download(d_s, (len(wm), 4))
                        f = reg.odometer_vgicp(0); f.setResolution(m.res); f.setCorrespondenceRandomness(m.k); f.setRegularizationMethod(m.method); f.setVoxelAccumulationMode(m.mode)
                        f.transformCloudDevice(d_w, len(wm), 16, q, t, d_s2 := dev(nbytes=aw.nbytes))
                        same_body = np.array_equal(f.download(d_s2, (len(wm), 4)), body)
                        f.setInputTarget(body[:, :3].copy())
                        same_cov = np.array_equal(f.getTargetCovariances(), other.getTargetCovariances())
                        f.close()
                        if not (same_body and same_cov):
                            rep["failures"].append(dict(tag, error="end_reframe: the next context's target", same_body=bool(same_body), same_covariances=bool(same_cov)))
                elif not was_pending and not np.array_equal(Tw, Tw0):
                    rep["failures"].append(dict(tag, error="a refused end_reframe touched the world pose"))
            elif op == "set_cov":
                if m.pending or m.stale_settings or m.tgt is None:
                    continue
                c = v.getTargetCovariances()
                _, ok = expect(True, lambda: v.setTargetCovariances(c), "setTargetCovariances (its own)")
                if not ok and only is not None:
                    f = reg.odometer_vgicp(0); f.setInputTarget(m.tgt); full = f.getTargetCovariances(); f.close()
                    bad = np.nonzero(np.any((c != full).reshape(len(c), -1), axis=1))[0]
                    print("covariances that differ from a full build's:", len(bad), bad[:10], "\n", c[bad[0]] if len(bad) else "", "\n", full[bad[0]] if len(bad) else "", v.stats(), file=sys.stderr)
            if len(rep["failures"]) > 12:
                break
        if os.environ.get("FUZZ_DEBUG_END"):
            print("state at the end:", v.stats(), "lazy", m.lazy, file=sys.stderr, flush=True)
            v.setResolution(float(os.environ["FUZZ_DEBUG_END"]))
            print("after the setting (enqueued)", file=sys.stderr, flush=True)
            v.synchronize()
            print("synchronised:", v.stats(), file=sys.stderr, flush=True)
        if m.pending:
            v.align_end()
    except Exception as e:
        import traceback
        rep["failures"].append(dict(tag, error="exception: %r" % (e,), where=traceback.format_exc()[-500:]))
    for p_ in bufs:
        v.device_free(p_)
    v.close(); other.close()
    rep["trials"] += 1
    if len(rep["failures"]) > 12:
        break
rep["wall_s"] = round(time.time() - t0, 1)
print(json.dumps(rep))
