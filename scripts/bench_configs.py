"""one line per configuration of BASELINE.json on one MI355X -- registered scans/s, algorithmic bytes per
scan, achieved algorithmic GB/s and its share of 8 TB/s, kernel breakdown, pose parity against the CPU oracle on the
first frame (and the oracle's own rate there).  c-main is bench.py's workload; c1 its 100 k-point-map plumbing case; c3 HDL-64 130 k
vs 5 M; c5 two interleaved HDL-64 patterns 250 k vs 20 M with the IMU-like rotation prior.  Extra: c-main with S independent
sequences sharing ONE GPU (one context + one host thread each) -- the headroom a single latency-bound sequence leaves.

    python scripts/bench_configs.py [c1 c-main c3 c5 multi]
"""
import sys, os, time, json, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa: F401
import rgc_slam_amd.synth as synth
from rgc_slam_amd import registration
from oracle import oracle

which = sys.argv[1:] or ["c1", "c-main", "c3", "c5", "multi"]


def algorithmic_bytes(n_s, n_t, n_vox, n_corr, n_lin, n_err):
    return 36.0 * n_s + 36.0 * n_t + (36.0 * n_t + 40.0 * n_vox) + (n_lin + n_err) * (36.0 * n_s + 40.0 * n_corr) + 24.0 * n_s


def rot_angle(Ra, Rb):
    R = Ra.astype(np.float64) @ Rb.astype(np.float64).T
    w = 0.5 * np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    return float(np.arcsin(min(1.0, np.linalg.norm(w))))


def to_dev(v, xyz):
    a = np.zeros((xyz.shape[0], 4), np.float32)
    a[:, :3] = xyz
    p = v.device_alloc(a.nbytes)
    v.upload(p, a)
    return p


def run_config(name, tgt, scans, guesses_true, frames, prior=None):
    """scans[i] registered to tgt, guess = previous result (or the corrupted truth when prior is given)"""
    v = registration.odometer_vgicp(0)
    d_tgt = to_dev(v, tgt)
    d_s = [to_dev(v, s) for s in scans]
    W = 2

    def step(i, g):
        v.setInputTargetDevice(d_tgt, len(tgt), 16)
        v.setInputSourceDevice(d_s[i], len(scans[i]), 16)
        v.align(g, want_output=False, want_fitness=True)
        return v.getFinalTransformation()
    g = guesses_true[0]
    for i in range(W):
        g = step(i, prior[i] if prior else g)
    v.synchronize()
    stats, finals, gin = [], [], []
    t0 = time.perf_counter()
    for i in range(W, W + frames):
        gi = prior[i] if prior else g
        gin.append(gi)
        g = step(i, gi)
        finals.append(g)
        st = v.stats()
        stats.append((st["outer_iterations"], st["n_linearize"], st["n_error"], st["n_corr"], st["n_voxels"]))
    v.synchronize()
    el = time.perf_counter() - t0
    v.profile_enable(True); v.profile_select(None); v.profile_reset()
    for j in range(min(frames, 3)):
        step(W + j, gin[j])
    v.synchronize()
    prof = {k: round(x["total_ms"] / min(frames, 3), 4) for k, x in v.profile().items() if x["total_ms"] > 0}
    v.profile_enable(False)
    pf = np.asarray(stats, float).mean(axis=0)
    B = algorithmic_bytes(len(scans[0]), len(tgt), pf[4], pf[3], pf[1], pf[2])
    sps = frames / el
    out = {"config": name, "n_source": int(len(scans[0])), "n_target": int(len(tgt)), "frames": frames, "scans_per_s": round(sps, 2), "ms_per_scan": round(1e3 * el / frames, 3),
           "mean_outer_iterations": round(pf[0], 2), "n_voxels": int(pf[4]), "algorithmic_MB_per_scan": round(B / 1e6, 1),
           "achieved_algorithmic_GBps": round(B * sps / 1e9, 1), "pct_of_8TBps": round(100 * B * sps / 8e12, 3), "kernel_ms_per_scan": prof}
    # CPU oracle on the first timed frame: its rate on this host and the pose parity of that frame
    cores = os.cpu_count() or 1
    o = oracle.Registration(num_threads=cores)
    c0 = time.perf_counter()
    o.set_target(tgt); o.set_source(scans[W])
    To = o.align(gin[0]); _ = o.fitness()
    tc = time.perf_counter() - c0
    out["cpu_oracle_scans_per_s"] = round(1 / tc, 4)
    out["cpu_cores"] = cores
    out["max_dt_m"] = float(np.abs(finals[0][:3, 3] - To[:3, 3]).max())
    out["max_dtheta_rad"] = rot_angle(finals[0][:3, :3], To[:3, :3])
    v.close()
    return out


res = []
if "c1" in which or "c-main" in which or "multi" in which:
    pass
for name, n_t, n_s, frames in (("c1 (30 k vs 100 k)", 100000, 30000, 20), ("c-main (30 k vs 1 M)", 1000000, 30000, 20)):
    if name.split()[0] not in which:
        continue
    world, tgt = synth.make_world_and_map(n_t, seed=synth.SEED)
    poses = synth.make_trajectory(frames + 3, seed=synth.SEED)
    scans = [synth.make_scan_n(world, poses[i + 1], n_s, seed=synth.SEED + 100 + i)["xyz"] for i in range(frames + 2)]
    res.append(run_config(name, tgt, scans, [poses[0].astype(np.float32)], frames))
    print(json.dumps(res[-1]), flush=True)

if "multi" in which:
    # S independent sequences on ONE GPU: one context and one host thread each (ctypes releases the GIL inside the library)
    world, tgt = synth.make_world_and_map(1000000, seed=synth.SEED)
    frames = 20
    poses = synth.make_trajectory(frames + 3, seed=synth.SEED)
    scans = [synth.make_scan_n(world, poses[i + 1], 30000, seed=synth.SEED + 100 + i)["xyz"] for i in range(frames + 2)]
    for S in (1, 2, 4, 8):
        ctxs = []
        for s in range(S):
            v = registration.odometer_vgicp(0)
            ctxs.append((v, to_dev(v, tgt), [to_dev(v, x) for x in scans]))
        finals = [None] * S

        def seq(s, lo, hi, g0):
            v, d_tgt, d_s = ctxs[s]
            g = g0
            for i in range(lo, hi):
                v.setInputTargetDevice(d_tgt, len(tgt), 16)
                v.setInputSourceDevice(d_s[i], 30000, 16)
                v.align(g, want_output=False, want_fitness=True)
                g = v.getFinalTransformation()
            v.synchronize()
            finals[s] = g
        for s in range(S):
            seq(s, 0, 2, poses[0].astype(np.float32))
        g0 = [finals[s] for s in range(S)]
        th = [threading.Thread(target=seq, args=(s, 2, 2 + frames, g0[s])) for s in range(S)]
        t0 = time.perf_counter()
        for t in th: t.start()
        for t in th: t.join()
        el = time.perf_counter() - t0
        same = all(np.array_equal(finals[0], f) for f in finals)
        res.append({"config": f"c-main x {S} sequences on one GPU", "scans_per_s": round(S * frames / el, 1), "ms_per_scan_per_sequence": round(1e3 * el / frames, 3),
                    "all_sequences_same_result": bool(same)})
        print(json.dumps(res[-1]), flush=True)
        for v, _, _ in ctxs:
            v.close()

if "c3" in which or "c5" in which:
    world, tile = synth.make_world_and_map(5_000_000, seed=synth.SEED + 7)
    e = synth.hdl64_elev()
    if "c3" in which:
        frames = 8
        poses = synth.make_trajectory(frames + 3, seed=synth.SEED + 7)
        scans = [synth.make_scan_n(world, poses[i + 1], 130000, elev_deg=e, seed=synth.SEED + 200 + i)["xyz"] for i in range(frames + 2)]
        res.append(run_config("c3 (HDL-64 130 k vs 5 M)", tile, scans, [poses[0].astype(np.float32)], frames))
        print(json.dumps(res[-1]), flush=True)
    if "c5" in which:
        frames = 4
        L = 2.0 * world.half_extent + 4.0
        tgt = np.concatenate([tile, tile + np.float32([L, 0, 0]), tile + np.float32([0, L, 0]), tile + np.float32([L, L, 0])]).astype(np.float32)
        poses = synth.make_trajectory(frames + 3, seed=synth.SEED + 9)
        rng = np.random.default_rng(11)
        scans, prior = [], []
        for i in range(frames + 2):
            a = synth.make_scan_n(world, poses[i + 1], 125000, elev_deg=e, seed=synth.SEED + 300 + i)["xyz"]
            b = synth.make_scan_n(world, poses[i + 1], 125000, elev_deg=e + 0.5 * float(np.abs(np.diff(np.sort(e))).min()), seed=synth.SEED + 400 + i)["xyz"]
            scans.append(np.concatenate([a, b]).astype(np.float32))
            ang = np.deg2rad(0.5) * rng.standard_normal(3)      # IMU-preintegrated prior: the true pose corrupted by ~0.5 deg of rotation
            prior.append((poses[i + 1] @ synth.se3(synth.rot_zyx(*ang), [0, 0, 0])).astype(np.float32))
        res.append(run_config("c5 (250 k vs 20 M, rotation prior)", tgt, scans, [poses[0].astype(np.float32)], frames, prior=prior))
        print(json.dumps(res[-1]), flush=True)
