"""Randomised campaign of the frame body's scalar host stages (csrc/rgc_host.cpp: B7 pose fusion, B8 composition, C9 extraction, B1 gyro
pre-integration, the attitude filter, the ground gate, R2ypr / ypr2R) against the literal numpy / scipy restatement oracle/py_fusion.py.
No GPU: runs anywhere the library loads.      python tests/fuzz/fuzz_host.py [trials] [seed]"""
import sys, os, json, time, math, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from rgc_slam_amd import _lib
from oracle import py_fusion as pf

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 500
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
h = _lib.load()
dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
rep = {"trials": 0, "failures": [], "max": {"fuse_q": 0.0, "fuse_t": 0.0, "compose": 0.0, "ypr": 0.0, "extract": 0.0, "preintegrate": 0.0, "filter": 0.0, "gate": 0.0}}


def rq(rng, angle):
    w = rng.normal(size=3); w *= angle / max(np.linalg.norm(w), 1e-300)
    n = np.linalg.norm(w)
    return np.array([*(math.sin(n / 2) / max(n, 1e-300) * w), math.cos(n / 2)]) if n > 0 else np.array([0, 0, 0, 1.0])


def note(k, v):
    rep["max"][k] = max(rep["max"][k], float(v))


t0 = time.time()
for trial in range(trials):
    rng = np.random.default_rng(seed0 * 15487469 + trial)
    tag = {"trial": trial}
    try:
        # ---- B7: the fusion solve, every combination of factors, tilts up to a few degrees, fitness over three decades ----
        use_ground, use_imu = bool(rng.random() < 0.6), bool(rng.random() < 0.5)
        n_last = np.array([rng.normal(0, 0.03), rng.normal(0, 0.03), 1.0]); n_last /= np.linalg.norm(n_last)
        v1 = np.cross(n_last, [1, 0, 0]); v1 /= np.linalg.norm(v1); v2 = np.cross(n_last, v1)
        n_cur = n_last + rng.normal(0, rng.choice([0.002, 0.01, 0.03]), 3); n_cur /= np.linalg.norm(n_cur)
        c = dict(q_lidar=rq(rng, rng.choice([0.001, 0.03, 0.2])), t_lidar=rng.normal(0, rng.choice([0.01, 0.1, 1.0]), 3), fitness=float(10 ** rng.uniform(-3, 0)),
                 use_ground=use_ground, ground_last=np.array([*n_last, *v1, *v2, 0.56, 0.02]), ground_cur=np.array([*n_cur, *v1, *v2, 0.56 + rng.normal(0, 0.02), abs(rng.normal(0, 0.05))]),
                 q_w_curr_f=rq(rng, rng.choice([0.01, 0.1, 1.0])), ground_cov=float(rng.choice([0.05, 0.2, 1.0])), use_imu=use_imu, q_imu=rq(rng, rng.choice([0.002, 0.05])))
        fin = _lib.FuseIn(); h.rgc_default_fuse_in(C.byref(fin))
        fin.q_lidar_xyzw[:] = list(c["q_lidar"]); fin.t_lidar[:] = list(c["t_lidar"]); fin.fitness = c["fitness"]
        fin.use_ground = int(use_ground); fin.ground_last[:] = list(c["ground_last"]); fin.ground_cur[:] = list(c["ground_cur"])
        fin.q_w_curr_f_xyzw[:] = list(c["q_w_curr_f"]); fin.ground_cov = c["ground_cov"]; fin.use_imu = int(use_imu); fin.q_imu_xyzw[:] = list(c["q_imu"])
        q, t, it = np.empty(4), np.empty(3), C.c_int(0)
        rc = h.rgc_fuse_pose(C.byref(fin), dp(q), dp(t), C.byref(it))
        qo, to = pf.fuse(c)
        if rc != 0:
            rep["failures"].append(dict(tag, stage="fuse", error="status %d" % rc))
        else:
            if np.dot(q, qo) < 0: qo = -qo
            if it.value < 5:
                note("fuse_q", np.abs(q - qo).max()); note("fuse_t", np.abs(t - to).max())
            # (the reference's solver options stop the library's solve after six iterations, RGC_odometer.cpp:1121-1127; the restatement solves to
            #  convergence: where the cap bites the two are a few micrometres apart)
            #  on inputs this hard -- tilts of degrees, weights over three decades -- the two ends lie up to millimetres apart along the directions the
            #  factors barely constrain: compared where the library's solve stopped BEFORE the cap, counted where it did not)
            capped = it.value >= 5
            rep["fusions_stopped_by_the_iteration_cap"] = rep.get("fusions_stopped_by_the_iteration_cap", 0) + int(capped)
            if not capped and not (np.abs(q - qo).max() < 1e-5 and np.abs(t - to).max() < 1e-5):
                rep["failures"].append(dict(tag, stage="fuse", dq=float(np.abs(q - qo).max()), dt=float(np.abs(t - to).max()), use_ground=use_ground, use_imu=use_imu, iterations=it.value))
        # ---- B8 composition, ypr ----
        q_w, q_f = rq(rng, rng.uniform(0, 3.1)), rq(rng, 0.03)
        t_w, t_f, t_l = rng.normal(0, 50, 3), rng.normal(0, 0.1, 3), rng.normal(0, 0.1, 3)
        R_imu = np.ascontiguousarray(pf.q2R(rq(rng, rng.uniform(0, 3.1))))
        ui = int(rng.random() < 0.5)
        qo_, to_, tl_ = np.empty(4), np.empty(3), np.empty(3)
        h.rgc_compose_pose(dp(q_w), dp(t_w), dp(q_f), dp(t_f), dp(t_l), ui, dp(R_imu), dp(qo_), dp(to_), dp(tl_))
        Re, te, tle = pf.compose(q_w, t_w, q_f, t_f, t_l, ui, R_imu)
        e = max(np.abs(pf.q2R(qo_) - Re).max(), np.abs(to_ - te).max() / 50, np.abs(tl_ - tle).max())
        note("compose", e)
        if not e < 1e-11: rep["failures"].append(dict(tag, stage="compose", err=float(e)))
        R = np.ascontiguousarray(pf.q2R(rq(rng, rng.uniform(0.0, 3.1))))
        ypr, R2 = np.empty(3), np.empty(9)
        h.rgc_R2ypr(dp(R), dp(ypr)); h.rgc_ypr2R(dp(ypr), dp(R2))
        e = max(np.abs(ypr - pf.R2ypr(R)).max(), np.abs(R2.reshape(3, 3) - R).max())
        note("ypr", e)
        if not e < 1e-10: rep["failures"].append(dict(tag, stage="ypr", err=float(e), ypr=ypr.tolist()))
        # ---- C9 extraction ----
        qq = rq(rng, rng.uniform(1e-4, 3.0))
        T = np.eye(4, dtype=np.float32); T[:3, :3] = pf.q2R(qq).astype(np.float32); T[:3, 3] = rng.normal(0, 10, 3).astype(np.float32)
        qx, tx = np.empty(4), np.empty(3)
        h.rgc_extract_pose(T.ctypes.data_as(C.POINTER(C.c_float)), dp(qx), dp(tx))
        if np.dot(qx, qq) < 0: qx = -qx
        e = np.abs(qx - qq).max(); note("extract", e)
        if not (e < 5e-7 and np.array_equal(tx, T[:3, 3].astype(np.float64))): rep["failures"].append(dict(tag, stage="extract", err=float(e)))
        # ---- B1 pre-integration ----
        n = int(rng.integers(2, 60))
        stamps = np.ascontiguousarray(100.0 + np.cumsum(rng.uniform(0.001, 0.01, n)))
        gyr = np.ascontiguousarray(rng.normal(0, rng.choice([0.05, 0.5, 3.0]), (n, 3)))
        prev, cur = 100.0, float(stamps[-1] - rng.uniform(0, 0.004))
        dq = np.empty(4)
        h.rgc_imu_preintegrate(dp(stamps), dp(gyr), None, n, prev, cur, dp(dq), None, None, None)
        e = np.abs(dq - pf.imu_delta_q(stamps, gyr, prev, cur)).max(); note("preintegrate", e)
        if not e < 1e-12: rep["failures"].append(dict(tag, stage="preintegrate", err=float(e)))
        # ---- the attitude filter on a random stream (first 100 dropped, fast phase, then steady) ----
        if trial % 10 == 0:
            f = _lib.ImuFilter(); h.rgc_imu_filter_init(C.byref(f)); ref = pf.ImuFilter()
            Rt = pf.ypr2R(np.array([rng.uniform(-180, 180), rng.uniform(-10, 10), rng.uniform(-10, 10)]))
            tt, worst = 0.0, 0.0
            for j in range(700):
                tt += float(rng.uniform(0.003, 0.007))
                a = np.ascontiguousarray(Rt.T @ np.array([0, 0, 9.81]) + np.array(f.ba[:]) + rng.normal(0, 0.05, 3))
                g = np.ascontiguousarray(np.array(f.bg[:]) + rng.normal(0, 0.02, 3))
                rc = h.rgc_imu_filter_push(C.byref(f), tt, dp(a), dp(g), None, None)
                r = ref.push(tt, a, g)
                if (rc == 1) != (r is not None):
                    rep["failures"].append(dict(tag, stage="filter", error="accepted differently at message %d" % j)); break
                if rc == 1: worst = max(worst, float(np.abs(np.array(f.Rwi[:]).reshape(3, 3) - ref.Rwi).max()))
            note("filter", worst)
            if not worst < 1e-11: rep["failures"].append(dict(tag, stage="filter", err=worst))
        # ---- the ground gate over a random drive ----
        if trial % 10 == 5:
            g_ = _lib.GroundGate(); h.rgc_ground_gate_init(C.byref(g_)); ref = pf.GroundGate()
            h.rgc_ground_gate_remember(C.byref(g_)); ref.remember()
            q_wc = np.array([0, 0, 0, 1.0]); pitch = 0.0

            def plane(p):
                nn = pf.ypr2R(np.array([0.0, p, 0.0])) @ np.array([0, 0, 1.0])
                a1 = np.cross(nn, [0, 1.0, 0]); a1 /= np.linalg.norm(a1)
                return np.array([*nn, *a1, *np.cross(nn, a1), 0.56, 0.01])
            worst = 0.0
            for kf in range(120):
                change = rng.random() < 0.06
                new_pitch = float(rng.choice([0.0, 4.0, 7.0, -5.0])) if change else pitch
                gl, gc = plane(pitch), plane(new_pitch)
                d = math.radians((new_pitch - pitch) / 2)
                dq_imu = rq(rng, 0.0005) if not change else np.array([0, math.sin(d / 2 * 0.2), 0, math.cos(d / 2 * 0.2)])
                q_l, t_l2 = rq(rng, 0.002), np.array([0.1, 0.0, 0.0])
                q_wc = pf.qmul(q_wc, pf.qmul(np.array([0, math.sin(d / 2), 0, math.cos(d / 2)]), q_l)); q_wc /= np.linalg.norm(q_wc)
                qf = np.empty(4)
                rc = h.rgc_ground_gate_step(C.byref(g_), dp(gl), dp(gc), dp(q_l), dp(t_l2), dp(dq_imu), dp(q_wc), dp(qf))
                rf, qfr = ref.step(gl, gc, q_l, t_l2, dq_imu, q_wc)
                if rc != rf:
                    rep["failures"].append(dict(tag, stage="gate", error="flag differs at frame %d" % kf, hip=int(rc), ref=int(rf))); break
                worst = max(worst, float(np.abs(qf - qfr).max()))
                pitch = new_pitch
            note("gate", worst)
            if not worst < 1e-12: rep["failures"].append(dict(tag, stage="gate", err=worst))
            if g_.n_history != len(ref.history): rep["failures"].append(dict(tag, stage="gate", error="history length", hip=int(g_.n_history), ref=len(ref.history)))
    except Exception as e:
        import traceback
        rep["failures"].append(dict(tag, error="exception: %r" % (e,), where=traceback.format_exc()[-400:]))
    rep["trials"] += 1
    if len(rep["failures"]) > 15:
        break
rep["wall_s"] = round(time.time() - t0, 1)
print(json.dumps(rep))
