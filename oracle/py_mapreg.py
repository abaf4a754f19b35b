"""Independent numpy/scipy restatement of the mapping node's feature registration (SURVEY.md §8f row f1;
RGC_mapping.cpp:1069-1358, lidarFactor.hpp:9-51,91-121).  TEST INFRASTRUCTURE ONLY: it pins oracle/rgc_oracle_map.c.

Deliberately different machinery from the C oracle: cKDTree for the 5-NN, numpy eigh / lstsq for the line test and the
plane fit, finite-difference Jacobians for the normal equations, scipy BFGS on the robust cost for the optimum.
PARITY UNPINNED: the reference holds no vectors for this path and cannot be built here (Ceres, PCL absent); this is the builder's second
restatement, not the reference's binary."""
import numpy as np
from scipy.spatial import cKDTree

HUBER_A = 0.1


def quat_rot(q, p):
    x, y, z, w = q
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                  [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                  [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    return p @ R.T


def associate(feat, q, t, map_xyz, kind):
    sel = (quat_rot(np.asarray(q, float), feat[:, :3].astype(np.float64)) + np.asarray(t, float)).astype(np.float32)
    tree = cKDTree(map_xyz[:, :3].astype(np.float64))
    d, idx = tree.query(sel.astype(np.float64), k=5)
    n = len(feat)
    out = dict(valid=np.zeros(n, bool), var=feat[:, 3].astype(np.float64))
    if kind == "edge":
        out.update(a=np.zeros((n, 3)), b=np.zeros((n, 3)))
    else:
        out.update(n=np.zeros((n, 3)), d=np.zeros(n))
    for i in range(n):
        P = map_xyz[idx[i], :3].astype(np.float64)
        if kind == "edge":
            if not d[i, 4] ** 2 < 1.0:
                continue
            c = P.mean(0)
            Z = P - c
            w, V = np.linalg.eigh(Z.T @ Z)
            if not w[2] > 3 * w[1]:
                continue
            out["a"][i], out["b"][i] = c + 0.1 * V[:, 2], c - 0.1 * V[:, 2]
            out["valid"][i] = True
        else:
            if not d[i, 4] ** 2 < 2.0:
                continue
            nrm = np.linalg.lstsq(P, -np.ones(5), rcond=None)[0]
            nn = np.linalg.norm(nrm)
            dd = 1.0 / nn
            nrm = nrm / nn
            if np.any(np.abs(P @ nrm + dd) > 0.2):
                continue
            out["n"][i], out["d"][i] = nrm, dd
            out["valid"][i] = True
    return out


def quat_plus(q, d):
    nd = np.linalg.norm(d)
    dq = np.concatenate([np.sin(nd) / nd * d, [np.cos(nd)]]) if nd > 0 else np.concatenate([d, [1.0]])
    ax, ay, az, aw = dq
    bx, by, bz, bw = q
    return np.array([aw * bx + ax * bw + ay * bz - az * by, aw * by - ax * bz + ay * bw + az * bx,
                     aw * bz + ax * by - ay * bx + az * bw, aw * bw - ax * bx - ay * by - az * bz])


def residual_blocks(cfeat, ef, sfeat, pf, q, t):
    """(edge residuals (ne,3), plane residuals (np,1)) of one pose"""
    e, p = np.zeros((0, 3)), np.zeros((0, 1))
    if len(cfeat):
        lp = quat_rot(q, cfeat[:, :3].astype(np.float64)) + t
        v = ef["valid"]
        nu = np.cross(lp[v] - ef["a"][v], lp[v] - ef["b"][v])
        de = np.linalg.norm(ef["a"][v] - ef["b"][v], axis=1)
        e = nu / de[:, None] * ef["var"][v][:, None]
    if len(sfeat):
        pw = quat_rot(q, sfeat[:, :3].astype(np.float64)) + t
        v = pf["valid"]
        p = ((np.sum(pf["n"][v] * pw[v], axis=1) + pf["d"][v]) * pf["var"][v])[:, None]
    return e, p


def _qmul(a, b):
    return np.array([a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1], a[3] * b[1] - a[0] * b[2] + a[1] * b[3] + a[2] * b[0],
                     a[3] * b[2] + a[0] * b[1] - a[1] * b[0] + a[2] * b[3], a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2]])


def ground_residual(G, q, t):
    """Ground_DeltaFactor_goable (lidarFactor.hpp:357-391); G = dict like oracle.make_ground's input"""
    lqc = np.array([-G["last_q"][0], -G["last_q"][1], -G["last_q"][2], G["last_q"][3]])
    q_lc = _qmul(lqc, np.asarray(q, float))
    t_lc = quat_rot(lqc, np.asarray(t, float) - np.asarray(G["last_t"], float))
    gn = quat_rot(q_lc, np.asarray(G["cur_norm"], float))
    delta_t = quat_rot(np.asarray(G["q_history"], float), t_lc)
    pv = G.get("p_var", 0.2)
    return np.array([(G["last_distance"] - (G["cur_distance"] + delta_t[2])) / (pv / 1000),
                     abs(np.dot(G["last_v1"], gn)) / (pv * 10), abs(np.dot(G["last_v2"], gn)) / (pv * 10)])


def pitch_roll(q):
    """Quaternion2EulerAngle (lidarFactor.hpp:405-433), q = x,y,z,w"""
    x, y, z, w = q
    sinp = 2 * (w * y - x * z)
    return (np.pi / 2 if sinp >= 1 else (-np.pi / 2 if sinp <= -1 else np.arcsin(sinp))), np.arctan2(2 * (w * x + y * z), 1 - 2 * (x * x + y * y))


def imu_residual(I, qc, ql):
    """RelativeRFactor on (q_last, q_cur) + PitchRollFactor on each (lidarFactor.hpp:174-226, 434-468); I = dict like oracle.make_imu's input"""
    qli = np.array([-ql[0], -ql[1], -ql[2], ql[3]])
    rqi = np.array([-I["delta_q"][0], -I["delta_q"][1], -I["delta_q"][2], I["delta_q"][3]])
    e = _qmul(rqi, _qmul(qli, np.asarray(qc, float)))
    pv = I.get("pr_var", 0.02)
    pc, rc = pitch_roll(qc)
    pl, rl = pitch_roll(ql)
    return np.concatenate([2 * e[:3] / I["imu_cov"], [2 * (pc - I["pitch_cur"]) / pv, 2 * (rc - I["roll_cur"]) / pv,
                                                      2 * (pl - I["pitch_last"]) / pv, 2 * (rl - I["roll_last"]) / pv]])


def _sq(blocks):
    return np.concatenate([np.sum(b * b, axis=1) for b in blocks])


def robust_cost(blocks):
    s = _sq(blocks)
    return float(0.5 * np.sum(np.where(s <= HUBER_A ** 2, s, 2 * HUBER_A * np.sqrt(np.maximum(s, 1e-300)) - HUBER_A ** 2)))


def _weights(blocks):
    """sqrt(rho') per scalar residual row (the Corrector's scaling), rows in the order of _flat"""
    out = []
    for b in blocks:
        s = np.sum(b * b, axis=1)
        w = np.where(s <= HUBER_A ** 2, 1.0, np.sqrt(HUBER_A / np.sqrt(np.maximum(s, 1e-300))))
        out.append(np.repeat(w, b.shape[1]))
    return np.concatenate(out)


def _flat(blocks):
    return np.concatenate([b.reshape(-1) for b in blocks])


def lm_solve(sets, poses14, max_iterations=6, imu=None):
    """Ceres-style LM (the loop restated in rgc_oracle_map.c) with finite-difference Jacobians; sets = [(cfeat, ef, sfeat, pf)] x 2."""
    x = np.array(poses14, float)
    def pose(xv, b):
        return (xv[0:4], xv[4:7]) if b == 0 else (xv[7:11], xv[11:14])
    def apply(xv, d12):
        o = xv.copy()
        for b in range(2):
            q, t = pose(xv, b)
            o[7 * b: 7 * b + 4] = quat_plus(q, d12[6 * b: 6 * b + 3])
            o[7 * b + 4: 7 * b + 7] = t + d12[6 * b + 3: 6 * b + 6]
        return o
    def blocks_at(xv, b):
        q, t = pose(xv, b)
        return residual_blocks(sets[b][0], sets[b][1], sets[b][2], sets[b][3], q, t)
    def ground_of(b):
        return sets[b][4] if len(sets[b]) > 4 else None
    def ground_r(xv, b):
        q, t = pose(xv, b)
        return ground_residual(ground_of(b), q, t) if ground_of(b) is not None else np.zeros(0)
    def imu_r(xv):
        return imu_residual(imu, xv[0:4], xv[7:11]) if imu is not None else np.zeros(0)
    def cost_at(xv):
        return (robust_cost(blocks_at(xv, 0)) + robust_cost(blocks_at(xv, 1)) + 0.5 * sum(float(ground_r(xv, b) @ ground_r(xv, b)) for b in range(2))
                + 0.5 * float(imu_r(xv) @ imu_r(xv)))
    def normal_eq(xv):
        H, g = np.zeros((12, 12)), np.zeros(12)
        for b in range(2):
            base = blocks_at(xv, b)
            w = _weights(base)  # frozen at xv (Corrector); only the raw residuals are differentiated
            r0 = _flat(base)
            J = np.zeros((len(r0), 6))
            h = 1e-6
            for a in range(6):
                d = np.zeros(12); d[6 * b + a] = h
                J[:, a] = (_flat(blocks_at(apply(xv, d), b)) - _flat(blocks_at(apply(xv, -d), b))) / (2 * h)
            Jw, rw = J * w[:, None], r0 * w
            if ground_of(b) is not None:  # NULL loss: no robustification
                rg = ground_r(xv, b)
                Jg = np.zeros((3, 6))
                for a in range(6):
                    d = np.zeros(12); d[6 * b + a] = h
                    Jg[:, a] = (ground_r(apply(xv, d), b) - ground_r(apply(xv, -d), b)) / (2 * h)
                Jw, rw = np.vstack([Jw, Jg]), np.concatenate([rw, rg])
            H[6 * b: 6 * b + 6, 6 * b: 6 * b + 6] = Jw.T @ Jw
            g[6 * b: 6 * b + 6] = Jw.T @ rw
        if imu is not None:  # couples the two rotations: a full 12-column Jacobian
            ri = imu_r(xv)
            Ji = np.zeros((7, 12))
            for a in range(12):
                d = np.zeros(12); d[a] = 1e-6
                Ji[:, a] = (imu_r(apply(xv, d)) - imu_r(apply(xv, -d))) / 2e-6
            H += Ji.T @ Ji
            g += Ji.T @ ri
        return H, g
    radius, dec = 1e4, 2.0
    cost = cost_at(x)
    H, g = normal_eq(x)
    trace = dict(initial_cost=cost, successful=0, iterations=0)
    it = 0
    while it < max_iterations:
        if np.abs(g).max() <= 1e-10:
            break
        D = np.clip(np.diag(H), 1e-6, 1e32)
        d = np.linalg.solve(H + np.diag(D) / radius, -g)
        model = -d @ (g + 0.5 * H @ d)
        rho, newc, xn = -1.0, cost, x
        if model > 0:
            xn = apply(x, d)
            newc = cost_at(xn)
            rho = (cost - newc) / model
        it += 1
        if rho > 1e-3:
            old = cost
            x = xn
            radius = min(radius / max(1.0 / 3.0, 1.0 - (2 * rho - 1) ** 3), 1e16)
            dec = 2.0
            trace["successful"] += 1
            cost = cost_at(x)
            H, g = normal_eq(x)
            if abs(old - cost) <= 1e-6 * old or np.linalg.norm(d) <= 1e-8 * (np.linalg.norm(x) + 1e-8):
                break
        else:
            radius /= dec
            dec *= 2.0
    trace.update(final_cost=cost, iterations=it)
    return x, trace


def total_cost(sets, x):
    return sum(robust_cost(residual_blocks(st[0], st[1], st[2], st[3], x[7 * b: 7 * b + 4], x[7 * b + 4: 7 * b + 7])) for b, st in enumerate(sets))


def tangent_apply(x0, z):
    x = np.array(x0, float)
    for b in range(2):
        x[7 * b: 7 * b + 4] = quat_plus(x0[7 * b: 7 * b + 4], z[6 * b: 6 * b + 3])
        x[7 * b + 4: 7 * b + 7] = x0[7 * b + 4: 7 * b + 7] + z[6 * b + 3: 6 * b + 6]
    return x


def polish(sets, x, maxiter=200):
    """a generic quasi-Newton minimiser (scipy BFGS, finite-difference gradient) on the same robust cost, started at x"""
    from scipy.optimize import minimize
    sol = minimize(lambda z: total_cost(sets, tangent_apply(x, z)), np.zeros(12), method="BFGS", options=dict(maxiter=maxiter, gtol=1e-9))
    return tangent_apply(x, sol.x), float(sol.fun)


def fd_gradient(sets, x, h=1e-6):
    g = np.zeros(12)
    for a in range(12):
        d = np.zeros(12); d[a] = h
        g[a] = (total_cost(sets, tangent_apply(x, d)) - total_cost(sets, tangent_apply(x, -d))) / (2 * h)
    return g
