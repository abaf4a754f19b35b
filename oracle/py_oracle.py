"""Independent numpy/scipy restatement of the registration path, used ONLY to generate and check the
golden fixtures under tests/golden/ (SURVEY.md §4 item 1, §8c).

TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED: the reference has no tests/fixtures and cannot be built
here; this file is a second, deliberately literal restatement (4x4 homogeneous matrices exactly as the
reference writes them, scipy cKDTree for the exact kNN, numpy SVD for the regularisation) so that the
C oracle (rgc_oracle.c, which uses the collapsed 3x3 forms of SURVEY Appendix A.1) is cross-checked by
code that shares nothing with it.

File:line citations are relative to /root/reference/rgc_slam/.
"""
from __future__ import annotations

import math

import numpy as np
from scipy.spatial import cKDTree

K = 20


def knn_indices(xyz: np.ndarray, k: int = K) -> np.ndarray:
    """exact kNN (self included) -- stands in for pcl::search::KdTree::nearestKSearch, fast_gicp_impl.hpp:254"""
    tree = cKDTree(xyz.astype(np.float64))
    _, idx = tree.query(xyz.astype(np.float64), k=k)
    return idx


def covariances(xyz: np.ndarray, k: int = K, regularization: str = "PLANE"):
    """fast_gicp_impl.hpp:241-298 under RegularizationMethod `regularization` (the odometer's: PLANE): returns (n,4,4) Matrix4d-like covs."""
    idx = knn_indices(xyz, k)
    n = xyz.shape[0]
    covs = np.zeros((n, 4, 4))
    for i in range(n):
        nb = np.ones((4, k))
        nb[:3, :] = xyz[idx[i]].astype(np.float64).T          # :256-259 getVector4fMap().cast<double>()
        nb = nb - nb.mean(axis=1, keepdims=True)               # :261
        cov = nb @ nb.T / k                                    # :262
        if regularization == "NONE":                           # :264-265
            covs[i] = cov
            continue
        if regularization == "FROBENIUS":                      # :266-271
            Cm = cov[:3, :3] + 1e-3 * np.eye(3)
            Ci = np.linalg.inv(Cm)
            covs[i, :3, :3] = np.linalg.inv(Ci / np.linalg.norm(Ci))
            continue
        U, s, Vt = np.linalg.svd(cov[:3, :3])                  # :273 JacobiSVD
        if regularization == "PLANE":
            values = np.array([1.0, 1.0, 1e-3])                # :280-282
        elif regularization == "MIN_EIG":
            values = np.maximum(s, 1e-3)                       # :283-285
        elif regularization == "NORMALIZED_MIN_EIG":
            values = np.maximum(s / s.max(), 1e-3)             # :286-289
        else:
            raise ValueError(regularization)
        covs[i, :3, :3] = U @ np.diag(values) @ Vt             # :293
    return covs, idx


def voxel_coord(x, res=1.0):
    """fast_vgicp_voxel.hpp:158-160"""
    return np.floor(np.asarray(x, dtype=np.float64)[:3] / res - 0.5).astype(np.int64)


def build_voxelmap(xyz: np.ndarray, covs: np.ndarray, res: float = 1.0, voxel_mode: str = "ADDITIVE"):
    """fast_vgicp_voxel.hpp:129-156; AdditiveGaussianVoxel :105-122 (ADDITIVE and ADDITIVE_WEIGHTED alike, :137-141),
    MultiplicativeGaussianVoxel :76-99"""
    vox = {}
    mult = voxel_mode == "MULTIPLICATIVE"
    for i in range(xyz.shape[0]):
        m = np.array([xyz[i, 0], xyz[i, 1], xyz[i, 2], 1.0], dtype=np.float64)
        c = tuple(voxel_coord(m, res))
        v = vox.get(c)
        if v is None:
            v = vox[c] = dict(n=0, mean=np.zeros(4), cov=np.zeros((4, 4)))
        v["n"] += 1
        if mult:                                               # :82-91
            ci = covs[i].copy()
            ci[3, 3] = 1.0
            ci = np.linalg.inv(ci)
            v["cov"] += ci
            v["mean"] += ci @ m
        else:
            v["mean"] += m
            v["cov"] += covs[i]
    for v in vox.values():
        if mult:                                               # :93-99
            v["cov"][3, 3] = 1.0
            v["mean"][3] = 1.0
            v["cov"] = np.linalg.inv(v["cov"])
            v["mean"] = v["cov"] @ v["mean"]
        else:
            v["mean"] /= v["n"]
            v["cov"] /= v["n"]
    return vox


def skewd(x):
    """so3/so3.hpp:21-31"""
    return np.array([[0, -x[2], x[1]], [x[2], 0, -x[0]], [-x[1], x[0], 0]], dtype=np.float64)


def so3_exp(w):
    """so3/so3.hpp:58-77 -> rotation matrix of the unit quaternion"""
    w = np.asarray(w, dtype=np.float64)
    th2 = float(w @ w)
    if th2 < 1e-10:
        th4 = th2 * th2
        imag = 0.5 - th2 / 48.0 + th4 / 3840.0
        real = 1.0 - th2 / 8.0 + th4 / 384.0
    else:
        th = math.sqrt(th2)
        imag = math.sin(0.5 * th) / th
        real = math.cos(0.5 * th)
    qw, qx, qy, qz = real, imag * w[0], imag * w[1], imag * w[2]
    return np.array([
        [1 - 2 * (qy * qy + qz * qz), 2 * (qx * qy - qz * qw), 2 * (qx * qz + qy * qw)],
        [2 * (qx * qy + qz * qw), 1 - 2 * (qx * qx + qz * qz), 2 * (qy * qz - qx * qw)],
        [2 * (qx * qz - qy * qw), 2 * (qy * qz + qx * qw), 1 - 2 * (qx * qx + qy * qy)]])


OFFSETS = {
    "DIRECT1": [(0, 0, 0)],
    "DIRECT7": [(0, 0, 0), (1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0), (0, 0, 1), (0, 0, -1)],
    "DIRECT27": [(i - 1, j - 1, k - 1) for i in range(3) for j in range(3) for k in range(3)],
}


class VGICP:
    """FastVGICP as driven at RGC_odometer.cpp:998-1009."""

    def __init__(self, res=1.0, max_iterations=25, rot_eps=2e-3, trans_eps=1e-6, method="DIRECT1", regularization="PLANE", voxel_mode="ADDITIVE"):
        self.res, self.max_iterations, self.rot_eps, self.trans_eps = res, max_iterations, rot_eps, trans_eps
        self.lm_max_iterations, self.lm_init_lambda_factor = 10, 1e-9
        self.method = method
        self.regularization, self.voxel_mode = regularization, voxel_mode
        self.trace = []

    def set_target(self, xyz):
        self.tgt = np.asarray(xyz, dtype=np.float32)
        self.tgt_covs, self.tgt_knn = covariances(self.tgt, regularization=self.regularization)
        self.vox = build_voxelmap(self.tgt, self.tgt_covs, self.res, self.voxel_mode)

    def set_source(self, xyz):
        self.src = np.asarray(xyz, dtype=np.float32)
        self.src_covs, self.src_knn = covariances(self.src, regularization=self.regularization)

    def update_correspondences(self, T):
        """fast_vgicp_impl.hpp:73-116"""
        self.corr = []
        for i in range(self.src.shape[0]):
            mA = np.array([*self.src[i].astype(np.float64), 1.0])
            c = voxel_coord(T @ mA, self.res)
            for o in OFFSETS[self.method]:
                v = self.vox.get((c[0] + o[0], c[1] + o[1], c[2] + o[2]))
                if v is not None:
                    RCR = v["cov"] + T @ self.src_covs[i] @ T.T
                    RCR[3, 3] = 1.0
                    M = np.linalg.inv(RCR)
                    M[3, 3] = 0.0
                    self.corr.append((i, v, M))

    def linearize(self, T, want=True):
        """fast_vgicp_impl.hpp:119-180"""
        self.update_correspondences(T)
        H, b, s = np.zeros((6, 6)), np.zeros(6), 0.0
        for i, v, M in self.corr:
            mA = np.array([*self.src[i].astype(np.float64), 1.0])
            tA = T @ mA
            e = v["mean"] - tA
            w = math.sqrt(v["n"])
            s += w * float(e @ M @ e)
            if want:
                J = np.zeros((4, 6))
                J[:3, :3] = skewd(tA[:3])
                J[:3, 3:] = -np.eye(3)
                H += w * J.T @ M @ J
                b += w * J.T @ M @ e
        return s, H, b

    def compute_error(self, T):
        """fast_vgicp_impl.hpp:183-204"""
        s = 0.0
        for i, v, M in self.corr:
            tA = T @ np.array([*self.src[i].astype(np.float64), 1.0])
            e = v["mean"] - tA
            s += math.sqrt(v["n"]) * float(e @ M @ e)
        return s

    def is_converged(self, delta):
        """lsq_registration_impl.hpp:82-91"""
        R = np.abs(delta[:3, :3] - np.eye(3)) / self.rot_eps
        t = np.abs(delta[:3, 3]) / self.trans_eps
        return max(R.max(), t.max()) < 1

    def align(self, guess):
        """lsq_registration_impl.hpp:53-79,125-172"""
        x0 = np.asarray(guess, dtype=np.float32).astype(np.float64).copy()
        x0[3] = [0, 0, 0, 1]
        lam, conv, self.trace = -1.0, False, []
        for it in range(self.max_iterations):
            if conv:
                break
            y0, H, b = self.linearize(x0)
            if lam < 0:
                lam = self.lm_init_lambda_factor * np.abs(np.diag(H)).max()
            nu, ok = 2.0, False
            rec = dict(outer=it, y0=y0, lambda_before=lam, n_corr=len(self.corr), H=H.copy(), b=b.copy())
            for k in range(self.lm_max_iterations):
                d = np.linalg.solve(H + lam * np.eye(6), -b)
                delta = np.eye(4)
                delta[:3, :3] = so3_exp(d[:3])
                delta[:3, 3] = d[3:]
                xi = delta @ x0
                yi = self.compute_error(xi)
                rho = (y0 - yi) / float(d @ (lam * d - b))
                rec.update(inner=k + 1, yi=yi, rho=rho)
                if rho < 0:
                    if self.is_converged(delta):
                        ok = True
                        rec["accepted"] = 0
                        break
                    lam, nu = nu * lam, 2 * nu
                    continue
                x0 = xi
                lam = lam * max(1.0 / 3.0, 1 - (2 * rho - 1) ** 3)
                ok = True
                rec["accepted"] = 1
                break
            rec.update(lambda_after=lam, x=x0.copy())
            self.trace.append(rec)
            if not ok:
                break
            conv = self.is_converged(delta)
        self.converged = conv
        self.final = x0.astype(np.float32)
        return self.final

    def fitness(self, T=None):
        """pcl::Registration::getFitnessScore [3P-memory], SURVEY A.6"""
        T = (self.final if T is None else np.asarray(T)).astype(np.float32)
        p = self.src
        q = np.empty_like(p)
        for r in range(3):
            q[:, r] = ((T[r, 0] * p[:, 0] + T[r, 1] * p[:, 1]) + T[r, 2] * p[:, 2]) + T[r, 3]
        tree = cKDTree(self.tgt.astype(np.float64))
        _, j = tree.query(q.astype(np.float64), k=1)
        dd = q - self.tgt[j]
        d2 = (dd[:, 0] * dd[:, 0] + dd[:, 1] * dd[:, 1]) + dd[:, 2] * dd[:, 2]   # float32 like L2_Simple
        return float(np.sum(d2.astype(np.float64)) / len(d2))


def voxelgrid_filter(xyzi: np.ndarray, leaf: float) -> np.ndarray:
    """pcl::VoxelGrid<PointXYZI>::filter [3P-memory], SURVEY A.6 (float32 arithmetic)."""
    p = np.asarray(xyzi, dtype=np.float32)
    inv = np.float32(1.0) / np.float32(leaf)
    mn = np.floor(p[:, :3].min(axis=0) * inv).astype(np.int64)
    mx = np.floor(p[:, :3].max(axis=0) * inv).astype(np.int64)
    div = mx - mn + 1
    ijk = np.floor(p[:, :3] * inv).astype(np.int64) - mn
    idx = ijk[:, 0] + ijk[:, 1] * div[0] + ijk[:, 2] * div[0] * div[1]
    order = np.argsort(idx, kind="stable")
    out = []
    s = 0
    ids = idx[order]
    while s < len(ids):
        e = s
        acc = np.zeros(4, np.float32)
        while e < len(ids) and ids[e] == ids[s]:
            acc = acc + p[order[e]]
            e += 1
        out.append(acc / np.float32(e - s))
        s = e
    return np.asarray(out, dtype=np.float32)
