"""Independent numpy/scipy restatement of the loop-closure ICP (pcl::IterativeClosestPoint as configured
at RGC_mapping.cpp:2050-2069).  TEST INFRASTRUCTURE ONLY: it pins the C oracle's orc_icp_align.  cKDTree for the nearest
neighbours, numpy SVD for the rigid fit -- different machinery from the C code's grid and Jacobi.
PARITY UNPINNED: PCL is absent from /root/reference and from this image; pcl::IterativeClosestPoint's behaviour is restated from its published
algorithm; this is the builder's second restatement, not the reference's binary."""
import numpy as np
from scipy.spatial import cKDTree


def rigid_fit(p, q):
    """R, t minimising sum |R p + t - q|^2 (Umeyama without scale)"""
    cp, cq = p.mean(0), q.mean(0)
    H = (p - cp).T @ (q - cq)
    U, _, Vt = np.linalg.svd(H)
    D = np.diag([1.0, 1.0, np.sign(np.linalg.det(Vt.T @ U.T))])
    R = Vt.T @ D @ U.T
    return R, cq - R @ cp


def transform_f32(pts, T):
    T = T.astype(np.float32)
    x, y, z = pts[:, 0], pts[:, 1], pts[:, 2]
    return np.stack([((T[r, 0] * x + T[r, 1] * y) + T[r, 2] * z) + T[r, 3] for r in range(3)], axis=1).astype(np.float32)


def icp_align(src, tgt, max_corr_dist=10.0, max_iterations=100, transformation_eps=1e-6, fitness_eps=1e-6):
    tgt64 = tgt[:, :3].astype(np.float64)
    tree = cKDTree(tgt64)
    cur = src[:, :3].astype(np.float32).copy()
    fin = np.eye(4, dtype=np.float32)
    prev, it, state = np.inf, 0, "not_converged"
    while True:
        d, idx = tree.query(cur.astype(np.float64), k=1)
        keep = d * d <= max_corr_dist ** 2
        if keep.sum() < 3:
            state = "no_correspondences"
            break
        p, q = cur[keep].astype(np.float64), tgt64[idx[keep]]
        R, t = rigid_fit(p, q)
        T = np.eye(4, dtype=np.float32)
        T[:3, :3], T[:3, 3] = R.astype(np.float32), t.astype(np.float32)
        cur = transform_f32(cur, T)
        fin = (T @ fin).astype(np.float32)
        it += 1
        if it >= max_iterations:
            state = "iterations"; break
        cos_angle = 0.5 * (float(T[0, 0]) + float(T[1, 1]) + float(T[2, 2]) - 1.0)
        if cos_angle >= 1.0 - transformation_eps and float((T[:3, 3].astype(np.float64) ** 2).sum()) <= transformation_eps:
            state = "transform"; break
        mse = float((d[keep] ** 2).mean())
        if abs(mse - prev) < 1e-12:
            state = "abs_mse"; break
        if abs(mse - prev) / prev < fitness_eps:
            state = "rel_mse"; break
        prev = mse
    d, _ = tree.query(transform_f32(src[:, :3].astype(np.float32), fin).astype(np.float64), k=1)
    return fin, dict(iterations=it, state=state, fitness=float((d ** 2).mean()))
