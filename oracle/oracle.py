"""ctypes binding of the CPU oracle (oracle/librgc_oracle.so).

TEST INFRASTRUCTURE ONLY -- imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg; never by the product package.  PARITY UNPINNED (see rgc_oracle.h).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# RGC_ORACLE_ASAN=1 (tests/test_sanitizers.py, a child process with libasan preloaded): the AddressSanitizer / UBSan build of the same sources
_ASAN = os.environ.get("RGC_ORACLE_ASAN") == "1"
_LIB = os.path.join(_HERE, "librgc_oracle_asan.so" if _ASAN else "librgc_oracle.so")

DIRECT27, DIRECT7, DIRECT1 = 0, 1, 2


def build(force: bool = False) -> str:
    srcs = [os.path.join(_HERE, f) for f in ("rgc_oracle.c", "rgc_oracle_aux.c", "rgc_oracle_map.c", "rgc_oracle.h", "Makefile")]
    if force or not os.path.exists(_LIB) or any(os.path.getmtime(s) > os.path.getmtime(_LIB) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["asan"] if _ASAN else []))
    return _LIB


class Params(C.Structure):
    _fields_ = [("voxel_res", C.c_double), ("max_iterations", C.c_int), ("lm_max_iterations", C.c_int),
                ("rotation_eps", C.c_double), ("translation_eps", C.c_double), ("lm_init_lambda_factor", C.c_double),
                ("k_correspondences", C.c_int), ("neighbor_method", C.c_int), ("num_threads", C.c_int),
                ("regularization", C.c_int), ("voxel_mode", C.c_int)]


REG_NONE, REG_MIN_EIG, REG_NORMALIZED_MIN_EIG, REG_PLANE, REG_FROBENIUS = range(5)      # fast_gicp::RegularizationMethod, gicp_settings.hpp:6
VOXEL_ADDITIVE, VOXEL_ADDITIVE_WEIGHTED, VOXEL_MULTIPLICATIVE = range(3)                # VoxelAccumulationMode, gicp_settings.hpp:10


class FeParams(C.Structure):
    _fields_ = [("n_scans", C.c_int), ("min_range", C.c_double), ("max_range", C.c_double), ("use_intensity", C.c_int)]


class FeOut(C.Structure):
    _fields_ = [("cloud", C.POINTER(C.c_float)), ("n_cloud", C.c_int), ("ring_count", C.c_int * 64), ("scan_start", C.c_int * 64),
                ("scan_end", C.c_int * 64), ("curvature", C.POINTER(C.c_float)), ("curvature2", C.POINTER(C.c_float)),
                ("inten_curvature", C.POINTER(C.c_float)), ("label", C.POINTER(C.c_int)), ("inten_label", C.POINTER(C.c_int)),
                ("picked", C.POINTER(C.c_int)), ("ground_marked", C.POINTER(C.c_int)), ("sharp", C.POINTER(C.c_float)),
                ("flat", C.POINTER(C.c_float)), ("inten", C.POINTER(C.c_float)), ("feat_cap", C.c_int), ("n_sharp", C.c_int),
                ("n_sharp_own", C.c_int), ("n_flat", C.c_int), ("n_inten", C.c_int), ("ground_pts", C.POINTER(C.c_float)),
                ("ground_cap", C.c_int), ("n_ground", C.c_int), ("groundparam", C.c_double * 11), ("ground_valid", C.c_int)]


class EdgeFactor(C.Structure):
    _fields_ = [("valid", C.c_int), ("pad", C.c_int), ("a", C.c_double * 3), ("b", C.c_double * 3), ("var", C.c_double)]


class PlaneFactor(C.Structure):
    _fields_ = [("valid", C.c_int), ("pad", C.c_int), ("n", C.c_double * 3), ("d", C.c_double), ("var", C.c_double)]


class IcpParams(C.Structure):
    _fields_ = [("max_iterations", C.c_int), ("pad", C.c_int), ("max_corr_dist", C.c_double), ("transformation_eps", C.c_double),
                ("fitness_eps", C.c_double)]


class IcpResult(C.Structure):
    _fields_ = [("iterations", C.c_int), ("converged", C.c_int), ("state", C.c_int), ("n_correspondences", C.c_int), ("fitness", C.c_double)]


class MapregGround(C.Structure):
    _fields_ = [("last_v1", C.c_double * 3), ("last_v2", C.c_double * 3), ("last_norm", C.c_double * 3), ("last_distance", C.c_double),
                ("cur_norm", C.c_double * 3), ("cur_distance", C.c_double), ("q_history", C.c_double * 4), ("last_q", C.c_double * 4),
                ("last_t", C.c_double * 3), ("p_var", C.c_double)]


def make_ground(d):
    """dict(last_v1, last_v2, last_norm, last_distance, cur_norm, cur_distance, q_history, last_q, last_t, p_var) -> MapregGround"""
    if d is None:
        return None
    g = MapregGround()
    for k in ("last_v1", "last_v2", "last_norm", "cur_norm", "last_t"):
        setattr(g, k, (C.c_double * 3)(*[float(x) for x in d[k]]))
    for k in ("q_history", "last_q"):
        setattr(g, k, (C.c_double * 4)(*[float(x) for x in d[k]]))
    g.last_distance, g.cur_distance, g.p_var = float(d["last_distance"]), float(d["cur_distance"]), float(d.get("p_var", 0.2))
    return g


class MapregImu(C.Structure):
    _fields_ = [("delta_q", C.c_double * 4), ("imu_cov", C.c_double), ("pitch_cur", C.c_double), ("roll_cur", C.c_double),
                ("pitch_last", C.c_double), ("roll_last", C.c_double), ("pr_var", C.c_double)]


def make_imu(d):
    """dict(delta_q (x,y,z,w), imu_cov, pitch_cur, roll_cur, pitch_last, roll_last, pr_var=0.02) -> MapregImu"""
    if d is None:
        return None
    m = MapregImu()
    m.delta_q = (C.c_double * 4)(*[float(x) for x in d["delta_q"]])
    m.imu_cov, m.pr_var = float(d["imu_cov"]), float(d.get("pr_var", 0.02))
    m.pitch_cur, m.roll_cur, m.pitch_last, m.roll_last = (float(d[k]) for k in ("pitch_cur", "roll_cur", "pitch_last", "roll_last"))
    return m


class MapregTrace(C.Structure):
    _fields_ = [("initial_cost", C.c_double), ("final_cost", C.c_double), ("radius", C.c_double), ("iterations", C.c_int),
                ("successful", C.c_int), ("n_edge_cur", C.c_int), ("n_edge_last", C.c_int), ("n_plane_cur", C.c_int),
                ("n_plane_last", C.c_int), ("pad", C.c_int)]


class LmTrace(C.Structure):
    _fields_ = [("outer", C.c_int), ("inner", C.c_int), ("n_corr", C.c_int), ("y0", C.c_double), ("yi", C.c_double),
                ("rho", C.c_double), ("lambda_before", C.c_double), ("lambda_after", C.c_double), ("accepted", C.c_int),
                ("x", C.c_double * 16)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()  # (re)builds only when a source is newer than the library
        L = C.CDLL(_LIB)
        fp, ip, dp, vp = C.POINTER(C.c_float), C.POINTER(C.c_int), C.POINTER(C.c_double), C.c_void_p
        L.orc_default_params.argtypes = [C.POINTER(Params)]
        L.orc_knn.argtypes = [fp, C.c_int, C.c_int, C.c_int, ip, fp, C.c_int]
        L.orc_covariances.argtypes = [fp, C.c_int, C.c_int, C.c_int, dp, dp, C.c_int]
        L.orc_cov_from_neighbors.argtypes = [fp, C.c_int, ip, C.c_int, dp, dp]
        L.orc_eig3.argtypes = [dp, dp, dp]
        L.orc_voxelmap_create.argtypes = [fp, C.c_int, C.c_int, dp, C.c_double]
        L.orc_voxelmap_create.restype = vp
        L.orc_voxelmap_create_m.argtypes = [fp, C.c_int, C.c_int, dp, C.c_double, C.c_int]
        L.orc_voxelmap_create_m.restype = vp
        L.orc_covariances_m.argtypes = [fp, C.c_int, C.c_int, C.c_int, C.c_int, dp, C.c_int]
        L.orc_regularize.argtypes = [dp, C.c_int, dp]
        L.orc_regularize.restype = None
        L.orc_voxelmap_free.argtypes = [vp]
        L.orc_voxelmap_size.argtypes = [vp]
        L.orc_voxelmap_dump.argtypes = [vp, ip, ip, dp, dp]
        L.orc_voxel_coord.argtypes = [dp, C.c_double, ip]
        L.orc_reg_create.argtypes = [C.POINTER(Params)]
        L.orc_reg_create.restype = vp
        L.orc_reg_free.argtypes = [vp]
        L.orc_reg_set_target.argtypes = [vp, fp, C.c_int, C.c_int]
        L.orc_reg_set_source.argtypes = [vp, fp, C.c_int, C.c_int]
        L.orc_reg_prepare.argtypes = [vp]
        L.orc_reg_linearize.argtypes = [vp, dp, dp, dp]
        L.orc_reg_linearize.restype = C.c_double
        L.orc_reg_compute_error.argtypes = [vp, dp]
        L.orc_reg_compute_error.restype = C.c_double
        L.orc_reg_num_correspondences.argtypes = [vp]
        L.orc_reg_align.argtypes = [vp, fp, fp, dp, ip, ip, C.POINTER(LmTrace), C.c_int]
        L.orc_reg_num_linearize.argtypes = [vp]
        L.orc_reg_num_error.argtypes = [vp]
        L.orc_reg_fitness.argtypes = [vp, fp]
        L.orc_reg_fitness.restype = C.c_double
        L.orc_reg_source_cov.argtypes = [vp]
        L.orc_reg_source_cov.restype = dp
        L.orc_reg_target_cov.argtypes = [vp]
        L.orc_reg_target_cov.restype = dp
        L.orc_reg_voxelmap.argtypes = [vp]
        L.orc_reg_voxelmap.restype = vp
        L.orc_voxelgrid_filter.argtypes = [fp, C.c_int, C.c_float, fp]
        L.orc_deskew.argtypes = [fp, C.c_int, C.c_int, dp, dp]
        L.orc_transform_cloud.argtypes = [fp, C.c_int, C.c_int, dp, dp, fp]
        L.orc_fe_default_params.argtypes = [C.POINTER(FeParams)]
        L.orc_frontend.argtypes = [fp, C.c_int, C.c_int, C.POINTER(FeParams), C.POINTER(FeOut)]
        L.orc_so3_exp.argtypes = [dp, dp]
        L.orc_is_converged.argtypes = [dp, C.c_double, C.c_double]
        L.orc_transform_f32.argtypes = [fp, C.c_int, C.c_int, fp, fp]
        L.orc_icp_align.argtypes = [fp, C.c_int, C.c_int, fp, C.c_int, C.c_int, C.POINTER(IcpParams), fp, C.POINTER(IcpResult), C.c_int]
        L.orc_rigid_from_sums.argtypes = [C.c_double, dp, dp, dp, dp, dp]
        L.orc_knn_query.argtypes = [fp, C.c_int, C.c_int, fp, C.c_int, C.c_int, C.c_int, ip, fp, C.c_int]
        ef, pf, tr = C.POINTER(EdgeFactor), C.POINTER(PlaneFactor), C.POINTER(MapregTrace)
        L.orc_mapreg_associate_edges.argtypes = [fp, C.c_int, dp, dp, fp, C.c_int, C.c_int, ef, C.c_int]
        L.orc_mapreg_associate_planes.argtypes = [fp, C.c_int, dp, dp, fp, C.c_int, C.c_int, pf, C.c_int]
        gp = C.POINTER(MapregGround)
        mi = C.POINTER(MapregImu)
        L.orc_mapreg_solve.argtypes = [fp, ef, C.c_int, fp, pf, C.c_int, fp, ef, C.c_int, fp, pf, C.c_int, gp, gp, mi, dp, C.c_int, tr]
        L.orc_mapreg_optimize.argtypes = [fp, C.c_int, fp, C.c_int, fp, C.c_int, fp, C.c_int, fp, C.c_int, fp, C.c_int, C.c_int, gp, gp, mi, dp, tr, C.c_int]
        _lib = L
    return _lib


def _f32(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(C.POINTER(C.c_float))


def _f64(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


def default_params(**kw) -> Params:
    p = Params()
    lib().orc_default_params(C.byref(p))
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def knn(xyz, k=20, threads=0):
    a, ap = _f32(xyz)
    n = a.shape[0]
    idx = np.empty((n, k), np.int32)
    d2 = np.empty((n, k), np.float32)
    rc = lib().orc_knn(ap, n, a.shape[1], k, _ip(idx), d2.ctypes.data_as(C.POINTER(C.c_float)), threads)
    if rc:
        raise RuntimeError(f"orc_knn rc={rc}")
    return idx, d2


def covariances(xyz, k=20, threads=0):
    a, ap = _f32(xyz)
    n = a.shape[0]
    cov = np.empty((n, 3, 3), np.float64)
    nrm = np.empty((n, 3), np.float64)
    rc = lib().orc_covariances(ap, n, a.shape[1], k, cov.ctypes.data_as(C.POINTER(C.c_double)),
                               nrm.ctypes.data_as(C.POINTER(C.c_double)), threads)
    if rc:
        raise RuntimeError(f"orc_covariances rc={rc}")
    return cov, nrm


def covariances_m(xyz, method, k=20, threads=0):
    """fast_gicp_impl.hpp:241-298 under any RegularizationMethod (REG_*): (n, 3, 3)"""
    a, ap = _f32(xyz)
    n = a.shape[0]
    cov = np.empty((n, 3, 3), np.float64)
    rc = lib().orc_covariances_m(ap, n, a.shape[1], k, int(method), cov.ctypes.data_as(C.POINTER(C.c_double)), threads)
    if rc:
        raise RuntimeError(f"orc_covariances_m rc={rc}")
    return cov


def regularize(S, method):
    s, sp = _f64(np.asarray(S, np.float64).reshape(9))
    out = np.empty((3, 3))
    lib().orc_regularize(sp, int(method), out.ctypes.data_as(C.POINTER(C.c_double)))
    return out


def eig3(A):
    a, ap = _f64(A)
    ev = np.empty(3)
    V = np.empty((3, 3))
    lib().orc_eig3(ap, ev.ctypes.data_as(C.POINTER(C.c_double)), V.ctypes.data_as(C.POINTER(C.c_double)))
    return ev, V


def _dump_vm(handle):
    L = lib()
    V = L.orc_voxelmap_size(handle)
    coords = np.empty((V, 3), np.int32)
    num = np.empty(V, np.int32)
    mean = np.empty((V, 3))
    cov = np.empty((V, 3, 3))
    L.orc_voxelmap_dump(handle, _ip(coords), _ip(num), mean.ctypes.data_as(C.POINTER(C.c_double)),
                        cov.ctypes.data_as(C.POINTER(C.c_double)))
    return dict(coords=coords, num=num, mean=mean, cov=cov)


def voxelmap(xyz, cov, res=1.0, multiplicative=False):
    a, ap = _f32(xyz)
    c, cp = _f64(cov)
    h = lib().orc_voxelmap_create_m(ap, a.shape[0], a.shape[1], cp, res, 1 if multiplicative else 0)
    try:
        return _dump_vm(h)
    finally:
        lib().orc_voxelmap_free(h)


def voxel_coord(x, res=1.0):
    a, ap = _f64(x)
    c = np.empty(3, np.int32)
    lib().orc_voxel_coord(ap, res, _ip(c))
    return c


def voxelgrid_filter(xyzi, leaf):
    a, ap = _f32(xyzi)
    assert a.shape[1] == 4
    out = np.empty_like(a)
    m = lib().orc_voxelgrid_filter(ap, a.shape[0], leaf, out.ctypes.data_as(C.POINTER(C.c_float)))
    if m < 0:
        return a.copy()
    return out[:m].copy()


def so3_exp(w):
    a, ap = _f64(w)
    q = np.empty(4)
    lib().orc_so3_exp(ap, q.ctypes.data_as(C.POINTER(C.c_double)))
    return q


def transform_f32(xyz, T):
    a, ap = _f32(xyz)
    t, tp = _f32(np.asarray(T).reshape(16))
    out = np.empty((a.shape[0], 3), np.float32)
    lib().orc_transform_f32(ap, a.shape[0], a.shape[1], tp, out.ctypes.data_as(C.POINTER(C.c_float)))
    return out


def deskew(xyzi, q_xyzw, t):
    a = np.array(xyzi, dtype=np.float32, order="C", copy=True)
    q, qp = _f64(q_xyzw)
    tt, tp = _f64(t)
    lib().orc_deskew(a.ctypes.data_as(C.POINTER(C.c_float)), a.shape[0], a.shape[1], qp, tp)
    return a


def transform_cloud(xyzi, q_xyzw, t):
    a, ap = _f32(xyzi)
    q, qp = _f64(q_xyzw)
    tt, tp = _f64(t)
    out = np.empty((a.shape[0], 4), np.float32)
    lib().orc_transform_cloud(ap, a.shape[0], a.shape[1], qp, tp, out.ctypes.data_as(C.POINTER(C.c_float)))
    return out


def frontend(xyzi, n_scans=16, min_range=0.5, max_range=80.0, use_intensity=1):
    """ScanRegistration::laserCloudHandler restatement; returns a dict of numpy arrays."""
    a, ap = _f32(xyzi)
    n = a.shape[0]
    prm = FeParams(n_scans, min_range, max_range, use_intensity)
    fcap, gcap = max(n, 1), max(10 * n, 1)
    bufs = dict(cloud=np.zeros((n, 4), np.float32), curvature=np.zeros(n, np.float32), curvature2=np.zeros(n, np.float32),
                inten_curvature=np.zeros(n, np.float32), label=np.zeros(n, np.int32), inten_label=np.zeros(n, np.int32),
                picked=np.zeros(n, np.int32), ground_marked=np.zeros(n, np.int32), sharp=np.zeros((fcap, 5), np.float32),
                flat=np.zeros((fcap, 5), np.float32), inten=np.zeros((fcap, 5), np.float32), ground_pts=np.zeros((gcap, 4), np.float32))
    o = FeOut()
    for k, v in bufs.items():
        setattr(o, k, v.ctypes.data_as(C.POINTER(C.c_int if v.dtype == np.int32 else C.c_float)))
    o.feat_cap, o.ground_cap = fcap, gcap
    rc = lib().orc_frontend(ap, n, a.shape[1], C.byref(prm), C.byref(o))
    if rc:
        raise RuntimeError(f"orc_frontend rc={rc}")
    m = o.n_cloud
    out = {k: (v[:m] if k in ("cloud", "curvature", "curvature2", "inten_curvature", "label", "inten_label", "picked", "ground_marked") else v)
           for k, v in bufs.items()}
    out["sharp"], out["flat"], out["inten"] = bufs["sharp"][:o.n_sharp], bufs["flat"][:o.n_flat], bufs["inten"][:o.n_inten]
    out["ground_pts"] = bufs["ground_pts"][:o.n_ground]
    out.update(n_cloud=m, n_sharp_own=o.n_sharp_own, ring_count=np.array(o.ring_count[:n_scans]), scan_start=np.array(o.scan_start[:n_scans]),
               scan_end=np.array(o.scan_end[:n_scans]), groundparam=np.array(o.groundparam[:]), ground_valid=bool(o.ground_valid))
    return out


def knn_query(xyz, queries, k, threads=0):
    """exact kNN of arbitrary queries (pcl::KdTreeFLANN::nearestKSearch restated): (idx (nq,k), d2 (nq,k) float32)"""
    a, ap = _f32(xyz)
    b, bp = _f32(queries)
    idx = np.empty((b.shape[0], k), np.int32)
    d2 = np.empty((b.shape[0], k), np.float32)
    rc = lib().orc_knn_query(ap, a.shape[0], a.shape[1], bp, b.shape[0], b.shape[1], k, idx.ctypes.data_as(C.POINTER(C.c_int)),
                             d2.ctypes.data_as(C.POINTER(C.c_float)), threads)
    if rc:
        raise RuntimeError(f"orc_knn_query rc={rc}")
    return idx, d2


def _factors_to_np(arr, kind):
    n = len(arr)
    valid = np.array([f.valid for f in arr], dtype=bool)
    if kind == "edge":
        return dict(valid=valid, a=np.array([list(f.a) for f in arr]).reshape(n, 3), b=np.array([list(f.b) for f in arr]).reshape(n, 3),
                    var=np.array([f.var for f in arr]))
    return dict(valid=valid, n=np.array([list(f.n) for f in arr]).reshape(n, 3), d=np.array([f.d for f in arr]),
                var=np.array([f.var for f in arr]))


def mapreg_associate(feat_xyzw, q_xyzw, t, map_xyz, kind, threads=0, raw=False):
    """RGC_mapping.cpp:1092-1138 (kind='edge') / :1191-1236 (kind='plane'): per-feature factor parameters."""
    f, fp_ = _f32(feat_xyzw)
    m, mp = _f32(map_xyz)
    q, qp = _f64(q_xyzw)
    tt, tp = _f64(t)
    arr = ((EdgeFactor if kind == "edge" else PlaneFactor) * max(f.shape[0], 1))()
    fn = lib().orc_mapreg_associate_edges if kind == "edge" else lib().orc_mapreg_associate_planes
    cnt = fn(fp_, f.shape[0], qp, tp, mp, m.shape[0], m.shape[1], arr, threads)
    if cnt < 0:
        raise RuntimeError(f"orc_mapreg_associate rc={cnt}")
    return arr if raw else _factors_to_np(arr[: f.shape[0]], kind)


def mapreg_solve(corner_cur, e_cur, surf_cur, p_cur, corner_last, e_last, surf_last, p_last, poses14, max_iterations=6, ground_cur=None,
                 ground_last=None, imu=None):
    """ceres::Solve restated (RGC_mapping.cpp:1333-1341); e_*/p_* are the RAW ctypes factor arrays of mapreg_associate(raw=True)."""
    cc, ccp = _f32(corner_cur); sc, scp = _f32(surf_cur); cl, clp = _f32(corner_last); sl, slp = _f32(surf_last)
    x = np.ascontiguousarray(poses14, dtype=np.float64).copy()
    tr = MapregTrace()
    gc, gl, im = make_ground(ground_cur), make_ground(ground_last), make_imu(imu)
    lib().orc_mapreg_solve(ccp, e_cur, cc.shape[0], scp, p_cur, sc.shape[0], clp, e_last, cl.shape[0], slp, p_last, sl.shape[0],
                           C.byref(gc) if gc else None, C.byref(gl) if gl else None, C.byref(im) if im else None,
                           x.ctypes.data_as(C.POINTER(C.c_double)), max_iterations, C.byref(tr))
    return x, {k: getattr(tr, k) for k, _ in MapregTrace._fields_ if k != "pad"}


def mapreg_optimize(corner_cur, surf_cur, corner_last, surf_last, corner_map, surf_map, poses14, threads=0, ground_cur=None, ground_last=None,
                    imu=None):
    """The optimisation block of one mapping frame (2 x associate + solve, then quaternion normalisation)."""
    cc, ccp = _f32(corner_cur); sc, scp = _f32(surf_cur); cl, clp = _f32(corner_last); sl, slp = _f32(surf_last)
    cm, cmp_ = _f32(corner_map); sm, smp = _f32(surf_map)
    assert cm.shape[1] == sm.shape[1]
    x = np.ascontiguousarray(poses14, dtype=np.float64).copy()
    tr = (MapregTrace * 2)()
    gc, gl, im = make_ground(ground_cur), make_ground(ground_last), make_imu(imu)
    rc = lib().orc_mapreg_optimize(ccp, cc.shape[0], scp, sc.shape[0], clp, cl.shape[0], slp, sl.shape[0], cmp_, cm.shape[0], smp, sm.shape[0],
                                   cm.shape[1], C.byref(gc) if gc else None, C.byref(gl) if gl else None, C.byref(im) if im else None,
                                   x.ctypes.data_as(C.POINTER(C.c_double)), tr, threads)
    if rc < 0:
        raise RuntimeError(f"orc_mapreg_optimize rc={rc}")
    return x, rc, [{k: getattr(t, k) for k, _ in MapregTrace._fields_ if k != "pad"} for t in tr]


def icp_align(source, target, max_corr_dist=10.0, max_iterations=100, transformation_eps=1e-6, fitness_eps=1e-6, threads=0):
    """pcl::IterativeClosestPoint as configured at RGC_mapping.cpp:2050-2069 -> (final_T (4,4) float32, dict)"""
    s, sp = _f32(source)
    g, gp = _f32(target)
    prm = IcpParams(max_iterations, 0, max_corr_dist, transformation_eps, fitness_eps)
    T = np.zeros(16, np.float32)
    res = IcpResult()
    rc = lib().orc_icp_align(sp, s.shape[0], s.shape[1], gp, g.shape[0], g.shape[1], C.byref(prm), T.ctypes.data_as(C.POINTER(C.c_float)),
                             C.byref(res), threads)
    if rc:
        raise RuntimeError(f"orc_icp_align rc={rc}")
    return T.reshape(4, 4), {k: getattr(res, k) for k, _ in IcpResult._fields_}


class Registration:
    """Mirror of the FastVGICP call sequence at RGC_odometer.cpp:998-1011, CPU oracle backend."""

    def __init__(self, **params):
        self.params = default_params(**params)
        self._h = lib().orc_reg_create(C.byref(self.params))
        self._keep = {}
        self.final_T = np.eye(4, dtype=np.float32)
        self.final_H = np.eye(6)
        self.converged = False
        self.lm_failed = False
        self.iterations = 0
        self.trace = []

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_reg_free(self._h)
            self._h = None

    def set_target(self, xyz):
        a, ap = _f32(xyz)
        lib().orc_reg_set_target(self._h, ap, a.shape[0], a.shape[1])

    def set_source(self, xyz):
        a, ap = _f32(xyz)
        lib().orc_reg_set_source(self._h, ap, a.shape[0], a.shape[1])

    def prepare(self):
        rc = lib().orc_reg_prepare(self._h)
        if rc:
            raise RuntimeError(f"orc_reg_prepare rc={rc}")

    def linearize(self, T, want_H=True):
        t, tp = _f64(np.asarray(T, dtype=np.float64).reshape(16))
        if want_H:
            H = np.empty((6, 6))
            b = np.empty(6)
            cost = lib().orc_reg_linearize(self._h, tp, H.ctypes.data_as(C.POINTER(C.c_double)),
                                           b.ctypes.data_as(C.POINTER(C.c_double)))
            return cost, H, b
        return lib().orc_reg_linearize(self._h, tp, None, None), None, None

    def compute_error(self, T):
        t, tp = _f64(np.asarray(T, dtype=np.float64).reshape(16))
        return lib().orc_reg_compute_error(self._h, tp)

    @property
    def num_correspondences(self):
        return lib().orc_reg_num_correspondences(self._h)

    def align(self, guess=None, max_trace=64):
        g = np.eye(4, dtype=np.float32) if guess is None else np.asarray(guess, dtype=np.float32)
        g, gp = _f32(g.reshape(16))
        fin = np.empty(16, np.float32)
        H = np.empty((6, 6))
        conv, fail = C.c_int(0), C.c_int(0)
        tr = (LmTrace * max_trace)()
        it = lib().orc_reg_align(self._h, gp, fin.ctypes.data_as(C.POINTER(C.c_float)),
                                 H.ctypes.data_as(C.POINTER(C.c_double)), C.byref(conv), C.byref(fail), tr, max_trace)
        if it < 0:
            raise RuntimeError("orc_reg_align failed (need >= k points in both clouds)")
        self.final_T = fin.reshape(4, 4)
        self.final_H = H
        self.converged, self.lm_failed, self.iterations = bool(conv.value), bool(fail.value), it
        self.n_linearize = lib().orc_reg_num_linearize(self._h)
        self.n_error = lib().orc_reg_num_error(self._h)
        self.trace = [dict(outer=t.outer, inner=t.inner, n_corr=t.n_corr, y0=t.y0, yi=t.yi, rho=t.rho,
                           lambda_before=t.lambda_before, lambda_after=t.lambda_after, accepted=t.accepted,
                           x=np.array(t.x[:]).reshape(4, 4)) for t in tr[:min(it, max_trace)]]
        return self.final_T

    def fitness(self, T=None):
        t = self.final_T if T is None else np.asarray(T, dtype=np.float32)
        t, tp = _f32(t.reshape(16))
        return lib().orc_reg_fitness(self._h, tp)

    def source_cov(self, n):
        p = lib().orc_reg_source_cov(self._h)
        return np.ctypeslib.as_array(p, shape=(n, 3, 3)).copy()

    def target_cov(self, n):
        p = lib().orc_reg_target_cov(self._h)
        return np.ctypeslib.as_array(p, shape=(n, 3, 3)).copy()

    def voxelmap(self):
        return _dump_vm(lib().orc_reg_voxelmap(self._h))
