"""C oracle (oracle/rgc_oracle.c) vs the golden fixtures produced by the independent numpy/scipy
restatement (tests/golden/gen_golden.py).  CPU only."""
import numpy as np

from conftest import tri6


def test_knn_sets(orc, fx_reg):
    idx, d2 = orc.knn(fx_reg["src"], 20)
    got = np.sort(idx, axis=1)
    bad = np.sum(np.any(got != fx_reg["src_knn"], axis=1))
    assert bad <= 2, f"{bad} source neighbourhoods differ (only float-vs-double near ties may)"
    assert np.all(np.diff(d2, axis=1) >= 0)
    assert np.all(d2[:, 0] == 0) and np.all(idx[:, 0] == np.arange(len(idx)))
    idx_t, _ = orc.knn(fx_reg["tgt"], 20)
    bad = np.sum(np.any(np.sort(idx_t[::8], axis=1) != fx_reg["tgt_knn_sub"], axis=1))
    assert bad <= 2


def test_covariances(orc, fx_reg):
    cov, nrm = orc.covariances(fx_reg["src"], 20)
    err = np.abs(tri6(cov) - fx_reg["src_cov6"]).max(axis=1)
    assert np.sum(err > 1e-9) <= 2, err.max()
    # PLANE regularisation: C = I - 0.999 n n^T, eigenvalues {1, 1, 1e-3}
    ev = np.linalg.eigvalsh(cov)
    assert np.allclose(ev, [1e-3, 1.0, 1.0], atol=1e-12)
    rebuilt = np.eye(3)[None] - 0.999 * nrm[:, :, None] * nrm[:, None, :]
    assert np.abs(rebuilt - cov).max() < 1e-12
    cov_t, _ = orc.covariances(fx_reg["tgt"], 20)
    err = np.abs(tri6(cov_t[::4]) - fx_reg["tgt_cov6_sub"]).max(axis=1)
    assert np.sum(err > 1e-9) <= 2


def test_eig3_known(orc):
    A = np.diag([3.0, 1.0, 2.0])
    ev, V = orc.eig3(A)
    assert np.allclose(ev, [3, 2, 1])
    assert np.allclose(np.abs(V), [[1, 0, 0], [0, 0, 1], [0, 1, 0]])
    rng = np.random.default_rng(5)
    for _ in range(50):
        B = rng.normal(size=(3, 3))
        S = B @ B.T
        ev, V = orc.eig3(S)
        assert np.allclose(V @ np.diag(ev) @ V.T, S, atol=1e-12)
        assert np.allclose(np.sort(ev), np.linalg.eigvalsh(S), atol=1e-12)


def test_voxelmap(orc, fx_reg):
    cov, _ = orc.covariances(fx_reg["tgt"], 20)
    vm = orc.voxelmap(fx_reg["tgt"], cov, 1.0)
    assert np.array_equal(vm["coords"], fx_reg["vox_coords"])
    assert np.array_equal(vm["num"], fx_reg["vox_num"])
    assert np.abs(vm["mean"] - fx_reg["vox_mean"]).max() < 1e-12
    assert np.abs(tri6(vm["cov"]) - fx_reg["vox_cov6"]).max() < 1e-9
    assert vm["num"].sum() == len(fx_reg["tgt"])


def test_voxel_coord_table(orc, fx_small):
    for x, c1, c05 in zip(fx_small["vc_x"], fx_small["vc_res1"], fx_small["vc_res05"]):
        assert np.array_equal(orc.voxel_coord(x, 1.0), c1)
        assert np.array_equal(orc.voxel_coord(x, 0.5), c05)
    # the -0.5 offset of fast_vgicp_voxel.hpp:158-160
    assert list(orc.voxel_coord([0.5, -0.5, -1.5], 1.0)) == [0, -1, -2]
    assert list(orc.voxel_coord([1.4999, 0.4999, -0.5001], 1.0)) == [0, -1, -2]


def test_so3_exp(orc, fx_small):
    for w, R in zip(fx_small["so3_w"], fx_small["so3_R"]):
        q = orc.so3_exp(w)
        qw, qx, qy, qz = q
        Rq = np.array([[1 - 2 * (qy * qy + qz * qz), 2 * (qx * qy - qz * qw), 2 * (qx * qz + qy * qw)],
                       [2 * (qx * qy + qz * qw), 1 - 2 * (qx * qx + qz * qz), 2 * (qy * qz - qx * qw)],
                       [2 * (qx * qz - qy * qw), 2 * (qy * qz + qx * qw), 1 - 2 * (qx * qx + qy * qy)]])
        assert np.abs(Rq - R).max() < 1e-14


def _reg(orc, fx, **kw):
    r = orc.Registration(num_threads=4, **kw)
    r.set_target(fx["tgt"])
    r.set_source(fx["src"])
    return r


def test_linearize_direct1(orc, fx_reg):
    r = _reg(orc, fx_reg)
    cost, H, b = r.linearize(fx_reg["guess"])
    assert r.num_correspondences == int(fx_reg["lin_ncorr"])
    assert abs(cost - fx_reg["lin_cost"]) <= 1e-9 * abs(fx_reg["lin_cost"])
    assert np.abs(H - fx_reg["lin_H"]).max() <= 1e-9 * np.abs(fx_reg["lin_H"]).max()
    assert np.abs(b - fx_reg["lin_b"]).max() <= 1e-9 * np.abs(fx_reg["lin_b"]).max()
    assert np.allclose(H, H.T, rtol=0, atol=1e-9 * np.abs(H).max())
    assert np.linalg.eigvalsh(H).min() > 0
    # frozen correspondences (fast_vgicp_impl.hpp:183-204)
    e = r.compute_error(fx_reg["err_T"])
    assert abs(e - fx_reg["err_cost"]) <= 1e-9 * abs(fx_reg["err_cost"])
    # cost-only path returns the same cost
    c2, _, _ = r.linearize(fx_reg["guess"], want_H=False)
    assert abs(c2 - cost) <= 1e-12 * abs(cost)


def test_linearize_direct7(orc, fx_reg):
    r = _reg(orc, fx_reg, neighbor_method=orc.DIRECT7)
    cost, H, b = r.linearize(fx_reg["guess"])
    assert r.num_correspondences == int(fx_reg["lin7_ncorr"])
    assert abs(cost - fx_reg["lin7_cost"]) <= 1e-9 * abs(fx_reg["lin7_cost"])
    assert np.abs(H - fx_reg["lin7_H"]).max() <= 1e-9 * np.abs(fx_reg["lin7_H"]).max()
    assert np.abs(b - fx_reg["lin7_b"]).max() <= 1e-9 * np.abs(fx_reg["lin7_b"]).max()


def test_lm_trace_and_final(orc, fx_reg):
    r = _reg(orc, fx_reg)
    T = r.align(fx_reg["guess"])
    n = len(fx_reg["lm_y0"])
    assert r.iterations == n
    assert r.converged == bool(fx_reg["converged"])
    for k, t in enumerate(r.trace):
        assert t["inner"] == fx_reg["lm_inner"][k]
        assert t["accepted"] == fx_reg["lm_accepted"][k]
        assert t["n_corr"] == fx_reg["lm_ncorr"][k]
        assert abs(t["y0"] - fx_reg["lm_y0"][k]) <= 1e-7 * abs(fx_reg["lm_y0"][k])
        assert np.abs(t["x"] - fx_reg["lm_x"][k]).max() < 1e-8
    assert np.abs(T - fx_reg["final_T"]).max() < 1e-6
    assert abs(r.fitness() - fx_reg["fitness"]) <= 1e-5 * fx_reg["fitness"]


def test_identity_registration(orc, fx_reg):
    """cloud against itself: every voxel's residuals sum to zero at identity, so the recovered motion is tiny
    (not exactly zero: each residual is weighted by its own Mahalanobis matrix)"""
    r = orc.Registration(num_threads=4)
    r.set_target(fx_reg["tgt"])
    r.set_source(fx_reg["tgt"])
    T = r.align(np.eye(4))
    assert np.abs(T - np.eye(4)).max() < 2e-2


def test_voxelgrid(orc, fx_vg):
    for leaf, key in ((0.2, "out_02"), (0.3, "out_03")):
        out = orc.voxelgrid_filter(fx_vg["xyzi"], leaf)
        assert out.shape == fx_vg[key].shape
        assert np.abs(out - fx_vg[key]).max() < 1e-4


def _check_frontend_against_fixture(o, fx, rel_tol_intensity=3e-4):
    """o: a front-end result (the C oracle's or the HIP path's dict); fx: tests/golden/fx_frontend.npz (the literal numpy restatement)"""
    order = fx["ring_major_order"]
    assert o["n_cloud"] == len(fx["raw"]) and np.array_equal(o["ring_count"][:16], fx["ring_count"])
    assert np.array_equal(o["cloud"][:, :3], fx["raw"][order, :3])                                   # A1 / A2
    for k in ("curvature", "curvature2", "inten_curvature"):                                         # A3 / A4, fp32 bit for bit
        assert np.array_equal(o[k], fx[k]), k
    assert np.array_equal(o["ground_marked"], fx["ground_marked"])                                   # A5 marks, ground points in push order
    n_g = o["n_ground"] if "n_ground" in o else len(o["ground_pts"])
    assert n_g == len(fx["ground_point_index"])
    assert np.array_equal(o["ground_pts"][:n_g, :3], fx["raw"][order][fx["ground_point_index"], :3])
    assert o["ground_valid"]
    g, gf = np.asarray(o["groundparam"]), fx["groundparam"]                                          # A5 plane (fp64; eigenvector signs are arbitrary)
    assert np.abs(g[0:3] - gf[0:3]).max() < 1e-8 and abs(g[9] - gf[9]) < 1e-9 and abs(g[10] - gf[10]) < 1e-9
    for a in (3, 6):
        assert min(np.abs(g[a:a + 3] - gf[a:a + 3]).max(), np.abs(g[a:a + 3] + gf[a:a + 3]).max()) < 1e-6
    assert np.array_equal(o["label"], fx["label"]) and np.array_equal(o["picked"], fx["picked"])     # A6 / A7 decisions
    for k in ("sharp", "flat", "inten"):                                                             # A8 clouds: x y z and the weight exactly;
        assert o[k].shape == fx[k].shape, k                                                          # ring + 0.1 relTime to the azimuth reconstruction
        assert np.array_equal(o[k][:, [0, 1, 2, 4]], fx[k][:, [0, 1, 2, 4]]), k
        assert len(o[k]) == 0 or np.abs(o[k][:, 3] - fx[k][:, 3]).max() < rel_tol_intensity, k


def test_frontend_fixture(orc):
    """SURVEY 8c's fx_frontend: the C oracle's front-end on the committed sweep against the literal numpy restatement's labels, curvatures,
    ground marks / points / plane and feature clouds (tests/golden/gen_frontend_fuse.py)."""
    import os
    from conftest import GOLDEN
    fx = dict(np.load(os.path.join(GOLDEN, "fx_frontend.npz")))
    _check_frontend_against_fixture(orc.frontend(fx["raw"]), fx)


def test_fuse_fixture():
    """SURVEY 8c's fx_fuse: the product's host-side pose fusion (rgc_fuse_pose: no GPU needed) against the committed scipy solutions of the
    restated Ceres problem (src/RGC_odometer.cpp:1025-1119; oracle/py_fusion.py)."""
    import ctypes as C
    import json
    import os
    from conftest import GOLDEN
    from rgc_slam_amd import _lib
    h = _lib.load()
    fx = json.load(open(os.path.join(GOLDEN, "fx_fuse.json")))
    assert len(fx["cases"]) >= 10
    for c in fx["cases"]:
        fin = _lib.FuseIn()
        h.rgc_default_fuse_in(C.byref(fin))
        fin.q_lidar_xyzw[:] = c["q_lidar"]; fin.t_lidar[:] = c["t_lidar"]; fin.fitness = c["fitness"]
        fin.use_ground = int(c["use_ground"]); fin.ground_last[:] = c["ground_last"]; fin.ground_cur[:] = c["ground_cur"]
        fin.q_w_curr_f_xyzw[:] = c["q_w_curr_f"]; fin.ground_cov = c["ground_cov"]
        fin.use_imu = int(c["use_imu"]); fin.q_imu_xyzw[:] = c["q_imu"]
        q, t, it = np.empty(4), np.empty(3), C.c_int(0)
        dp = C.POINTER(C.c_double)
        assert h.rgc_fuse_pose(C.byref(fin), q.ctypes.data_as(dp), t.ctypes.data_as(dp), C.byref(it)) == 0
        qo, to = np.asarray(c["q_fused_xyzw"]), np.asarray(c["t_fused"])
        if np.dot(q, qo) < 0:
            qo = -qo
        assert np.abs(q - qo).max() < 1e-6 and np.abs(t - to).max() < 1e-6 and it.value <= 6


def test_general_regularization_methods_and_multiplicative_voxels(orc, fx_reg):
    """The C oracle's restatement of fast_gicp_impl.hpp:262-293 (NONE, MIN_EIG, NORMALIZED_MIN_EIG, FROBENIUS beside PLANE) and of
    MultiplicativeGaussianVoxel (fast_vgicp_voxel.hpp:76-99) against the literal numpy restatement (oracle/py_oracle.py: np.linalg.svd /
    inv as the reference's JacobiSVD / inverse()): covariances, H / b / cost, and the registration's pose.  CPU only."""
    from oracle import py_oracle as po
    tgt, src = fx_reg["tgt"][:2500], fx_reg["src"][:600]
    names = {"NONE": orc.REG_NONE, "MIN_EIG": orc.REG_MIN_EIG, "NORMALIZED_MIN_EIG": orc.REG_NORMALIZED_MIN_EIG, "PLANE": orc.REG_PLANE,
             "FROBENIUS": orc.REG_FROBENIUS}
    for nm, m in names.items():
        c_py, _ = po.covariances(src, regularization=nm)
        c_c = orc.covariances_m(src, m)
        assert np.abs(c_py[:, :3, :3] - c_c).max() <= 1e-9 * max(1.0, np.abs(c_c).max()), nm
    S = np.array([[0.04, 0.01, 0.0], [0.01, 0.03, 0.002], [0.0, 0.002, 1e-5]])
    ev = np.linalg.eigvalsh(S)[::-1]
    assert np.allclose(np.linalg.eigvalsh(orc.regularize(S, orc.REG_MIN_EIG))[::-1], np.maximum(ev, 1e-3), atol=1e-12)
    assert np.allclose(np.linalg.eigvalsh(orc.regularize(S, orc.REG_NORMALIZED_MIN_EIG))[::-1], np.maximum(ev / ev[0], 1e-3), atol=1e-12)
    assert np.allclose(np.linalg.eigvalsh(orc.regularize(S, orc.REG_PLANE))[::-1], [1.0, 1.0, 1e-3], atol=1e-12)
    Ci = np.linalg.inv(S + 1e-3 * np.eye(3))
    assert np.allclose(orc.regularize(S, orc.REG_FROBENIUS), np.linalg.inv(Ci / np.linalg.norm(Ci)), rtol=1e-10)
    I4 = np.eye(4)
    for nm, vm in (("MIN_EIG", "ADDITIVE"), ("FROBENIUS", "MULTIPLICATIVE"), ("PLANE", "MULTIPLICATIVE")):
        r_py = po.VGICP(regularization=nm, voxel_mode=vm); r_py.set_target(tgt); r_py.set_source(src)
        r_c = orc.Registration(regularization=names[nm], voxel_mode=orc.VOXEL_MULTIPLICATIVE if vm == "MULTIPLICATIVE" else orc.VOXEL_ADDITIVE, num_threads=2)
        r_c.set_target(tgt); r_c.set_source(src)
        cost_py, H_py, b_py = r_py.linearize(I4)
        cost_c, H_c, b_c = r_c.linearize(I4)
        assert abs(cost_py - cost_c) <= 1e-10 * abs(cost_c) and np.abs(H_py - H_c).max() <= 1e-10 * np.abs(H_c).max() and np.abs(b_py - b_c).max() <= 1e-10 * np.abs(b_c).max(), (nm, vm)
        Tp, Tc = r_py.align(np.eye(4, dtype=np.float32)), r_c.align(np.eye(4, dtype=np.float32))
        assert np.abs(np.asarray(Tp, np.float64) - np.asarray(Tc, np.float64)).max() <= 1e-6, (nm, vm)
