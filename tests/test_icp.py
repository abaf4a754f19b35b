"""f4: loop-closure ICP.  CPU: the C oracle (orc_icp_align) against the independent numpy/scipy restatement (oracle/py_icp.py).
GPU (-m gpu): the HIP path through the C-ABI against the oracle."""
import numpy as np
import pytest


@pytest.fixture(scope="module")
def case():
    import rgc_slam_amd.synth as synth
    world, tgt = synth.make_world_and_map(60000)
    T_true = synth.se3(synth.rot_zyx(0.05, 0.01, -0.008), [0.4, -0.25, 0.05])
    src = synth.make_scan_n(world, np.eye(4), 8000, seed=synth.SEED + 6)["xyz"]
    Ti = np.linalg.inv(T_true)   # the latest key frame sits at a drifted pose: ICP has to find T_true
    return dict(tgt=tgt, src=(src @ Ti[:3, :3].T + Ti[:3, 3]).astype(np.float32), T_true=T_true)


def test_rigid_fit_matches_numpy_svd():
    from oracle import oracle, py_icp
    import ctypes as C
    rng = np.random.default_rng(2)
    for trial in range(4):
        p = rng.normal(0, 5, (50, 3))
        if trial == 3:
            p[:, 2] = 0.0      # planar cloud: rank-deficient correlation, reflection case must still give a rotation
        a = rng.normal(0, 0.4, 3)
        R0, _ = np.linalg.qr(rng.normal(size=(3, 3)))
        R0 *= np.sign(np.linalg.det(R0))
        q = p @ R0.T + a + rng.normal(0, 0.01, p.shape)
        R, t = np.zeros(9), np.zeros(3)
        dp = C.POINTER(C.c_double)
        spq = (p[:, :, None] * q[:, None, :]).sum(0).reshape(9)
        oracle.lib().orc_rigid_from_sums(float(len(p)), np.ascontiguousarray(p.sum(0)).ctypes.data_as(dp), np.ascontiguousarray(q.sum(0)).ctypes.data_as(dp),
                                         np.ascontiguousarray(spq).ctypes.data_as(dp), R.ctypes.data_as(dp), t.ctypes.data_as(dp))
        Rn, tn = py_icp.rigid_fit(p, q)
        assert np.abs(R.reshape(3, 3) - Rn).max() < 1e-9 and np.abs(t - tn).max() < 1e-9
        assert abs(np.linalg.det(R.reshape(3, 3)) - 1) < 1e-12


def test_oracle_icp_matches_numpy(case):
    from oracle import oracle, py_icp
    To, ro = oracle.icp_align(case["src"], case["tgt"], max_corr_dist=10.0)
    Tn, rn = py_icp.icp_align(case["src"], case["tgt"], max_corr_dist=10.0)
    assert ro["converged"] == 1 and ro["iterations"] == rn["iterations"]
    assert ("not_converged", "iterations", "transform", "abs_mse", "rel_mse", "no_correspondences")[ro["state"]] == rn["state"]
    assert np.abs(To - Tn).max() < 1e-6 and abs(ro["fitness"] - rn["fitness"]) <= 1e-6 * rn["fitness"]
    # the drift is recovered to the accuracy point-to-point ICP has on a 0.3 m-sampled map
    assert np.abs(To[:3, 3] - case["T_true"][:3, 3]).max() < 0.05 and np.abs(To[:3, :3] - case["T_true"][:3, :3]).max() < 5e-3
    # iteration cap and correspondence gate
    T2, r2 = oracle.icp_align(case["src"], case["tgt"], max_corr_dist=10.0, max_iterations=3)
    assert r2["iterations"] == 3 and r2["state"] == 1 and r2["converged"] == 1
    T3, r3 = oracle.icp_align(case["src"] + np.float32([0, 0, 60.0]), case["tgt"], max_corr_dist=2.0)
    assert r3["converged"] == 0 and r3["state"] == 5 and r3["n_correspondences"] < 3


@pytest.mark.gpu
def test_hip_icp_matches_oracle(case):
    from rgc_slam_amd import loop_closure
    from oracle import oracle
    icp = loop_closure.IterativeClosestPoint(0)
    icp.setMaxCorrespondenceDistance(10.0)
    icp.setMaximumIterations(100)
    icp.setTransformationEpsilon(1e-6)
    icp.setEuclideanFitnessEpsilon(1e-6)
    icp.setInputSource(case["src"])
    icp.setInputTarget(case["tgt"])
    T = icp.align().copy()
    To, ro = oracle.icp_align(case["src"], case["tgt"], max_corr_dist=10.0)
    assert icp.hasConverged() and icp.nr_iterations == ro["iterations"] and icp.convergence_state == "transform"
    assert np.abs(T - To).max() < 1e-5
    assert abs(icp.getFitnessScore() - ro["fitness"]) <= 1e-5 * ro["fitness"]
    icp.setMaximumIterations(3)
    T3 = icp.align()
    T3o, r3 = oracle.icp_align(case["src"], case["tgt"], max_corr_dist=10.0, max_iterations=3)
    assert icp.nr_iterations == 3 and icp.convergence_state == "iterations" and np.abs(T3 - T3o).max() < 1e-5
    icp.setMaximumIterations(100)
    icp.setMaxCorrespondenceDistance(2.0)
    icp.setInputSource(case["src"] + np.float32([0, 0, 60.0]))
    icp.align()
    assert not icp.hasConverged() and icp.convergence_state == "no_correspondences"
    icp.close()
