"""f1: mapping-node feature registration, HIP path (through the C-ABI) vs the CPU oracle.  -m gpu.

Association: validity flags identical up to features sitting on a threshold; factor parameters 1e-9 (edge points up to the
free sign of the eigenvector).  Solve: same LM loop, device sums in a different order: poses 1e-7, costs 1e-9 relative."""
import numpy as np
import pytest

import mapreg_data as md

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def case():
    import rgc_slam_amd.synth as synth
    from oracle import oracle
    c = md.make_case(synth, oracle.frontend, n_map_frames=12, n_az=1800, voxelgrid=oracle.voxelgrid_filter)
    rng = np.random.default_rng(3)
    c["x0"] = md.poses14(md.perturb(c["T_cur"], rng), md.perturb(c["T_last"], rng))
    c["xt"] = md.poses14(c["T_cur"], c["T_last"])
    return c


@pytest.fixture(scope="module")
def reg(case):
    from rgc_slam_amd import mapping
    r = mapping.MapFeatureRegistration(0)
    r.setInputMaps(case["corner_map"], case["surf_map"])
    yield r
    r.close()


@pytest.mark.parametrize("kind", ["edge", "plane"])
def test_association_vs_oracle(case, reg, kind):
    from oracle import oracle
    feat, mp = (case["corner_cur"], case["corner_map"]) if kind == "edge" else (case["surf_cur"], case["surf_map"])
    q, t = case["x0"][0:4], case["x0"][4:7]
    a = reg.associate(feat, q, t, kind)
    b = oracle.mapreg_associate(feat, q, t, mp, kind)
    assert a["n_valid"] == int(a["valid"].sum()) and a["n_valid"] > 100
    assert (a["valid"] != b["valid"]).sum() <= 2
    both = a["valid"] & b["valid"]
    if kind == "edge":
        d1 = np.abs(a["a"][both] - b["a"][both]).max(axis=1)
        d2 = np.abs(a["a"][both] - b["b"][both]).max(axis=1)
        assert np.minimum(d1, d2).max() < 1e-9
        mid = 0.5 * (a["a"][both] + a["b"][both]) - 0.5 * (b["a"][both] + b["b"][both])
        assert np.abs(mid).max() < 1e-12
    else:
        assert np.abs(a["n"][both] - b["n"][both]).max() < 1e-9 and np.abs(a["d"][both] - b["d"][both]).max() < 1e-8
    assert np.array_equal(a["var"][both], b["var"][both])


def test_optimize_vs_oracle(case, reg):
    from oracle import oracle
    x0 = case["x0"]
    qc, tc, ql, tl, rep = reg.optimize(case["corner_cur"], case["surf_cur"], case["corner_last"], case["surf_last"], x0[0:4], x0[4:7], x0[7:11], x0[11:14])
    xo, rc, tr = oracle.mapreg_optimize(case["corner_cur"], case["surf_cur"], case["corner_last"], case["surf_last"], case["corner_map"],
                                        case["surf_map"], x0)
    assert rc == 0 and rep is not None
    x = np.concatenate([qc, tc, ql, tl])
    for i in range(2):
        assert rep[i]["iterations"] == tr[i]["iterations"] and rep[i]["successful"] == tr[i]["successful"]
        assert (rep[i]["n_edge_cur"], rep[i]["n_edge_last"], rep[i]["n_plane_cur"], rep[i]["n_plane_last"]) == \
               (tr[i]["n_edge_cur"], tr[i]["n_edge_last"], tr[i]["n_plane_cur"], tr[i]["n_plane_last"])
        assert abs(rep[i]["initial_cost"] - tr[i]["initial_cost"]) <= 1e-9 * tr[i]["initial_cost"]
        assert abs(rep[i]["final_cost"] - tr[i]["final_cost"]) <= 1e-9 * tr[i]["final_cost"]
    assert np.abs(x - xo).max() < 1e-7
    # and it moved towards the truth
    assert np.abs(x - case["xt"]).max() < 0.5 * np.abs(x0 - case["xt"]).max()


def test_optimize_with_ground_block_vs_oracle(case, reg):
    from oracle import oracle
    x0 = case["x0"]
    gc, gl = md.make_ground(case["T_cur"], case["T_last"]), md.make_ground(case["T_last"], case["T_last"], tilt=(-0.004, 0.006))
    qc, tc, ql, tl, rep = reg.optimize(case["corner_cur"], case["surf_cur"], case["corner_last"], case["surf_last"], x0[0:4], x0[4:7], x0[7:11], x0[11:14],
                                       ground_cur=gc, ground_last=gl)
    xo, rc, tr = oracle.mapreg_optimize(case["corner_cur"], case["surf_cur"], case["corner_last"], case["surf_last"], case["corner_map"],
                                        case["surf_map"], x0, ground_cur=gc, ground_last=gl)
    x = np.concatenate([qc, tc, ql, tl])
    for i in range(2):
        assert rep[i]["iterations"] == tr[i]["iterations"] and rep[i]["successful"] == tr[i]["successful"]
        assert abs(rep[i]["final_cost"] - tr[i]["final_cost"]) <= 1e-9 * tr[i]["final_cost"]
    assert np.abs(x - xo).max() < 1e-7
    plain = np.concatenate(reg.optimize(case["corner_cur"], case["surf_cur"], case["corner_last"], case["surf_last"], x0[0:4], x0[4:7], x0[7:11], x0[11:14])[:4])
    assert np.abs(x - plain).max() > 1e-6   # the block is really in the problem


def test_optimize_with_imu_block_vs_oracle(case, reg):
    """the IMU block couples the two rotations: a 12 x 12 solve, alone and together with the ground block"""
    from oracle import oracle
    x0 = case["x0"]
    gc, gl = md.make_ground(case["T_cur"], case["T_last"]), md.make_ground(case["T_last"], case["T_last"], tilt=(-0.004, 0.006))
    plain = np.concatenate(reg.optimize(case["corner_cur"], case["surf_cur"], case["corner_last"], case["surf_last"], x0[0:4], x0[4:7], x0[7:11], x0[11:14])[:4])
    for imu_cov, ground in ((0.4, False), (0.004, False), (0.4, True)):
        im = md.make_imu(case["T_cur"], case["T_last"], imu_cov=imu_cov)
        kw = dict(ground_cur=gc, ground_last=gl) if ground else {}
        qc, tc, ql, tl, rep = reg.optimize(case["corner_cur"], case["surf_cur"], case["corner_last"], case["surf_last"], x0[0:4], x0[4:7], x0[7:11],
                                           x0[11:14], imu=im, **kw)
        xo, rc, tr = oracle.mapreg_optimize(case["corner_cur"], case["surf_cur"], case["corner_last"], case["surf_last"], case["corner_map"],
                                            case["surf_map"], x0, imu=im, **kw)
        x = np.concatenate([qc, tc, ql, tl])
        for i in range(2):
            assert rep[i]["iterations"] == tr[i]["iterations"] and rep[i]["successful"] == tr[i]["successful"]
            assert abs(rep[i]["initial_cost"] - tr[i]["initial_cost"]) <= 1e-9 * tr[i]["initial_cost"]
            assert abs(rep[i]["final_cost"] - tr[i]["final_cost"]) <= 1e-9 * tr[i]["final_cost"]
        assert np.abs(x - xo).max() < 1e-7
        assert np.abs(x - plain).max() > 1e-6


def test_gate_and_errors(case, reg):
    x0 = case["x0"]
    out = reg.optimize(case["corner_cur"][:5], case["surf_cur"], case["corner_last"], case["surf_last"], x0[0:4], x0[4:7], x0[7:11], x0[11:14])
    assert out[4] is None and np.array_equal(np.concatenate(out[:4]), x0)   # the gate of :1069: poses untouched
    from rgc_slam_amd import mapping, _lib
    r = mapping.MapFeatureRegistration(0)
    with pytest.raises(_lib.RgcError):
        r.optimize(case["corner_cur"], case["surf_cur"], case["corner_last"], case["surf_last"], x0[0:4], x0[4:7], x0[7:11], x0[11:14])  # no maps yet
    r.close()
