"""BASELINE config 2 stand-in (the example bag is not reachable offline): a synthetic VLP-16 sequence through the whole
frame body -- front-end, de-skew, VoxelGrid 0.2/0.3, FastVGICP, fitness, ground-constrained pose fusion, sliding
3-keyframe sub-map -- on the GPU vs the same frame body driven by the CPU oracle.  Per-frame pose deltas must agree to
1e-4 m / 1e-4 rad (north_star).  -m gpu."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _angle(qa, qb):
    d = abs(float(np.dot(qa, qb)))
    return 2 * np.arccos(min(1.0, d))


def test_sequence_with_ground_constraint():
    import rgc_slam_amd.synth as synth
    from rgc_slam_amd import odometry
    from oracle_backend import OracleBackend
    world = synth.make_world(half_extent=45.0, seed=synth.SEED)
    poses = synth.make_trajectory(9, seed=synth.SEED)
    raws = []
    for k in range(8):   # sweeps with motion distortion: pose interpolated between consecutive trajectory poses
        sc = synth.make_scan(world, poses[k], n_az=1200, seed=synth.SEED + 50 + k, T_ws_end=poses[k + 1])
        raws.append(np.concatenate([sc["xyz"], sc["intensity"][:, None]], axis=1).astype(np.float32))
    hb = odometry.HipBackend(0)
    og, oc = odometry.Odometer(hb), odometry.Odometer(OracleBackend())
    prev_g, prev_c = (np.array([0, 0, 0, 1.0]), np.zeros(3)), (np.array([0, 0, 0, 1.0]), np.zeros(3))
    worst_t = worst_r = 0.0
    for k, raw in enumerate(raws):
        qg, tg = og.process(raw)
        qc, tc = oc.process(raw)
        # per-frame pose delta of each path
        dtg, dtc = tg - prev_g[1], tc - prev_c[1]
        worst_t = max(worst_t, float(np.abs(dtg - dtc).max()))
        worst_r = max(worst_r, abs(_angle(qg, prev_g[0]) - _angle(qc, prev_c[0])), _angle(qg, qc) if k < 3 else 0.0)
        prev_g, prev_c = (qg, tg), (qc, tc)
    hb.close()
    assert og.frames == 8 and np.linalg.norm(og.t_w_curr) > 0.3               # the platform really moved
    assert worst_t <= 1e-4 and worst_r <= 1e-4, (worst_t, worst_r)
    # and the estimate follows the true motion (sensor frame of pose 8 vs pose 1 ... coarse sanity, not parity)
    true = np.linalg.inv(poses[1]) @ poses[8]
    assert np.linalg.norm(og.t_w_curr - true[:3, 3]) < 0.5
