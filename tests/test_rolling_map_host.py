"""f2 host logic on the CPU: the RollingOdometer frame body (map-frame guess, delta recovery, keyframe insert / evict / re-base)
driven by the oracle backend, against the reference-semantics Odometer driven by the same backend.  The two differ only by the
frame the 0.3 m leaf lattice and the 1 m voxel lattice are aligned to, so they must follow the same motion.  No GPU."""
import numpy as np


def _raws(n, n_az=600):
    import rgc_slam_amd.synth as synth
    world = synth.make_world(half_extent=45.0, seed=synth.SEED)
    poses = synth.make_trajectory(n + 1, seed=synth.SEED)
    raws = []
    for k in range(n):
        sc = synth.make_scan(world, poses[k], n_az=n_az, seed=synth.SEED + 50 + k, T_ws_end=poses[k + 1])
        raws.append(np.concatenate([sc["xyz"], sc["intensity"][:, None]], axis=1).astype(np.float32))
    return raws, poses


def test_rolling_odometer_follows_reference_semantics():
    from rgc_slam_amd import odometry
    from oracle_backend import OracleBackend
    raws, poses = _raws(7)
    roll, ref = odometry.RollingOdometer(OracleBackend()), odometry.Odometer(OracleBackend())
    roll.rebase_distance = 0.4
    d = []
    for raw in raws:
        qa, ta = roll.process(raw)
        qb, tb = ref.process(raw)
        d.append((np.linalg.norm(ta - tb), 2 * np.arccos(min(1.0, abs(float(np.dot(qa, qb)))))))
    assert roll.frames == 7 and np.linalg.norm(roll.t_w_curr) > 0.3
    assert max(x for x, _ in d) < 0.05 and max(a for _, a in d) < 5e-3, d
    assert len(roll.b._kf) <= roll.slipwide and roll.n_commits >= 2
    assert not np.array_equal(roll.origin, np.zeros(3))      # re-based during the run
    # the map never holds more than slipwide keyframes and ids keep counting
    assert roll.b._kf[-1][0] == roll.b._next - 1


def test_rolling_odometer_distance_eviction():
    from rgc_slam_amd import odometry
    from oracle_backend import OracleBackend
    raws, _ = _raws(6)
    roll = odometry.RollingOdometer(OracleBackend(), max_keyframes=0, radius=0.35)
    for raw in raws:
        roll.process(raw)
    kf_t = np.array([k[2] for k in roll.b._kf])
    assert len(kf_t) >= 1 and np.all(np.linalg.norm(kf_t - roll.b._kf[-1][2], axis=1) <= 0.35 + 1e-9)
