"""Generates tests/golden/fx_sequence.npz: a 10-sweep synthetic VLP-16 sequence with a 200 Hz IMU stream through the literal restatement
of vg_ICP::ICP_thread's frame body (oracle/py_odometer.py, USE_IMU = 1, USE_GROUND = 1 -- launch/run.launch:18,20) on the CPU oracle's
stages.  The inputs are regenerated from their seeds by the tests (rgc_slam_amd.synth is deterministic); the fixture holds their SHA-256
and, per sweep, what the frame body produced: pose, gflag, changegroundflag, number of keyframes, sub-map size, fitness.

    python tests/golden/gen_sequence.py        # CPU only, ~1 min
"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def inputs():
    """the sweeps (x, y, z, intensity; motion-distorted), their stamps and the IMU stream"""
    import rgc_slam_amd.synth as synth
    world = synth.make_world(half_extent=45.0, seed=synth.SEED)
    poses = synth.make_trajectory(11, seed=synth.SEED + 2)
    stamps, acc, gyr = synth.make_imu(poses, seed=synth.SEED + 2)
    raws = []
    for k in range(10):
        sc = synth.make_scan(world, poses[k], n_az=1200, seed=synth.SEED + 70 + k, T_ws_end=poses[k + 1])
        raws.append(np.concatenate([sc["xyz"], sc["intensity"][:, None]], axis=1).astype(np.float32))
    return raws, [0.1 * (k + 1) for k in range(10)], (stamps, acc, gyr)


def digest(raws, imu):
    h = hashlib.sha256()
    for r in raws:
        h.update(np.ascontiguousarray(r).tobytes())
    for a in imu:
        h.update(np.ascontiguousarray(a, np.float64).tobytes())
    return h.hexdigest()


def feed(node, raws, sweep_stamps, imu, imu_cb, handle):
    """the messages in arrival order: every IMU sample up to one past a sweep's stamp (:1405-1406), then the sweep"""
    stamps, acc, gyr = imu
    j, out = 0, []
    for raw, t_k in zip(raws, sweep_stamps):
        while j < len(stamps) and stamps[j] <= t_k + 0.011:
            imu_cb(stamps[j], acc[j], gyr[j])
            j += 1
        out.append(handle(raw, t_k))
    return out


def main():
    from oracle import oracle, py_odometer
    oracle.build()
    raws, sweep_stamps, imu = inputs()
    node = py_odometer.IcpThread(USE_IMU=1, USE_GROUND=1, firstflagnum=2)
    rows = []

    def handle(raw, t_k):
        r = node.handle(raw, t_k)
        rows.append(dict(produced=r is not None, q=node.q_w_curr.copy(), t=node.t_w_curr.copy(), gflag=node.gflag, changegroundflag=node.changegroundflag,
                         keyframes=len(node.surroundingCloud), submap=len(node.laserCloudsubmap), fitness=node.vgicp_source, submapflag=node.submapflag))
        return r
    feed(node, raws, sweep_stamps, imu, node.imuCallback, handle)
    out = {k: np.array([r[k] for r in rows]) for k in rows[0]}
    out["inputs_sha256"] = np.array(digest(raws, imu))
    np.savez(os.path.join(ROOT, "tests", "golden", "fx_sequence.npz"), **out)
    for i, r in enumerate(rows):
        print(i, r["produced"], np.round(r["t"], 4), "gflag", r["gflag"], "kf", r["keyframes"], "submap", r["submap"], "fit %.4f" % r["fitness"])


if __name__ == "__main__":
    main()
