"""Generates tests/golden/fx_frontend.npz and fx_fuse.json (SURVEY.md 8c's list) from the literal numpy / scipy restatements
(oracle/py_frontend.py, oracle/py_fusion.py) -- NOT from the C oracle and not from the product: they pin both.

fx_frontend.npz: one synthetic VLP-16 sweep (tilted pose: the ground plane is not the sensor's xy plane), its ring-major cloud as the
ring bucket of A2 orders it (pinned separately by the sensor model's known ring ids, tests/test_oracle_frontend.py), and from the
restatement: curvatures (A3 / A4), occlusion mask (A6), ground marks, ground points and the ground plane (A5), labels and the three
feature clouds (A7).  fx_fuse.json: pose-fusion problems (B7: lidar prior + ground factor + IMU factor, src/RGC_odometer.cpp:1025-1119)
-> fused (q, t) from scipy.optimize.least_squares on the restated residuals.
The reference holds no vectors for either stage ("parity unpinned").   python tests/golden/gen_frontend_fuse.py"""
import json
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import rgc_slam_amd.synth as synth  # noqa: E402
from oracle import py_frontend as pf  # noqa: E402
from oracle import py_fusion as pfu  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
SEED = synth.SEED


def frontend_fixture():
    w = synth.make_world(half_extent=50.0, seed=SEED)
    T = synth.se3(synth.rot_zyx(0.3, 0.03, -0.015), [1.0, -2.0, 0.0])
    sc = synth.make_scan(w, T, n_az=600, seed=SEED + 21)
    raw = np.concatenate([sc["xyz"], sc["intensity"][:, None]], axis=1).astype(np.float32)
    order = np.argsort(sc["ring"], kind="stable")                      # A2: stable bucket by ring, firing order inside a ring
    ring = sc["ring"][order]
    cloud = raw[order].copy()
    # intensity = scanID + 0.1 * relTime, relTime from the azimuth as :186-210 reconstruct it -- taken from the generator's rel_time here;
    # the encoded value is an INPUT of the stages restated below only through floor() (the ring) -- the fixture stores what it used
    cloud[:, 3] = (ring + 0.1 * sc["rel_time"][order]).astype(np.float32)
    ring_count = np.bincount(ring, minlength=16).astype(np.int32)
    st = pf.stencils(cloud[:, :3], raw[order, 3].astype(np.int64))
    mark, pushed, g = pf.ground(cloud, ring_count, st["range"])
    scan_start = np.zeros(16, np.int32); scan_end = np.zeros(16, np.int32)
    acc = 0
    for i in range(16):                                                 # scanStartInd / scanEndInd, :222-230
        scan_start[i] = acc + 5
        acc += int(ring_count[i])
        scan_end[i] = acc - 5
    sel = pf.select(cloud, st, pf.occlusion(st["range"]), mark, scan_start, scan_end)
    np.savez_compressed(os.path.join(OUT, "fx_frontend.npz"), seed=SEED, raw=raw, ring_major_order=order.astype(np.int32), ring_count=ring_count,
                        range=st["range"], curvature=st["curvature"], curvature2=st["curvature2"], inten_curvature=st["inten_curvature"],
                        occlusion=pf.occlusion(st["range"]).astype(np.int32), ground_marked=mark, ground_point_index=pushed.astype(np.int32),
                        groundparam=g, label=sel["label"].astype(np.int32), picked=sel["picked"].astype(np.int32),
                        sharp=sel["sharp"], flat=sel["flat"], inten=sel["inten"], scan_start=scan_start, scan_end=scan_end)
    print("fx_frontend.npz:", len(raw), "points,", int(mark.sum()), "ground marks,", len(sel["sharp"]), "sharp,", len(sel["flat"]), "flat,", len(sel["inten"]), "inten")


def rand_q(rng, angle):
    w = rng.normal(size=3); w *= angle / np.linalg.norm(w)
    n = np.linalg.norm(w)
    return np.array([*(math.sin(n / 2) / n * w), math.cos(n / 2)])


def fuse_fixture():
    rng = np.random.default_rng(SEED)
    cases = []
    for use_ground, use_imu, big in ((True, True, False), (True, True, True), (True, False, False), (False, True, False), (False, False, False)):
        for _ in range(3):
            n_last = np.array([0.01, -0.02, 1.0]); n_last /= np.linalg.norm(n_last)
            v1 = np.cross(n_last, [1, 0, 0]); v1 /= np.linalg.norm(v1)
            v2 = np.cross(n_last, v1)
            n_cur = n_last + rng.normal(0, 0.01, 3); n_cur /= np.linalg.norm(n_cur)
            c = dict(q_lidar=rand_q(rng, 0.03), t_lidar=rng.normal(0, 0.1, 3), fitness=float(rng.uniform(0.02, 0.3)), use_ground=use_ground,
                     ground_last=np.array([*n_last, *v1, *v2, 0.56, 0.02]), ground_cur=np.array([*n_cur, *v1, *v2, 0.56 + rng.normal(0, 0.01), 0.03]),
                     q_w_curr_f=rand_q(rng, 0.05), ground_cov=0.2, use_imu=use_imu, q_imu=rand_q(rng, 0.05 if big else 0.005))
            q, t = pfu.fuse(c)
            cases.append({k: (v.tolist() if isinstance(v, np.ndarray) else v) for k, v in c.items()} | {"q_fused_xyzw": q.tolist(), "t_fused": t.tolist()})
    json.dump({"seed": SEED, "what": "B7 pose fusion: inputs -> (q, t) of scipy.optimize.least_squares over oracle/py_fusion.residuals", "cases": cases},
              open(os.path.join(OUT, "fx_fuse.json"), "w"), indent=1)
    print("fx_fuse.json:", len(cases), "cases")


if __name__ == "__main__":
    frontend_fixture()
    fuse_fixture()
