"""The front-end half of one trial of tests/fuzz/fuzz_pre.py, with the sweep's encoded time compared point by point.  python tests/fuzz/fuzz_pre_repro.py <trial> <seed>"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import rgc_slam_amd.synth as synth
from rgc_slam_amd import frontend
import oracle as orc
trial, seed0 = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed0 * 104729 + trial)
beams = int(rng.choice([16, 16, 32, 64]))
w = synth.make_world(half_extent=float(rng.choice([25.0, 40.0, 60.0])), seed=int(rng.integers(1, 1 << 30)))
elev = synth.VLP16_ELEV if beams == 16 else (synth.hdl32_elev() if beams == 32 else synth.hdl64_elev())
T = synth.se3(synth.rot_zyx(rng.uniform(-np.pi, np.pi), rng.normal(0, 0.02), rng.normal(0, 0.02)), rng.uniform(-10, 10, 3) * np.array([1, 1, 0.01]))
n_az = int(rng.integers(150, 2200 if beams < 64 else 1200))
sc = synth.make_scan(w, T, elev_deg=elev, n_az=n_az, seed=int(rng.integers(1, 1 << 30)))
raw = np.concatenate([sc["xyz"], sc["intensity"][:, None]], axis=1).astype(np.float32)
what = str(rng.choice(["plain", "plain", "dropped", "nan", "shuffled", "short", "high", "scaled"]))
print(beams, n_az, what, len(raw))
if what == "shuffled": raw = raw[rng.permutation(len(raw))]
elif what == "dropped": raw = raw[rng.random(len(raw)) < rng.uniform(0.2, 0.9)]
f = frontend.ScanRegistration(beams)
g = f.laserCloudHandler(raw); o = orc.frontend(raw, n_scans=beams)
d = np.abs(g["cloud"][:, 3] - o["cloud"][:, 3])
print("n_cloud", g["n_cloud"], o["n_cloud"], "xyz equal", np.array_equal(g["cloud"][:, :3], o["cloud"][:, :3]))
print("encoded time: max diff", d.max(), "points > 8e-6:", int((d > 8e-6).sum()), "their diffs (sorted, first 10):", np.sort(d[d > 8e-6])[-10:])
bad = np.nonzero(d > 8e-6)[0][:8]
for i in bad:
    p = g["cloud"][i]; print("  ", i, p[:3], "hip", p[3], "oracle", o["cloud"][i, 3], "azimuth", -np.degrees(np.arctan2(p[1], p[0])))
if o["ground_valid"]:
    print("groundparam hip", g["groundparam"]); print("groundparam orc", o["groundparam"]); print("n_ground", g["n_ground"], len(o["ground_pts"]))
