"""One trial of tests/fuzz/fuzz_sequence.py with the registrations' iteration counts per frame.   python tests/fuzz/fuzz_sequence_repro.py <trial> <seed> <sweeps>"""
import sys, os, math
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import rgc_slam_amd.synth as synth
from rgc_slam_amd import odometry
import oracle_backend
from oracle_backend import OracleBackend
from oracle import oracle as orc
trial, seed0, n_sweeps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
rng = np.random.default_rng(seed0 * 15485863 + trial)
use_imu = bool(rng.random() < 0.5); ramp = bool(rng.random() < 0.4); n_az = int(rng.choice([600, 900, 1200, 1800]))
wseed, tseed = int(rng.integers(1, 1 << 30)), int(rng.integers(1, 1 << 30))
world = synth.make_world(half_extent=float(rng.choice([35.0, 45.0, 60.0])), seed=wseed)
base = synth.make_trajectory(n_sweeps + 1, seed=tseed); poses = base
if ramp:
    slope = float(rng.uniform(0.03, 0.08)); x0 = base[min(3, n_sweeps // 2)][0, 3]; world.ramp = (x0, x0 + 4.0, slope); poses = []
    for P in base:
        Q = P.copy(); Q[2, 3] += float(world.ground_height(Q[0, 3])) - world.ground_z
        if world.ramp[0] <= Q[0, 3] <= world.ramp[1]: Q[:3, :3] = Q[:3, :3] @ synth.rot_zyx(0.0, -math.atan(slope), 0.0)
        poses.append(Q)
raws = []
for k in range(n_sweeps):
    sc = synth.make_scan(world, poses[k], n_az=n_az, seed=int(rng.integers(1, 1 << 30)), T_ws_end=poses[k + 1])
    raws.append(np.concatenate([sc["xyz"], sc["intensity"][:, None]], axis=1).astype(np.float32))
print("use_imu", use_imu, "ramp", ramp, "n_az", n_az)
hb = odometry.HipBackend(0); ob = OracleBackend()
log = {"hip": [], "orc": []}
hreg = hb.register
def hip_register(source, target, guess):
    T, f = hreg(source, target, guess); log["hip"].append((hb.reg.nr_iterations, hb.reg.hasConverged(), len(source), len(target), f)); return T, f
hb.register = hip_register
def orc_register(source, target, guess):
    r = orc.Registration(num_threads=14); r.set_target(target); r.set_source(source); T = r.align(guess)
    log["orc"].append((r.iterations, r.converged, len(source), len(target), r.fitness())); return T, r.fitness()
ob.register = orc_register
kw = dict(use_imu=True, first_frames=2) if use_imu else {}
og, oc = odometry.Odometer(hb, **kw), odometry.Odometer(ob, **kw)
prev = None
for k, raw in enumerate(raws):
    rg, rc = og.process(raw), oc.process(raw)
    if rg is None: continue
    (qg, tg), (qc, tc) = rg, rc
    if prev is not None:
        print(k, "dt", float(np.abs((tg - prev[0][1]) - (tc - prev[1][1])).max()), "submap", len(og.submap), len(oc.submap), "hip", log["hip"][-1] if log["hip"] else None, "orc", log["orc"][-1] if log["orc"] else None)
    prev = ((qg, tg), (qc, tc))
