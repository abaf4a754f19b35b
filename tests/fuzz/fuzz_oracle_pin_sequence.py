"""Randomised pinning of the FRAME BODY's orchestration (vg_ICP::ICP_thread, RGC_odometer.cpp:848-1256): the literal restatement
oracle/py_odometer.py against the Python mirror rgc_slam_amd.odometry.Odometer -- whose orchestration the GPU sequence tests share between the
library and the oracle -- both on the CPU oracle's stages, on random sequences instead of the one committed fixture (tests/golden/fx_sequence.npz):
random trajectories, sweep counts and densities, with and without motion distortion, USE_IMU / USE_GROUND on and off, firstflagnum.  Sweep by
sweep: produced or not, pose, ground flag, number of keyframes, sub-map size, fitness.  No GPU (slow: two CPU registrations per sweep).
    python tests/fuzz/fuzz_oracle_pin_sequence.py [trials] [seed]"""
import sys, os, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np
import rgc_slam_amd.synth as synth
from rgc_slam_amd import odometry
from oracle import py_odometer, oracle as orc
from oracle_backend import OracleBackend
import gen_sequence

solves = {"a": [], "b": []}      # (sweep, LM iterations) of every registration each side makes


class _Counted(orc.Registration):
    side, sweep = "a", 0

    def align(self, guess=None, max_trace=64):
        T = super().align(guess, max_trace)
        solves[_Counted.side].append((_Counted.sweep, int(self.iterations)))
        return T


orc.Registration = _Counted
trials = int(sys.argv[1]) if len(sys.argv) > 1 else 4
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rep = {"trials": 0, "sweeps": 0, "sweeps_compared": 0, "trials_past_an_unequal_solve": 0, "failures": [], "with_imu": 0, "with_ground": 0, "ground_flag_trips": 0, "keyframes_pushed": 0, "max": {"q": 0.0, "t": 0.0, "fitness": 0.0}}
t0 = time.time()
for trial in range(trials):
    rng = np.random.default_rng(seed0 * 1013904223 % (1 << 32) + trial)
    n_sw, n_az = int(rng.integers(5, 10)), int(rng.integers(400, 900))
    use_imu, use_ground, first = int(rng.random() < 0.5), int(rng.random() < 0.75), int(rng.choice([0, 2, 3]))
    distort = bool(rng.random() < 0.6)
    tseed = int(rng.integers(1, 1 << 20))
    tag = {"trial": trial, "sweeps": n_sw, "n_az": n_az, "imu": use_imu, "ground": use_ground, "firstflagnum": first, "distorted": distort, "seed": tseed}
    try:
        world = synth.make_world(half_extent=45.0, seed=synth.SEED + int(rng.integers(0, 4)))
        poses = synth.make_trajectory(n_sw + 1, seed=tseed)
        imu = synth.make_imu(poses, seed=tseed)
        raws = []
        for k in range(n_sw):
            sc = synth.make_scan(world, poses[k], n_az=n_az, seed=tseed + 70 + k, T_ws_end=poses[k + 1] if distort else None)
            raws.append(np.concatenate([sc["xyz"], sc["intensity"][:, None]], axis=1).astype(np.float32))
        stamps = [0.1 * (k + 1) for k in range(n_sw)]
        node = py_odometer.IcpThread(USE_IMU=use_imu, USE_GROUND=use_ground, firstflagnum=first)
        od = odometry.Odometer(OracleBackend(), use_ground=bool(use_ground), use_imu=bool(use_imu), first_frames=first)
        ra, rb = [], []

        def ha(raw, t_k):
            _Counted.side, _Counted.sweep = "a", len(ra)
            r = node.handle(raw, t_k)
            ra.append((r is not None, node.q_w_curr.copy(), node.t_w_curr.copy(), node.gflag, len(node.surroundingCloud), len(node.laserCloudsubmap), node.vgicp_source, node.submapflag))

        def hb(raw, t_k):
            _Counted.side, _Counted.sweep = "b", len(rb)
            r = od.process(raw, t_k)
            rb.append((r is not None, od.q_w_curr.copy(), od.t_w_curr.copy(), od.gflag, len(od.surrounding), len(od.submap), od.fitness))
        solves["a"].clear(); solves["b"].clear()
        nothing = lambda *a: None     # (USE_IMU = 0: the reference buffers the samples, RGC_odometer.cpp:444-486, and never reads them; the mirror does not take them)
        gen_sequence.feed(node, raws, stamps, imu, node.imuCallback, ha)
        gen_sequence.feed(od, raws, stamps, imu, od.imu_callback if use_imu else nothing, hb)
        # Two solves on inputs that differ by rounding (the fusion solve's last bits, 1e-8) may stop one LM iteration apart; in a flat valley that is
        # 1e-5 .. 1e-4 m, and every later sweep inherits it (EXPERIMENTS.md "flat valley"): the orchestration is compared up to the
        # first such solve -- decisions and poses to rounding -- and the trial counted
        if [x[0] for x in solves["a"]] != [x[0] for x in solves["b"]]:
            rep["failures"].append(dict(tag, error="registrations made", literal=[x[0] for x in solves["a"]], mirror=[x[0] for x in solves["b"]]))
        unequal = [x[0] for x, y in zip(solves["a"], solves["b"]) if x[1] != y[1]]
        stop = min(unequal) if unequal else len(ra)
        rep["trials_past_an_unequal_solve"] += int(bool(unequal))
        rep["sweeps_compared"] += min(stop, len(ra))
        for i, (a, b) in enumerate(zip(ra[:stop], rb[:stop])):
            if a[0] != b[0] or a[3] != b[3] or a[4] != b[4] or a[5] != b[5]:
                rep["failures"].append(dict(tag, sweep=i, error="decisions", literal=[bool(a[0]), int(a[3]), a[4], a[5]], mirror=[bool(b[0]), int(b[3]), b[4], b[5]]))
                break
            eq, et = float(min(np.abs(a[1] - b[1]).max(), np.abs(a[1] + b[1]).max())), float(np.abs(a[2] - b[2]).max())
            rep["max"]["q"], rep["max"]["t"] = max(rep["max"]["q"], eq), max(rep["max"]["t"], et)
            # rounding only -- but rounding of a guess (fp32: an ulp is 4e-9) moves where an LM trajectory stops by up to its stopping tolerance
            # (translation_eps 1e-6) per sweep, and world poses accumulate it: the bar is 2e-5 m / 2e-6 over these 5-9 sweeps (north_star's is 1e-4 m per
            # frame; seen: 7e-6 once in 200 sequences, 1e-6 otherwise); the committed fixture's sequence holds 1e-9
            if not (eq <= 2e-6 and et <= 2e-5):
                rep["failures"].append(dict(tag, sweep=i, error="pose", q=eq, t=et))
                break
            if a[0] and a[7] > 0:
                ef = float(abs(a[6] - b[6]))
                rep["max"]["fitness"] = max(rep["max"]["fitness"], ef)
                if not ef <= 1e-4 * abs(a[6]):      # (a nearest-neighbour sum of squares: 2 d delta for a pose delta inside the bar above)
                    rep["failures"].append(dict(tag, sweep=i, error="fitness", literal=float(a[6]), mirror=float(b[6])))
                    break
        rep["sweeps"] += len(ra)
        rep["with_imu"] += use_imu; rep["with_ground"] += use_ground
        g = [r[3] for r in ra]
        rep["ground_flag_trips"] += int(sum(1 for i in range(1, len(g)) if g[i] != g[i - 1]))
        rep["keyframes_pushed"] += int(max(r[4] for r in ra))
    except Exception as e:
        import traceback
        rep["failures"].append(dict(tag, error="exception: %r" % (e,), where=traceback.format_exc()[-700:]))
    rep["trials"] += 1
    if len(rep["failures"]) > 8:
        break
rep["wall_s"] = round(time.time() - t0, 1)
print(json.dumps(rep))
