#!/usr/bin/env python3
"""bench.py -- registered scans/s of the HIP scan-to-map registration path (BASELINE.json metric).

One "step" = one complete registration of a synthetic VLP-16 scan (30 k points) against a 1 M-point local map,
exactly what the reference does per frame at RGC_odometer.cpp:998-1011: target covariances + Gaussian voxel map
rebuilt from scratch (the reference re-creates FastVGICP every frame), source covariances, LM solve, fitness.
Inputs (maps and scans) are resident in HBM before the timed region starts; nothing is cached across steps.
The reference re-frames its sub-map every frame (RGC_odometer.cpp:1248-1256), so consecutive steps see maps with different
bounding boxes: the timed loop alternates between three translated copies of the map (the library's speculative grid has to
earn its hits).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--configs c1,c3[,c5]]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

N > 1: one process per GPU, one independent sequence per rank (different seed), no data-path collective
(the path shards across sequences); torch.distributed (RCCL) is used only for the barrier and the
MAX over ranks of the elapsed time.  Rank 0 prints ONE JSON line.

Extra keys beside the contract's: `scan_h2d_and_output` (the same loop with each scan uploaded from pinned host memory inside
the timed step and align()'s output cloud produced on the device: never `value`), `issue_roofline` (the dominant kernel against
the measured VALU issue rates of profiles/r02_valu_issue.jsonl), `configs` (BASELINE.json's other single-GPU configurations,
a few frames each).
"""
import argparse
import gc
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

N_SOURCE = 30000
N_TARGET = 1000000
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
MAP_SHIFTS = ((0.0, 0.0, 0.0), (0.37, -0.23, 0.011), (-0.29, 0.41, -0.007))  # m: the map copies the timed loop alternates between


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def algorithmic_bytes(n_s, n_t, n_vox, n_corr, n_lin, n_err):
    """SURVEY.md §8d yardstick (fixed; fused stages may not shrink it)."""
    return 36.0 * n_s + 36.0 * n_t + (36.0 * n_t + 40.0 * n_vox) + (n_lin + n_err) * (36.0 * n_s + 40.0 * n_corr) + 24.0 * n_s


def rot_angle(Ra, Rb):
    import numpy as np
    R = Ra.astype(np.float64) @ Rb.astype(np.float64).T
    w = 0.5 * np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    return float(np.arcsin(min(1.0, np.linalg.norm(w))))


def pin_to_gpu_numa_node(local_rank):
    """Host thread on the cores of the GPU's NUMA node (8 ranks fed by one host: keep each rank's launches local).  Best effort."""
    try:
        import torch
        p = torch.cuda.get_device_properties(local_rank)
        bdf = f"{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"
        node = int(open(f"/sys/bus/pci/devices/{bdf}/numa_node").read())
        if node < 0:
            return None
        cpus = set()
        for part in open(f"/sys/devices/system/node/node{node}/cpulist").read().strip().split(","):
            lo, _, hi = part.partition("-")
            cpus.update(range(int(lo), int(hi or lo) + 1))
        cpus &= os.sched_getaffinity(0)
        if cpus:
            os.sched_setaffinity(0, cpus)
            return node
    except Exception:
        pass
    return None


def shifted(T, d, sign):
    """pose of a scan in a map translated by -d: translate(sign * d) * T"""
    import numpy as np
    out = np.array(T, dtype=np.float32, copy=True)
    out[:3, 3] += sign * np.asarray(d, np.float32)
    return out


def time_config(registration, oracle, name, tgt, scans, guess0, prior=None):
    """scans[i] registered to tgt (rebuilt per frame, inputs resident); frame 0 is the warm-up (once per context); frame 1 is checked against
    the CPU oracle.  Timed twice: the two-context pipeline (the figure) and one frame at a time."""
    import numpy as np
    pv = registration.PipelinedVGICP(int(os.environ.get("LOCAL_RANK", "0")), depth=2)
    v = pv.v[0]

    def to_dev(xyz):
        a = np.zeros((xyz.shape[0], 4), np.float32)
        a[:, :3] = xyz
        p = v.device_alloc(a.nbytes)
        v.upload(p, a)
        return p
    d_tgt, d_s = to_dev(tgt), [to_dev(s) for s in scans]

    def setc(i, w):
        w.setInputTargetDevice(d_tgt, len(tgt), 16)
        w.setInputSourceDevice(d_s[i], len(scans[i]), 16)
    frames = len(scans) - 1
    for w in pv.v:
        setc(0, w)
        w.align(prior[0] if prior else guess0, want_output=False, want_fitness=True)
    g_in1 = prior[1] if prior else v.getFinalTransformation()
    pv.synchronize()
    # a cyclic-GC pass over torch's object graph (40-65 ms, scripts/exp_stall.py) inside a 3-frame timed region quarters the figure:
    # the garbage made so far (data generation, the oracle's arrays) is collected here and the survivors are frozen, as before the main loop
    import gc
    gc.collect()
    gc.freeze()
    t1 = time.perf_counter()
    g, fin_seq = g_in1, []
    trace = os.environ.get("RGC_BENCH_TRACE")  # developer aid: where a frame's time goes, per call
    for i in range(1, frames + 1):
        ta = time.perf_counter()
        setc(i, v)
        tb = time.perf_counter()
        v.align(prior[i] if prior else g, want_output=False, want_fitness=True)
        g = v.getFinalTransformation()
        fin_seq.append(g)
        if trace:
            print(f"[bench] {name[:2]} frame {i}: set clouds {1e3 * (tb - ta):.2f} ms, align {1e3 * (time.perf_counter() - tb):.2f} ms", file=sys.stderr)
    v.synchronize()
    el_seq = time.perf_counter() - t1
    t1 = time.perf_counter()
    fin = pv.run(frames, lambda j, w: setc(j + 1, w), g_in1, want_fitness=True,
                 next_guess=(lambda j, T: prior[min(j + 2, frames)]) if prior else None)
    pv.synchronize()
    el = time.perf_counter() - t1
    st = pv.v[(frames - 1) % 2].stats()
    out = {"config": name, "n_source": int(len(scans[0])), "n_target": int(len(tgt)), "frames": frames, "scans_per_s": round(frames / el, 2),
           "ms_per_scan": round(1e3 * el / frames, 3), "one_frame_at_a_time_scans_per_s": round(frames / el_seq, 2),
           "same_poses_both_ways": bool(all(np.array_equal(x, y) for x, y in zip(fin, fin_seq))), "outer_iterations_last": st["outer_iterations"]}
    if oracle is not None:
        o = oracle.Registration(num_threads=min(14, os.cpu_count() or 1))  # the reference's setNumThreads(14): also the oracle's faster setting
        c0 = time.perf_counter()
        o.set_target(tgt)
        o.set_source(scans[1])
        To = o.align(g_in1)
        _ = o.fitness()
        out["cpu_oracle_scans_per_s"] = round(1.0 / (time.perf_counter() - c0), 4)
        out["max_dt_m"] = float(np.abs(fin[0][:3, 3] - To[:3, 3]).max())
        out["max_dtheta_rad"] = rot_angle(fin[0][:3, :3], To[:3, :3])
    pv.close()
    return out


def run_extra_configs(registration, synth, oracle, keys):
    """BASELINE.json's other single-GPU configurations, a few frames each (scripts/bench_configs.py runs them at length):
    c1 30 k vs 100 k; c3 HDL-64 130 k vs 5 M; c5 two interleaved 64-beam patterns 250 k vs 20 M (four copies of the 5 M tile) with the
    true pose corrupted by ~0.5 deg of rotation as the IMU-like prior."""
    import numpy as np
    res = []
    def guarded(name, fn):
        try:
            t0 = time.time()
            r = fn()
            r["wall_s_incl_datagen"] = round(time.time() - t0, 1)
            res.append(r)
        except Exception as e:  # the metric line must not be lost to a side configuration
            res.append({"config": name, "error": str(e)[:200]})
    if "c1" in keys:
        def c1():
            world, tgt = synth.make_world_and_map(100000, seed=synth.SEED)
            poses = synth.make_trajectory(12, seed=synth.SEED)
            scans = [synth.make_scan_n(world, poses[i + 1], 30000, seed=synth.SEED + 100 + i)["xyz"] for i in range(11)]
            return time_config(registration, oracle, "c1: VLP-16 30 k-pt scans vs 100 k-pt fixed map", tgt, scans, poses[0].astype(np.float32))
        guarded("c1", c1)
    if "c3" in keys or "c5" in keys:
        world = tile = None
        try:
            world, tile = synth.make_world_and_map(5_000_000, seed=synth.SEED + 7)
        except Exception as e:
            res.append({"config": "c3/c5", "error": str(e)[:200]})
        e64 = synth.hdl64_elev()
        if tile is not None and "c3" in keys:
            def c3():
                poses = synth.make_trajectory(8, seed=synth.SEED + 7)
                scans = [synth.make_scan_n(world, poses[i + 1], 130000, elev_deg=e64, seed=synth.SEED + 200 + i)["xyz"] for i in range(6)]
                return time_config(registration, oracle, "c3: HDL-64 130 k-pt scans vs 5 M-pt map", tile, scans, poses[0].astype(np.float32))
            guarded("c3", c3)
        if tile is not None and "c5" in keys:
            def c5():
                L = 2.0 * world.half_extent + 4.0
                tgt = np.concatenate([tile, tile + np.float32([L, 0, 0]), tile + np.float32([0, L, 0]), tile + np.float32([L, L, 0])]).astype(np.float32)
                poses = synth.make_trajectory(6, seed=synth.SEED + 9)
                from rgc_slam_amd import odometry
                # the guesses the odometer forms with USE_IMU = 1 (RGC_odometer.cpp:929-931, 993-996): the gyro's pre-integrated rotation over
                # the sweep (synthetic 200 Hz IMU stream through rgc_imu_preintegrate) and the previous sweep's translation
                imu_prior = odometry.imu_rotation_priors(poses)
                scans, prior = [], []
                for i in range(4):
                    a = synth.make_scan_n(world, poses[i + 1], 125000, elev_deg=e64, seed=synth.SEED + 300 + i)["xyz"]
                    b = synth.make_scan_n(world, poses[i + 1], 125000, elev_deg=e64 + 0.5 * float(np.abs(np.diff(np.sort(e64))).min()),
                                          seed=synth.SEED + 400 + i)["xyz"]
                    scans.append(np.concatenate([a, b]).astype(np.float32))
                    prior.append(imu_prior[i + 1])
                return time_config(registration, oracle, "c5: 2 x 64-beam 250 k-pt scans vs 20 M-pt map, IMU-preintegrated prior", tgt, scans, None, prior=prior)
            guarded("c5", c5)
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--n-target", type=int, default=N_TARGET)
    ap.add_argument("--n-source", type=int, default=N_SOURCE)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--configs", default="c1,c3", help="extra single-GPU configurations of BASELINE.json to run after the metric (c1,c3,c5 or 'none')")
    args = ap.parse_args()

    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world_size == 1:
        # not launched through torch.distributed.run: start it as a child (nothing here has touched the GPU yet)
        port = 29500 + (os.getpid() % 2000)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))

    import numpy as np
    import torch  # first: its bundled HIP runtime must be the one librgc_hip.so binds to in this process
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    torch.cuda.set_device(local_rank)
    numa = pin_to_gpu_numa_node(local_rank)
    if world_size > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world_size, device_id=torch.device("cuda", local_rank))

    import rgc_slam_amd.synth as synth
    from rgc_slam_amd import registration

    K, W = args.steps, args.warmup
    seed = synth.SEED + rank  # one independent sequence per rank
    t0 = time.time()
    world, tgt = synth.make_world_and_map(args.n_target, seed=seed)
    poses = synth.make_trajectory(K + W + 1, seed=seed)
    scans = [synth.make_scan_n(world, poses[i + 1], args.n_source, seed=seed + 100 + i)["xyz"] for i in range(K + W)]
    maps = [(tgt - np.asarray(d, np.float32)).astype(np.float32) for d in MAP_SHIFTS]
    log(f"[rank {rank}] synthetic data: map {tgt.shape} x {len(maps)} copies, {len(scans)} scans of {scans[0].shape[0]} pts, world half-extent "
        f"{world.half_extent:.1f} m, {time.time() - t0:.1f} s, NUMA node {numa}")

    # Two contexts take turns (registration.PipelinedVGICP): while frame i is solved on one, frame i + 1's clouds are prepared on the
    # other.  Same kernels, same inputs, same poses as one frame at a time (measured below as well, and compared).
    pv = registration.PipelinedVGICP(local_rank, depth=2)
    v = pv.v[0]
    # inputs resident in HBM (x,y,z,pad; 16-byte stride) before anything is timed
    def to_dev(xyz):
        a = np.zeros((xyz.shape[0], 4), np.float32)
        a[:, :3] = xyz
        p = v.device_alloc(a.nbytes)
        v.upload(p, a)
        return p
    d_maps = [to_dev(m) for m in maps]
    d_scans = [to_dev(s) for s in scans]
    # the second loop's inputs: every scan in pinned host memory (x,y,z,pad), one device buffer per context for align()'s output cloud
    pinned = []
    for s in scans:
        t = torch.zeros((s.shape[0], 4), dtype=torch.float32).pin_memory()
        t[:, :3] = torch.from_numpy(s)
        pinned.append(t)
    d_aligned = {id(w): w.device_alloc(16 * args.n_source) for w in pv.v}

    def to_map(i, T_world):   # world pose -> pose in the (translated) map copy frame i registers to
        return shifted(T_world, MAP_SHIFTS[i % len(maps)], -1.0)

    def to_world(i, T):
        return shifted(T, MAP_SHIFTS[i % len(maps)], +1.0)

    def set_clouds(i, w, from_host=False):
        w.setInputTargetDevice(d_maps[i % len(maps)], tgt.shape[0], 16)   # full per-frame rebuild, like the reference
        if from_host:
            w.setInputSource(pinned[i].numpy())                           # H2D inside the step (pinned host memory)
        else:
            w.setInputSourceDevice(d_scans[i], scans[i].shape[0], 16)

    def run_pipelined(first, count, guess_world, from_host=False, collect=None):
        """frames first .. first + count - 1 through the pipeline; returns the world poses"""
        def done(j, w):
            if from_host:
                w.alignedToDevice(d_aligned[id(w)], 16)                   # pcl::transformPointCloud(*input_, output, final), left on the device
            if collect is not None:
                st = w.stats()
                collect.append((st["outer_iterations"], st["n_linearize"], st["n_error"], st["n_corr"], st["n_voxels"]))
        Ts = pv.run(count, lambda j, w: set_clouds(first + j, w, from_host), to_map(first, guess_world), want_fitness=True,
                    next_guess=lambda j, T: to_map(first + j + 1, to_world(first + j, T)), on_result=done)
        return [to_world(first + j, T) for j, T in enumerate(Ts)]

    def step(i, guess_world):   # one frame at a time on one context
        set_clouds(i, v)
        v.align(to_map(i, guess_world), want_output=False, want_fitness=True)
        return to_world(i, v.getFinalTransformation())

    finals, per_frame = [], []
    guess = poses[0].astype(np.float32)
    for w in pv.v:   # context start-up (first allocations, the first cloud's bounding-box round trip): frame 0 once on each, untimed
        set_clouds(0, w)
        w.align(to_map(0, guess), want_output=False, want_fitness=True)
    # The one-off ~40 ms stall that earlier rounds hid behind 96 untimed frames is CPython's cyclic garbage collector doing a full
    # collection over torch's object graph (scripts/exp_stall.py: gone with gc.freeze(), unmoved by anything done to the HIP
    # runtime): it belongs to this harness, not to the path.  Freeze what exists; the loops below allocate nothing cyclic.
    # (Before the warm-up, so that the timed loop follows it directly: 50 ms of host-only work would let the GPU's clocks drop.)
    gc.collect()
    gc.freeze()
    # HIP-event regions cost two hipEventRecord each: in the timed loop only the dominant kernel (the map's bulk kNN +
    # covariance launch -- rocprofv3 agrees, profiles/) is bracketed; the other stages are timed in a separate pass below.
    DOMINANT = "knn_cov_target"
    for w in pv.v:
        w.profile_enable(True)
        w.profile_select([DOMINANT])
    if W > 0:
        guess = run_pipelined(0, W, guess)[-1]
    pv.synchronize()
    for w in pv.v:
        w.profile_reset()
    if world_size > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t_start = time.perf_counter()
    finals = run_pipelined(W, K, guess, collect=per_frame)
    pv.synchronize()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t_start
    guess_in = [guess] + finals[:-1]
    if os.environ.get("RGC_BENCH_AGAIN"):  # developer aid: the same timed loop again, with and without the dominant kernel's event pair
        for sel in ([DOMINANT], []):
            for w in pv.v:
                w.profile_select(sel)
            t_a = time.perf_counter()
            run_pipelined(W, K, guess)
            pv.synchronize()
            log(f"[again] events on {sel}: {K / (time.perf_counter() - t_a):.1f} scans/s")
        for w in pv.v:
            w.profile_select([DOMINANT])
    if world_size > 1:
        dist.barrier()
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    prof_dom = {"total_ms": 0.0, "launches": 0, "points": 0}
    for w in pv.v:
        d = w.profile()[DOMINANT]
        for kk in prof_dom:
            prof_dom[kk] += d[kk]
        w.profile_enable(False)
    # the same K frames one at a time on one context (what a caller of the blocking align() gets: the frame's latency)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    g1, seq_finals = guess_in[0], []
    for i in range(W, W + K):
        g1 = step(i, g1)
        seq_finals.append(g1)
    v.synchronize()
    elapsed_seq = time.perf_counter() - t1
    seq_same = bool(all(np.array_equal(a_, b_) for a_, b_ in zip(finals, seq_finals)))
    # the same K steps with the scan crossing PCIe inside the step and the output cloud produced (device-resident): an extra key
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    g2 = run_pipelined(W, K, guess_in[0], from_host=True)[-1]
    pv.synchronize()
    elapsed_h2d = time.perf_counter() - t2
    h2d_same = bool(np.array_equal(g2, finals[-1]))
    # the dominant kernel by itself (nothing else on the GPU): what the kernel costs, as opposed to what it costs while it shares the chip
    v.profile_enable(True)
    v.profile_select([DOMINANT])
    v.profile_reset()
    for j in range(5):
        v.setInputTargetDevice(d_maps[j % len(maps)], tgt.shape[0], 16)
        v.synchronize()
    dom_alone = v.profile()[DOMINANT]
    v.profile_enable(False)
    # per-stage breakdown: a few more frames, one at a time, with every region bracketed (untimed, informational)
    v.profile_enable(True)
    v.profile_select(None)
    v.profile_reset()
    KB = min(K, 5)
    for j in range(KB):  # the same frames with the same initial guesses as the timed loop
        step(W + j, guess_in[j])
    v.synchronize()
    prof = v.profile()
    v.profile_enable(False)

    if rank != 0:
        if world_size > 1:
            dist.destroy_process_group()
        return

    pf = np.asarray(per_frame, dtype=np.float64)
    mean_outer, mean_lin, mean_err, mean_corr, n_vox = pf.mean(axis=0)
    scans_per_s = K * world_size / elapsed
    B = algorithmic_bytes(args.n_source, args.n_target, n_vox, mean_corr, mean_lin, mean_err)

    # dominant kernel: checked against the all-stages pass (largest summed HIP-event time), measured over the timed region
    name = max(prof.items(), key=lambda kv: kv[1]["total_ms"])[0]
    if name != DOMINANT:
        log(f"warning: the breakdown pass names {name} as dominant, the timed region bracketed {DOMINANT}")
    name, d = DOMINANT, prof_dom
    per_unit = 36.0  # 12 B read + 24 B written per point of the kNN / covariance stage
    avg_ms = d["total_ms"] / max(d["launches"], 1)
    units = d["points"] / max(d["launches"], 1)
    achieved = per_unit * units / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    traffic = None
    pmc = None
    pfile = os.path.join(ROOT, "profiles", "r02_pmc_knn.json")  # rocprofv3 --pmc passes of this kernel (scripts/pmc_kernel.sh), if committed
    if os.path.exists(pfile):
        try:
            pmc = json.load(open(pfile))
            traffic = pmc.get("hbm_bytes_per_launch")
        except Exception:
            pmc = None
    # The same kernel against the roof that binds it -- VALU instruction issue.  Peaks are MEASURED (scripts/ubench/valu_issue.hip,
    # profiles/r02_valu_issue.jsonl, 8 waves per SIMD, every CU): add / sub / mul / fma / and / or / mov issue at ~1060 G
    # wave-instructions/s chip-wide (the 2-cycles-per-wave64 figure of the guide, 1229 G/s at 2.4 GHz, less the clock held under
    # load); min / max / med3 / compare / select / shifts / three-operand integer ops and ALL fp64 at ~595 G/s -- half rate.  The
    # kernel's selection work is in the second class.
    issue = None
    if pmc and avg_ms > 0 and pmc.get("valu_wave_instructions_per_query"):
        per_q = float(pmc["valu_wave_instructions_per_query"])
        ach = per_q * units / (avg_ms * 1e-3) / 1e9
        issue = {"bound": "valu_issue", "kernel": name, "achieved": round(ach, 1), "unit": "G wave-instr/s", "valu_wave_instructions_per_query": per_q,
                 "peak_full_rate_measured": 1060.0, "peak_half_rate_measured": 595.0, "peak_2cyc_at_2.4GHz": 1228.8,
                 "frac_of_half_rate_peak": round(ach / 595.0, 4), "frac_of_full_rate_peak": round(ach / 1060.0, 4)}
    roofline = {"bound": "hbm", "kernel": name, "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic,
                "avg_launch_ms": round(avg_ms, 4), "algorithmic_bytes_per_launch": per_unit * units}
    alone_ms = dom_alone["total_ms"] / max(dom_alone["launches"], 1)
    if alone_ms > 0:  # the timed region runs the launch beside another frame's kernels; alone it is shorter
        roofline["launch_alone_ms"] = round(alone_ms, 4)
        roofline["frac_launch_alone"] = round(per_unit * units / (alone_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 6)

    out = {
        "metric": "registered scans/sec (16-beam -> 1M-pt map)", "value": round(scans_per_s, 3), "unit": "scans/s",
        "n_gpus": world_size, "steps": K, "warmup": W, "ms_per_step": round(1e3 * elapsed / K, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32 points and neighbour search, f64 covariances and solve",
        "data": "synthetic",
        "config": {"workload": f"c-main: synthetic VLP-16 {args.n_source}-pt scans registered to a {args.n_target}-pt local map, rebuilt every step, "
                               f"{len(maps)} translated map copies in turn (BASELINE.md c-main; one independent sequence per GPU; two contexts "
                               f"take turns so that a frame's preparation overlaps the previous frame's solve)",
                   "n_source": args.n_source, "n_target": args.n_target, "voxel_res": 1.0, "k": 20, "max_iterations": 25,
                   "parallelism": f"sequences x{world_size}"},
        "algorithmic_bytes_per_scan": round(B), "hbm_gbps_algorithmic": round(B * scans_per_s / world_size / 1e9, 3),
        "hbm_frac_whole_frame": round(B * scans_per_s / world_size / 1e9 / HBM_PEAK_GBS, 6),
        "mean_outer_iterations": round(mean_outer, 2), "mean_linearize": round(mean_lin, 2), "mean_compute_error": round(mean_err, 2),
        "mean_correspondences": round(mean_corr, 1), "n_voxels": int(n_vox),
        "kernel_ms_per_step": {k: round(x["total_ms"] / KB, 4) for k, x in prof.items()},
        "roofline": roofline, "issue_roofline": issue,
        "one_frame_at_a_time": {"scans_per_s": round(K / elapsed_seq, 3), "ms_per_step": round(1e3 * elapsed_seq / K, 3), "same_poses": seq_same,
                                "what": "the same K frames through the blocking align() on one context: a frame's latency"},
        "scan_h2d_and_output": {"scans_per_s": round(K / elapsed_h2d, 3), "ms_per_step": round(1e3 * elapsed_h2d / K, 3), "same_final_pose": h2d_same,
                                "what": "same steps; each scan uploaded from pinned host memory inside the step, align()'s output cloud written to a device buffer"},
        "final_pose_checksum": float(np.sum(np.abs(np.asarray(finals, np.float64)))),
    }

    oracle = None
    if world_size == 1 and not args.no_cpu_baseline:
        # CPU baseline: the oracle (a port; the reference itself cannot be built here) on this box's host cores,
        # on a bounded sample of the same workload; also the parity check of those frames.
        from oracle import oracle
        cores = os.cpu_count() or 1
        # at the reference's own thread count (setNumThreads(14), RGC_odometer.cpp:1006): on a many-core host more OpenMP threads make
        # this path SLOWER (its parallel loops are short), so this is also the faster setting; one frame on all cores is timed beside it
        nthr = min(14, cores)
        o = oracle.Registration(num_threads=nthr)
        n_done, t_cpu, dts, dths = 0, 0.0, [], []
        for j in range(K):
            i = W + j
            m = i % len(maps)
            c0 = time.perf_counter()
            o.set_target(maps[m])
            o.set_source(scans[i])
            To = shifted(o.align(shifted(guess_in[j], MAP_SHIFTS[m], -1.0)), MAP_SHIFTS[m], +1.0)
            _ = o.fitness()
            t_cpu += time.perf_counter() - c0
            Tg = finals[j]
            dts.append(float(np.abs(Tg[:3, 3] - To[:3, 3]).max()))
            dths.append(rot_angle(Tg[:3, :3], To[:3, :3]))
            n_done += 1
            if t_cpu > 20.0:  # bounded: all timed frames (~0.25 s each) or 20 s of CPU work
                break
        out["cpu_baseline"] = {"value": round(n_done / t_cpu, 4), "unit": "scans/s", "cores": nthr, "kind": "port",
                               "sample": f"{n_done} frame(s) of the same workload (first timed frames, each from the GPU path's own guess), "
                                         f"OpenMP x{nthr} (the reference's setNumThreads), {t_cpu:.1f} s of CPU work"}
        if cores > nthr:
            oa = oracle.Registration(num_threads=cores)
            c0 = time.perf_counter()
            oa.set_target(maps[W % len(maps)])
            oa.set_source(scans[W])
            oa.align(shifted(guess_in[0], MAP_SHIFTS[W % len(maps)], -1.0))
            _ = oa.fitness()
            ta = time.perf_counter() - c0
            out["cpu_baseline"]["value_all_cores"] = round(1.0 / ta, 4)
            out["cpu_baseline"]["sample"] += f"; 1 frame at {cores} threads, {ta:.1f} s"
        out["pose_parity_vs_cpu"] = {"frames": n_done, "max_dt_m": max(dts), "max_dtheta_rad": max(dths),
                                     "rmse_dt_m": float(np.sqrt(np.mean(np.square(dts)))),
                                     "rmse_dtheta_rad": float(np.sqrt(np.mean(np.square(dths))))}
    pv.close()

    if world_size == 1 and args.configs != "none":
        out["configs"] = run_extra_configs(registration, synth, oracle, [c.strip() for c in args.configs.split(",") if c.strip()])

    print(json.dumps(out), flush=True)
    if world_size > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
