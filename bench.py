#!/usr/bin/env python3
"""bench.py -- registered scans/s of the HIP scan-to-map registration path (BASELINE.json metric).

One "step" = one complete registration of a synthetic VLP-16 scan (30 k points) against a 1 M-point local map,
exactly what the reference does per frame at RGC_odometer.cpp:998-1011: target covariances + Gaussian voxel map
rebuilt from scratch (the reference re-creates FastVGICP every frame), source covariances, LM solve, fitness.
Inputs (map and scans) are resident in HBM before the timed region starts; nothing is cached across steps.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

N > 1: one process per GPU, one independent sequence per rank (different seed), no data-path collective
(the path shards across sequences); torch.distributed (RCCL) is used only for the barrier and the
MAX over ranks of the elapsed time.  Rank 0 prints ONE JSON line.
"""
import argparse
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--n-target", type=int, default=N_TARGET)
    ap.add_argument("--n-source", type=int, default=N_SOURCE)
    ap.add_argument("--no-cpu-baseline", action="store_true")
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
    log(f"[rank {rank}] synthetic data: map {tgt.shape}, {len(scans)} scans of {scans[0].shape[0]} pts, world half-extent "
        f"{world.half_extent:.1f} m, {time.time() - t0:.1f} s")

    v = registration.odometer_vgicp(local_rank)
    # inputs resident in HBM (x,y,z,pad; 16-byte stride) before anything is timed
    def to_dev(xyz):
        a = np.zeros((xyz.shape[0], 4), np.float32)
        a[:, :3] = xyz
        p = v.device_alloc(a.nbytes)
        v.upload(p, a)
        return p
    d_tgt = to_dev(tgt)
    d_scans = [to_dev(s) for s in scans]

    finals, per_frame = [], []

    def step(i, guess):
        v.setInputTargetDevice(d_tgt, tgt.shape[0], 16)          # full per-frame rebuild, like the reference
        v.setInputSourceDevice(d_scans[i], scans[i].shape[0], 16)
        v.align(guess, want_output=False, want_fitness=True)
        return v.getFinalTransformation()

    # Process spin-up, untimed and outside W: a fresh process pays ONE ~40 ms stall inside the HIP runtime at its ~84th frame
    # (a runtime pool growing once; measured with scripts/exp_bench_overhead.py: frame 83 exactly, never again in 600 frames).
    # A 10 Hz node never notices it; a 20-step timed loop would report it as a 3x slowdown if it fell inside.
    SPINUP = int(os.environ.get("RGC_BENCH_SPINUP", "96"))
    g0 = poses[0].astype(np.float32)
    for j in range(SPINUP):
        step(j % max(W, 1), g0)
    v.synchronize()
    guess = poses[0].astype(np.float32)
    for i in range(W):
        guess = step(i, guess)
    v.synchronize()
    # HIP-event regions cost two hipEventRecord each: in the timed loop only the dominant kernel (the map's bulk kNN +
    # covariance launch -- rocprofv3 agrees, profiles/) is bracketed; the other stages are timed in a separate pass below.
    DOMINANT = "knn_cov_target"
    v.profile_enable(True)
    v.profile_select([DOMINANT])
    v.profile_reset()
    if world_size > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t_start = time.perf_counter()
    guess_in = []
    for i in range(W, W + K):
        guess_in.append(guess)
        guess = step(i, guess)
        finals.append(guess)
        st = v.stats()
        per_frame.append((st["outer_iterations"], st["n_linearize"], st["n_error"], st["n_corr"], st["n_voxels"]))
    v.synchronize()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t_start
    if world_size > 1:
        dist.barrier()
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    prof_dom = v.profile()[DOMINANT]
    # per-stage breakdown: a few more frames with every region bracketed (untimed, informational)
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
    per_unit = {"knn_cov_target": 36.0, "knn_cov_source": 36.0, "knn_coop_target": 36.0, "knn_coop_source": 36.0, "voxel_build": 36.0 + 40.0 * n_vox / args.n_target,
                "linearize": 36.0 + 40.0 * mean_corr / args.n_source, "compute_error": 36.0 + 40.0 * mean_corr / args.n_source,
                "fitness": 24.0, "grid_build": 0.0}[name]
    avg_ms = d["total_ms"] / max(d["launches"], 1)
    units = d["points"] / max(d["launches"], 1)
    achieved = per_unit * units / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    traffic = None
    tfile = os.path.join(ROOT, "profiles", "pmc_traffic.json")  # measured HBM bytes per launch (rocprofv3 --pmc), if committed
    if os.path.exists(tfile):
        try:
            traffic = json.load(open(tfile)).get(name)
        except Exception:
            traffic = None
    # the same kernel against the roof that actually binds it -- VALU instruction issue: wave-instructions per query from the
    # committed PMC pass (profiles/r01_pmc_issue.json, rocprofv3 --pmc SQ_INSTS_VALU) / this run's launch duration, against
    # 256 CUs x 4 SIMDs x one wave64 VALU instruction per 4 cycles at 2.4 GHz
    issue = None
    ifile = os.path.join(ROOT, "profiles", "r01_pmc_issue.json")
    if name == "knn_cov_target" and os.path.exists(ifile) and avg_ms > 0:
        try:
            per_q = float(json.load(open(ifile))["valu_wave_instructions_per_query"])
            ach = per_q * units / (avg_ms * 1e-3) / 1e9
            peak = 256 * 4 * 2.4 / 4.0
            issue = {"bound": "valu_issue", "kernel": name, "achieved": round(ach, 1), "peak": round(peak, 1), "unit": "G wave-instr/s",
                     "frac": round(ach / peak, 4), "valu_wave_instructions_per_query": per_q}
        except Exception:
            issue = None
    roofline = {"bound": "hbm", "kernel": name, "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic,
                "avg_launch_ms": round(avg_ms, 4), "algorithmic_bytes_per_launch": per_unit * units}

    out = {
        "metric": "registered scans/sec (16-beam -> 1M-pt map)", "value": round(scans_per_s, 3), "unit": "scans/s",
        "n_gpus": world_size, "steps": K, "warmup": W, "ms_per_step": round(1e3 * elapsed / K, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"c-main: synthetic VLP-16 {args.n_source}-pt scans registered to a {args.n_target}-pt local map "
                               f"(BASELINE.md c-main; one independent sequence per GPU)",
                   "n_source": args.n_source, "n_target": args.n_target, "voxel_res": 1.0, "k": 20, "max_iterations": 25,
                   "parallelism": f"sequences x{world_size}"},
        "algorithmic_bytes_per_scan": round(B), "hbm_gbps_algorithmic": round(B * scans_per_s / world_size / 1e9, 3),
        "hbm_frac_whole_frame": round(B * scans_per_s / world_size / 1e9 / HBM_PEAK_GBS, 6),
        "mean_outer_iterations": round(mean_outer, 2), "mean_linearize": round(mean_lin, 2), "mean_compute_error": round(mean_err, 2),
        "mean_correspondences": round(mean_corr, 1), "n_voxels": int(n_vox),
        "kernel_ms_per_step": {k: round(x["total_ms"] / KB, 4) for k, x in prof.items()},
        "roofline": roofline, "issue_roofline": issue, "spinup_frames": SPINUP,
    }

    if world_size == 1 and not args.no_cpu_baseline:
        # CPU baseline: the oracle (a port; the reference itself cannot be built here) on this box's host cores,
        # on a bounded sample of the same workload; also the parity check of those frames.
        from oracle import oracle
        cores = os.cpu_count() or 1
        o = oracle.Registration(num_threads=cores)
        n_done, t_cpu, dts, dths = 0, 0.0, [], []
        g = poses[0].astype(np.float32) if W == 0 else None
        # replay from the first timed frame with the GPU's own guess so both paths see identical inputs
        gpu_guess = poses[0].astype(np.float32)
        for i in range(W):
            gpu_guess = step(i, gpu_guess)
        for j in range(K):
            i = W + j
            c0 = time.perf_counter()
            o.set_target(tgt)
            o.set_source(scans[i])
            To = o.align(gpu_guess)
            _ = o.fitness()
            t_cpu += time.perf_counter() - c0
            Tg = finals[j]
            dts.append(float(np.abs(Tg[:3, 3] - To[:3, 3]).max()))
            dths.append(rot_angle(Tg[:3, :3], To[:3, :3]))
            gpu_guess = Tg
            n_done += 1
            if t_cpu > 20.0:  # bounded: all timed frames (~0.3 s each on 256 host threads) or 20 s of CPU work
                break
        out["cpu_baseline"] = {"value": round(n_done / t_cpu, 4), "unit": "scans/s", "cores": cores, "kind": "port",
                               "sample": f"{n_done} frame(s) of the same workload (first timed frames), OpenMP x{cores}, "
                                         f"{t_cpu:.1f} s of CPU work"}
        # the reference hard-codes 14 OpenMP threads (RGC_odometer.cpp:1006): one frame of the same workload at that setting
        o14 = oracle.Registration(num_threads=min(14, cores))
        c0 = time.perf_counter()
        o14.set_target(tgt)
        o14.set_source(scans[W])
        o14.align(guess_in[0])
        _ = o14.fitness()
        t14 = time.perf_counter() - c0
        out["cpu_baseline"]["value_14_threads"] = round(1.0 / t14, 4)
        out["cpu_baseline"]["sample"] += f"; 1 frame at {min(14, cores)} threads (the reference's setNumThreads), {t14:.1f} s"
        out["pose_parity_vs_cpu"] = {"frames": n_done, "max_dt_m": max(dts), "max_dtheta_rad": max(dths),
                                     "rmse_dt_m": float(np.sqrt(np.mean(np.square(dts)))),
                                     "rmse_dtheta_rad": float(np.sqrt(np.mean(np.square(dths))))}
        # PCIe-inclusive rate (host buffers handed over each frame) -- reported beside, never as `value`
        t1 = time.perf_counter()
        gg = poses[0].astype(np.float32)
        for i in range(3):
            v.setInputTarget(tgt)
            v.setInputSource(scans[i])
            v.align(gg, want_output=False, want_fitness=True)
            gg = v.getFinalTransformation()
        v.synchronize()
        out["pcie_inclusive_scans_per_s"] = round(3 / (time.perf_counter() - t1), 3)

    print(json.dumps(out), flush=True)
    if world_size > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
