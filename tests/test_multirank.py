"""N>1 control flow of bench.py without GPUs: two gloo ranks, each with its own independent sequence seed, barrier +
MAX-over-ranks of the elapsed time, aggregate = units of all ranks / that time (no data-path collective)."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import os, sys, time, json
    sys.path.insert(0, %r)
    import numpy as np, torch, torch.distributed as dist
    import rgc_slam_amd.synth as synth
    from oracle import oracle
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    seed = synth.SEED + rank                       # one independent sequence per rank, as bench.py does
    w, tgt = synth.make_world_and_map(6000, seed=seed)
    poses = synth.make_trajectory(3, seed=seed)
    scans = [synth.make_scan_n(w, poses[i + 1], 1500, seed=seed + 100 + i)["xyz"] for i in range(2)]
    reg = oracle.Registration(num_threads=1)      # CPU stand-in for the per-rank context (no GPU in this test)
    dist.barrier()
    t0 = time.perf_counter()
    g = np.eye(4, dtype=np.float32); finals = []
    for s in scans:
        reg.set_target(tgt); reg.set_source(s); g = reg.align(g); finals.append(g.copy())
    el = time.perf_counter() - t0
    dist.barrier()
    t = torch.tensor([el], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    # every rank's result must equal a single-process run of the same sequence bit for bit (sequences share nothing)
    chk = torch.tensor(np.stack(finals).astype(np.float64).ravel())
    gathered = [torch.zeros_like(chk) for _ in range(world)]
    dist.all_gather(gathered, chk)
    if rank == 0:
        print(json.dumps({"value": 2 * world / float(t.item()), "max_elapsed": float(t.item()),
                          "finals": [x.tolist() for x in gathered]}))
    dist.destroy_process_group()
""") % ROOT


def _single(seed_rank):
    import numpy as np
    sys.path.insert(0, ROOT)
    import rgc_slam_amd.synth as synth
    from oracle import oracle
    seed = synth.SEED + seed_rank
    w, tgt = synth.make_world_and_map(6000, seed=seed)
    poses = synth.make_trajectory(3, seed=seed)
    scans = [synth.make_scan_n(w, poses[i + 1], 1500, seed=seed + 100 + i)["xyz"] for i in range(2)]
    reg = oracle.Registration(num_threads=1)
    g = np.eye(4, dtype=np.float32)
    out = []
    for s in scans:
        reg.set_target(tgt); reg.set_source(s); g = reg.align(g); out.append(g.copy())
    return np.stack(out).astype(np.float64).ravel()


def test_two_rank_gloo(tmp_path):
    import json
    import numpy as np
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
                          "127.0.0.1", "--master-port", "29617", str(script)], capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["value"] > 0 and d["max_elapsed"] > 0
    for r in range(2):
        assert np.array_equal(np.asarray(d["finals"][r]), _single(r)), f"rank {r} differs from its single-process run"
    assert not np.array_equal(np.asarray(d["finals"][0]), np.asarray(d["finals"][1]))  # really different sequences
