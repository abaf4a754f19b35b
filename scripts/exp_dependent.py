"""Developer aid: where the host time of a dependent step goes (bench.DependentSequence), one context and two.
    python scripts/exp_dependent.py [frames]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
import rgc_slam_amd.synth as synth
from rgc_slam_amd import registration
K = int(sys.argv[1]) if len(sys.argv) > 1 else 12
world, tgt = synth.make_world_and_map(1000000, seed=synth.SEED)
poses = synth.make_trajectory(K + 2, seed=synth.SEED)
scans = [synth.make_scan_n(world, poses[i + 1], 30000, seed=synth.SEED + 100 + i)["xyz"] for i in range(K + 1)]
pv = registration.PipelinedVGICP(0, depth=2)
v = pv.v[0]
def to_dev(xyz):
    a = np.zeros((xyz.shape[0], 4), np.float32); a[:, :3] = xyz
    p = v.device_alloc(a.nbytes); v.upload(p, a); return p
d_map, d_scans = to_dev(tgt), [to_dev(s) for s in scans]
seq = bench.DependentSequence(pv.v, d_map, len(tgt), d_scans, [len(s) for s in scans])
I4 = np.eye(4, dtype=np.float32)
for w in pv.v:
    seq.v = [w]; seq.run(0, 1, poses[0], I4, False)
seq.v = pv.v
T = time.perf_counter
for overlap in (False, True) * int(os.environ.get('RGC_EXP_PASSES', '2')):
    vv = pv.v if overlap else pv.v[:1]
    Tw, g = np.asarray(poses[0], np.float64), I4
    rows = []
    if overlap: seq.frame_source(1, vv[0])
    t_all = T()
    for j in range(K):
        cur = vv[j % len(vv)]
        t0 = T(); q, t = bench.world_to_body(Tw)
        t1 = T()
        t2 = T(); cur.setInputTargetReframed(d_map, len(tgt), 16, q, t, seq.d_body[id(cur)])
        t3 = T()
        if not overlap: seq.frame_source(1 + j, cur)
        t4 = T(); cur.align_begin(g, True)
        t5 = T()
        if overlap and j + 1 < K:
            if os.environ.get("RGC_EXP_HOLD", "1") == "1": vv[(j + 1) % 2].holdSourceUntilTargetOf(cur)
            seq.frame_source(2 + j, vv[(j + 1) % 2])
        t6 = T(); Tm = cur.align_end()
        t7 = T(); Tw = Tw @ Tm.astype(np.float64); g = Tm
        rows.append([t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t6 - t5, t7 - t6, T() - t7])
    pv.synchronize()
    tot = (T() - t_all) / K * 1e3
    r = np.median(np.array(rows), axis=0) * 1e3
    print(f"overlap={overlap}: {tot:.3f} ms/frame; median ms: pose math {r[0]:.3f} transform {r[1]:.3f} set_target {r[2]:.3f} set_source(seq) {r[3]:.3f} "
          f"align_begin {r[4]:.3f} set_source(next) {r[5]:.3f} align_end {r[6]:.3f} compose {r[7]:.3f}", flush=True)
