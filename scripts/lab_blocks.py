"""Developer aid (library built with RGC_EXTRA_FLAGS=-DRGC_LAB): when every workgroup of the map's bulk kNN launch started and ended
(100 MHz wall clock) and on which XCD, for the unseeded first launch and a seeded one on a re-framed 1 M-point map."""
import sys, os, json, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rgc_slam_amd.synth as synth
from rgc_slam_amd import registration
import bench
nt = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
T = int(os.environ.get("RGC_LAB_T", "256"))
world, tgt = synth.make_world_and_map(nt, seed=synth.SEED)
poses = synth.make_trajectory(6, seed=synth.SEED)
v = registration.odometer_vgicp(0)
lib = v._L
lib.rgc_lab_blocks.argtypes = [C.c_void_p, C.c_void_p]
a = np.zeros((nt, 4), np.float32); a[:, :3] = tgt
d_map, d_body = v.device_alloc(a.nbytes), v.device_alloc(a.nbytes)
v.upload(d_map, a)
ts = np.zeros(4 * 16384, np.int64)
for f in range(4):
    q, t = bench.world_to_body(np.asarray(poses[f], np.float64))
    v.setInputTargetReframed(d_map, nt, 16, q, t, d_body); v.synchronize()
    lib.rgc_lab_blocks(v._h, ts.ctypes.data)
    nb = min((nt + T - 1) // T, 16384)
    r = ts.reshape(-1, 4)[:nb]
    why = (r[:, 3] >> 32).astype(np.int64)
    r[:, 3] &= 0xffffffff
    keep = r[:, 3] < nt
    r, why = r[keep], why[keep]
    t0 = r[:, 0].min()
    st, en = (r[:, 0] - t0) * 0.01, (r[:, 1] - t0) * 0.01
    dur = en - st
    xcc = r[:, 2] & 15
    per_x = {int(x): [int((xcc == x).sum()), round(float(en[xcc == x].max()), 1), round(float(dur[xcc == x].sum()), 0)] for x in np.unique(xcc)}
    names = ["", "no seed", "crowded row", "fewer than k", "more than k+1", "k keys undecided", "k+2 may contend", "three contenders", "exact tie"]
    by_reason = {names[b_]: [int(((why >> b_) & 1).sum()), round(float(np.median(dur[((why >> b_) & 1) == 1])), 1) if ((why >> b_) & 1).any() else 0] for b_ in range(1, 9)}
    order = np.argsort(-dur)[:12]
    longest = [[round(float(st[o]), 1), round(float(dur[o]), 1), int(why[o])] for o in order]
    edges = np.linspace(0.0, float(en.max()), 21)
    infl = [int(((st <= e_) & (en > e_)).sum()) for e_ in edges[:-1]]
    print(json.dumps({"frame": f, "blocks": int(len(r)), "span_us": round(float(en.max()), 1), "last_start_us": round(float(st.max()), 1),
                      "block_us": {"median": round(float(np.median(dur)), 1), "p10": round(float(np.percentile(dur, 10)), 1), "p90": round(float(np.percentile(dur, 90)), 1), "max": round(float(dur.max()), 1)},
                      "sum_block_us": round(float(dur.sum()), 0), "per_xcc [blocks, last end us, sum us]": per_x,
                      "blocks_in_flight_at_twentieths_of_the_span": infl,
                      "blocks with a declined lane, by reason [blocks, median us]": by_reason, "undeclined blocks median us": round(float(np.median(dur[why == 0])), 1),
                      "longest blocks [start us, us, reason mask]": longest}), flush=True)
v.close()
