"""Where the time goes right after a shared target is rebuilt: per-call host times of the first frames on two contexts."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rgc_slam_amd.synth as synth
from rgc_slam_amd import registration, local_map
world, tgt = synth.make_world_and_map(1000000)
poses = synth.make_trajectory(9)
scans = [synth.make_scan_n(world, poses[i + 1], 30000, seed=synth.SEED + 100 + i)["xyz"] for i in range(8)]
a, b = registration.odometer_vgicp(0), registration.odometer_vgicp(0)
def to_dev(xyz):
    x = np.zeros((xyz.shape[0], 4), np.float32); x[:, :3] = xyz
    p = a.device_alloc(x.nbytes); a.upload(p, x); return p
d_t = to_dev(tgt); d_s = [to_dev(s) for s in scans]
g = poses[0].astype(np.float32)
def T(f, *args):
    t0 = time.perf_counter(); r = f(*args); return r, 1e3 * (time.perf_counter() - t0)
m = local_map.RollingLocalMap(a)
m.reset(None)
tgt4 = np.zeros((len(tgt), 4), np.float32); tgt4[:, :3] = tgt
chunks = np.array_split(tgt4, 32)
for ch in chunks:
    m.insert(ch, [0, 0, 0, 1.0], [0, 0, 0])
def recommit(k):
    m.insert(chunks[k % 32], [0, 0, 0, 1.0], [0, 0, 0]); m.evict(32); m.commit(0.3)
for rep in range(4):
    _, t_set = T(recommit, rep)
    _, t_sh = T(b.shareTargetFrom, a)
    line = [f"set_target {t_set:.3f}", f"share {t_sh:.3f}"]
    for i in range(4):
        w = (a, b)[i & 1]
        _, t1 = T(w.setInputSourceDevice, d_s[i], 30000, 16)
        _, t2 = T(w.align_begin, g, True)
        g2, t3 = T(w.align_end)
        g = g2
        line.append(f"[{'ab'[i & 1]}: src {t1:.3f} begin {t2:.3f} end {t3:.3f}]")
    print(" ".join(line))
    g = poses[0].astype(np.float32)

print("pipelined order after a commit:")
for rep in range(3):
    recommit(rep + 8); b.shareTargetFrom(a)
    g = poses[0].astype(np.float32)
    line = []
    _, t = T(a.setInputSourceDevice, d_s[0], 30000, 16); line.append(f"a.src {t:.3f}")
    for i in range(3):
        cur, nxt = (a, b)[i & 1], (a, b)[(i + 1) & 1]
        _, t = T(cur.align_begin, g, True); line.append(f"{'ab'[i & 1]}.begin {t:.3f}")
        if i + 1 < 3:
            _, t = T(nxt.setInputSourceDevice, d_s[i + 1], 30000, 16); line.append(f"{'ab'[(i + 1) & 1]}.src {t:.3f}")
        g, t = T(cur.align_end); line.append(f"{'ab'[i & 1]}.end {t:.3f}")
    print(" ".join(line))
