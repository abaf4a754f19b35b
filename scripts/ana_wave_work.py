"""Offline analysis (numpy, no GPU): candidate counts per query and per 64-query wave for the 3x3x3 block search."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rgc_slam_amd.synth as synth
nt = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
world, tgt = synth.make_world_and_map(nt)
res = 1.0
c = np.floor(tgt.astype(np.float32) / np.float32(res) - np.float32(0.5)).astype(np.int64)
mn = c.min(0); c -= mn; dim = c.max(0) + 1
cell = (c[:, 2] * dim[1] + c[:, 1]) * dim[0] + c[:, 0]
order = np.argsort(cell, kind="stable")
cs = cell[order]; cc = c[order]
ncell = int(dim.prod())
cnt = np.bincount(cs, minlength=ncell)
start = np.concatenate([[0], np.cumsum(cnt)])
n = len(cs)
# per query: 9 row lengths
lens = np.zeros((n, 9), np.int64)
x0 = np.maximum(cc[:, 0] - 1, 0); x1 = np.minimum(cc[:, 0] + 1, dim[0] - 1)
for r in range(9):
    y = cc[:, 1] + r % 3 - 1; z = cc[:, 2] + r // 3 - 1
    ok = (y >= 0) & (y < dim[1]) & (z >= 0) & (z < dim[2])
    yy = np.where(ok, y, cc[:, 1]); zz = np.where(ok, z, cc[:, 2])
    a = start[(zz * dim[1] + yy) * dim[0] + x0]; b = start[(zz * dim[1] + yy) * dim[0] + x1 + 1]
    lens[:, r] = np.where(ok, b - a, 0)
tot = lens.sum(1)
own = cnt[cs]
print("queries", n, "occupied cells", (cnt > 0).sum(), "mean own", own.mean(), "mean tot", tot.mean(), "p50/p90/p99/max", np.percentile(tot, [50, 90, 99, 100]))
W = 64
nw = n // W
L = lens[: nw * W].reshape(nw, W, 9)
T = tot[: nw * W].reshape(nw, W)
rows4 = (np.ceil(L / 4) * 4)
print("per wave: mean of lane-mean tot", T.mean(), " mean of lane-max tot", T.max(1).mean(), " mean sum_r max_lane ceil4(len_r)", rows4.max(1).sum(1).mean())
print("heavy>640 frac", (tot > 640).mean())
