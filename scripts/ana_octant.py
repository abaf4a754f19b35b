"""Offline analysis (numpy): candidates examined by an 'octant first, ball-clipped remainder' search vs the full 3x3x3 block."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rgc_slam_amd.synth as synth
nt = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
world, tgt = synth.make_world_and_map(nt)
res, k = 1.0, 20
P = tgt.astype(np.float32)
c = np.floor(P.astype(np.float64) / res - 0.5).astype(np.int64)
mn = c.min(0); c -= mn; dim = c.max(0) + 1
cell = (c[:, 2] * dim[1] + c[:, 1]) * dim[0] + c[:, 0]
order = np.argsort(cell, kind="stable")
Ps = P[order]; cs = cell[order]; cc = c[order]
cnt = np.bincount(cs, minlength=int(dim.prod())); start = np.concatenate([[0], np.cumsum(cnt)])
rng = np.random.default_rng(1)
sel = rng.choice(len(Ps), 4000, replace=False)
def pts(x, y, z):
    if x < 0 or y < 0 or z < 0 or x >= dim[0] or y >= dim[1] or z >= dim[2]: return np.zeros((0, 3), np.float32)
    ci = (z * dim[1] + y) * dim[0] + x
    return Ps[start[ci]:start[ci + 1]]
full, octc, rem, ok_oct, tot2 = [], [], [], [], []
for i in sel:
    q = Ps[i].astype(np.float64); cq = cc[i]
    lo = (cq + mn + 0.5) * res
    side = np.where(q - lo < 0.5 * res, -1, 1)
    blk = [(dx, dy, dz) for dz in (-1, 0, 1) for dy in (-1, 0, 1) for dx in (-1, 0, 1)]
    allp = {o: pts(cq[0] + o[0], cq[1] + o[1], cq[2] + o[2]) for o in blk}
    full.append(sum(len(v) for v in allp.values()))
    octs = [o for o in blk if all(o[a] in (0, side[a]) for a in range(3))]
    op = np.concatenate([allp[o] for o in octs])
    octc.append(len(op))
    if len(op) >= k:
        d2 = np.sort(((op - q) ** 2).sum(1))[k - 1]; rad = np.sqrt(d2) * (1 + 1e-5); ok_oct.append(1)
    else:
        rad = 1e30; ok_oct.append(0)
    r = 0
    for o in blk:
        if o in octs: continue
        # cell box distance to q
        clo = (cq + np.array(o) + mn + 0.5) * res; chi = clo + res
        d = np.maximum(np.maximum(clo - q, q - chi), 0)
        # row-level clipping as implemented: (y,z) wall distance + common x extent
        dyz = d[1] ** 2 + d[2] ** 2
        if dyz > rad * rad: continue
        if d[0] > rad: continue
        r += len(allp[o])
    rem.append(r); tot2.append(len(op) + r)
full, octc, rem, ok_oct, tot2 = map(np.array, (full, octc, rem, ok_oct, tot2))
print("full block mean", full.mean(), " octant mean", octc.mean(), " octant>=k frac", ok_oct.mean())
print("clipped remainder mean", rem.mean(), " total examined mean", tot2.mean(), " (p90", np.percentile(tot2, 90), ")")
