"""Lattices of exact distance ties (regular and stretched spacing, with repeated points): how many of the map's queries the bulk kNN defers to the
cooperative kernel, and whether every covariance still agrees with the oracle's (the tie rule: ascending original index).  GPU.
    python scripts/exp_ties.py"""
import sys; sys.path.insert(0,'/root/repo')
import numpy as np
from rgc_slam_amd import registration
from oracle import oracle as orc
for sp in ((0.25,0.25,0.25),(0.25,0.27,0.31)):
    g = np.stack(np.meshgrid(np.arange(14), np.arange(14), np.arange(5), indexing="ij"), axis=-1).reshape(-1, 3).astype(np.float32) * np.float32(sp)
    rng = np.random.default_rng(9)
    pts = np.concatenate([g, g[rng.choice(len(g), 150, replace=False)]])
    pts = pts[rng.permutation(len(pts))] + np.float32([3.0, -2.0, 0.5])
    v = registration.odometer_vgicp(0); v.setInputTarget(pts); v.setInputSource(pts[:400])
    ct = v.getTargetCovariances(); st = v.stats()
    o = orc.Registration(num_threads=0); o.set_target(pts); o.set_source(pts[:400]); o.prepare()
    ot = o.target_cov(len(pts))
    d = np.abs(ct - ot).reshape(len(pts), -1).max(axis=1)
    print(sp, 'deferred', st['deferred_target'], 'of', len(pts), 'agree<1e-9:', np.mean(d<1e-9), 'max', d.max(), 'n>1e-6:', (d>1e-6).sum())
