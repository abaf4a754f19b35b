"""Design study (CPU): what an INCREMENTAL commit of the keyframe store could skip.  A store of N_KF sweeps (the synthetic sensor moving
along the bench's trajectory), leaf-filtered at 0.3 m; then one keyframe arrives and the oldest leaves, as the reference's deque does every
keyframe (RGC_odometer.cpp:1237-1247).  Counted: the leaves whose membership (hence centroid) changes, the 1 m cells of the registration grid
that hold such a leaf, the cells within one cell of those (a query's exact 20-NN is decided inside its 3x3x3 block, so every query of such a
cell must be searched again), and the share of the filtered map's points that live in them -- the share of the kNN launch and of the voxel
map an incremental commit would still have to redo.   python scripts/sim_incremental_commit.py [keyframes] [points per sweep] [step m]"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rgc_slam_amd.synth as synth

N_KF = int(sys.argv[1]) if len(sys.argv) > 1 else 32
N_S = int(sys.argv[2]) if len(sys.argv) > 2 else 31000
STEP = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
LEAF, CELL = 0.3, 1.0

world = synth.make_world(half_extent=45.0, seed=synth.SEED)
poses = synth.make_trajectory(N_KF + 3, seed=synth.SEED)
# stretch the trajectory so that consecutive keyframes are STEP metres apart (a keyframe is taken every ~1 m / 10 degrees in the reference)
base = poses[0][:3, 3].copy()
for k, T in enumerate(poses):
    d = T[:3, 3] - base
    n = np.linalg.norm(d)
    T[:3, 3] = base + (d / n * STEP * k if n > 0 else 0)
sweeps = []
for k in range(N_KF + 1):
    sc = synth.make_scan_n(world, poses[k], N_S, seed=synth.SEED + 700 + k)["xyz"].astype(np.float64)
    sweeps.append((poses[k][:3, :3] @ sc.T).T + poses[k][:3, 3])          # world frame


def leaf_keys(p, size):
    return np.floor(p / size).astype(np.int64)


def pack(k):
    k = k + (1 << 20)
    return (k[:, 0] << 42) | (k[:, 1] << 21) | k[:, 2]


def filtered(store):
    """the leaf filter's output: one centroid per occupied leaf; returns {leaf key: (count, sum)} as sorted arrays"""
    keys = pack(leaf_keys(store, LEAF))
    order = np.argsort(keys, kind="stable")
    ks = keys[order]
    head = np.r_[True, ks[1:] != ks[:-1]]
    idx = np.flatnonzero(head)
    cnt = np.diff(np.r_[idx, len(ks)])
    sums = np.add.reduceat(store[order], idx, axis=0)
    return ks[idx], cnt, sums / cnt[:, None]


before = np.concatenate(sweeps[:N_KF])
after = np.concatenate(sweeps[1:N_KF + 1])
kb, cb, pb = filtered(before)
ka, ca, pa = filtered(after)
# leaves whose output changes: present in only one of the two, or present in both with a different member set (the sweeps differ, so a
# leaf touched by the leaving or by the arriving sweep changes its centroid)
touched = np.union1d(np.unique(pack(leaf_keys(sweeps[0], LEAF))), np.unique(pack(leaf_keys(sweeps[N_KF], LEAF))))
changed_after = np.isin(ka, touched)
cells_after = pack(leaf_keys(pa, CELL))
occ_cells = np.unique(cells_after)
changed_cells = np.unique(cells_after[changed_after])
# cells of leaves that vanished altogether
vanished = ~np.isin(kb, ka)
changed_cells = np.union1d(changed_cells, np.unique(pack(leaf_keys(pb[vanished], CELL))))
# dilation by one cell
def unpack(c):
    return np.stack([(c >> 42) - (1 << 20), ((c >> 21) & ((1 << 21) - 1)) - (1 << 20), (c & ((1 << 21) - 1)) - (1 << 20)], axis=1)
cc = unpack(changed_cells)
offs = np.array([[x, y, z] for x in (-1, 0, 1) for y in (-1, 0, 1) for z in (-1, 0, 1)])
dil = np.unique(pack((cc[:, None, :] + offs[None]).reshape(-1, 3)))
dirty_cells = np.intersect1d(dil, occ_cells)
pts_dirty = np.isin(cells_after, dirty_cells).sum()
out = {"keyframes": N_KF, "points_per_sweep": N_S, "metres_between_keyframes": STEP, "store_points": int(len(after)),
       "filtered_points": int(len(ka)), "leaves_whose_output_changes": int(changed_after.sum() + vanished.sum()),
       "share_of_filtered_points_that_change": round(float(changed_after.mean()), 4),
       "occupied_cells": int(len(occ_cells)), "cells_holding_a_changed_leaf": int(np.isin(occ_cells, changed_cells).sum()),
       "cells_within_one_cell_of_those": int(len(dirty_cells)),
       "share_of_points_whose_knn_must_be_searched_again": round(float(pts_dirty / len(ka)), 4)}
print(json.dumps(out))
