"""Generates tests/golden/*.npz from the independent numpy/scipy restatement (oracle/py_oracle.py).

The reference holds no golden vectors for this path (parity unpinned) and cannot
be compiled or imported, so these fixtures pin the C oracle and the HIP path against a second,
literal restatement.  Run from the repo root:  python tests/golden/gen_golden.py
Seeds are recorded inside every file.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import rgc_slam_amd.synth as synth  # noqa: E402
from oracle import py_oracle as po  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
SEED = synth.SEED


def tri6(c):
    c = np.asarray(c)
    return np.stack([c[..., 0, 0], c[..., 0, 1], c[..., 0, 2], c[..., 1, 1], c[..., 1, 2], c[..., 2, 2]], axis=-1)


def main():
    world = synth.make_world(half_extent=14.0, seed=SEED)
    tgt = synth.make_map(world, 8000, seed=SEED)
    T_true = synth.se3(synth.rot_zyx(0.015, 0.002, -0.003), [0.12, -0.03, 0.004])
    src = synth.make_scan_n(world, T_true, 2000, seed=SEED)["xyz"]
    guess = np.eye(4, dtype=np.float32)

    reg = po.VGICP()
    reg.set_target(tgt)
    reg.set_source(src)
    cost, H, b = reg.linearize(guess.astype(np.float64))
    n_corr = len(reg.corr)
    reg7 = po.VGICP(method="DIRECT7")
    reg7.tgt, reg7.tgt_covs, reg7.vox = reg.tgt, reg.tgt_covs, reg.vox
    reg7.src, reg7.src_covs = reg.src, reg.src_covs
    cost7, H7, b7 = reg7.linearize(guess.astype(np.float64))
    n_corr7 = len(reg7.corr)
    # error at a perturbed pose with correspondences frozen at `guess`
    Tp = synth.se3(synth.rot_zyx(0.004, -0.001, 0.002), [0.03, 0.01, -0.002])
    err_p = reg.compute_error(Tp)

    final = reg.align(guess)
    fit = reg.fitness()
    tr = reg.trace
    keys = sorted(reg.vox.keys())
    np.savez_compressed(
        os.path.join(OUT, "fx_registration.npz"),
        seed=SEED, src=src, tgt=tgt, guess=guess, T_true=T_true,
        src_knn=np.sort(reg.src_knn, axis=1).astype(np.int32),
        tgt_knn_sub=np.sort(reg.tgt_knn[::8], axis=1).astype(np.int32),
        src_cov6=tri6(reg.src_covs[:, :3, :3]), tgt_cov6_sub=tri6(reg.tgt_covs[::4, :3, :3]),
        vox_coords=np.asarray(keys, dtype=np.int32), vox_num=np.asarray([reg.vox[k]["n"] for k in keys], dtype=np.int32),
        vox_mean=np.asarray([reg.vox[k]["mean"][:3] for k in keys]), vox_cov6=tri6(np.asarray([reg.vox[k]["cov"][:3, :3] for k in keys])),
        lin_cost=cost, lin_H=H, lin_b=b, lin_ncorr=n_corr,
        lin7_cost=cost7, lin7_H=H7, lin7_b=b7, lin7_ncorr=n_corr7,
        err_T=Tp, err_cost=err_p,
        lm_y0=np.asarray([t["y0"] for t in tr]), lm_yi=np.asarray([t["yi"] for t in tr]),
        lm_rho=np.asarray([t["rho"] for t in tr]), lm_lambda=np.asarray([t["lambda_after"] for t in tr]),
        lm_inner=np.asarray([t["inner"] for t in tr]), lm_accepted=np.asarray([t["accepted"] for t in tr]),
        lm_ncorr=np.asarray([t["n_corr"] for t in tr]), lm_x=np.asarray([t["x"] for t in tr]),
        final_T=final, converged=reg.converged, fitness=fit)
    print("fx_registration: n_corr", n_corr, "iters", len(tr), "converged", reg.converged, "fitness", fit)
    print(" final t", final[:3, 3], "true t", T_true[:3, 3])

    # VoxelGrid fixture (B3)
    sc = synth.make_scan(world, np.eye(4), n_az=200, seed=SEED + 11)
    xyzi = np.concatenate([sc["xyz"], (sc["ring"] + 0.1 * sc["rel_time"])[:, None].astype(np.float32)], axis=1)
    vg02 = po.voxelgrid_filter(xyzi, 0.2)
    vg03 = po.voxelgrid_filter(xyzi, 0.3)
    np.savez_compressed(os.path.join(OUT, "fx_voxelgrid.npz"), seed=SEED + 11, xyzi=xyzi, out_02=vg02, out_03=vg03)
    print("fx_voxelgrid:", xyzi.shape, "->", vg02.shape, vg03.shape)

    # small known-answer table: so3_exp, voxel_coord
    rng = np.random.default_rng(SEED)
    ws = np.concatenate([rng.normal(0, 0.2, (8, 3)), rng.normal(0, 1e-6, (4, 3)), np.zeros((1, 3))])
    Rs = np.asarray([po.so3_exp(w) for w in ws])
    xs = np.array([[0.5, -0.5, 1.49999], [-1.5, 0.49999, -0.50001], [1.5, 2.5, -2.5], [0.0, -0.0, 0.25], [-100.3, 99.7, 0.5]])
    cs = np.asarray([po.voxel_coord(x, 1.0) for x in xs])
    cs05 = np.asarray([po.voxel_coord(x, 0.5) for x in xs])
    np.savez_compressed(os.path.join(OUT, "fx_small.npz"), so3_w=ws, so3_R=Rs, vc_x=xs, vc_res1=cs, vc_res05=cs05)


if __name__ == "__main__":
    main()
