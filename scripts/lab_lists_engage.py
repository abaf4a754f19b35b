"""When do the neighbour lists of an unchanged map engage?  A map handed over five times by rgc_set_target_reframed at random poses: queries searched,
deferred, and cells per frame (VAR=align / vox: with a solve or a voxel read-back between the frames).  GPU.
    VAR=align python scripts/lab_lists_engage.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rgc_slam_amd.synth as synth
from rgc_slam_amd import registration
import bench
for n, kind in ((20000, "synth"), (200000, "synth"), (50000, "uniform")):
    rng = np.random.default_rng(3)
    pts = synth.make_world_and_map(n, seed=7)[1].astype(np.float32) if kind == "synth" else rng.uniform(-8, 8, (n, 3)).astype(np.float32)
    n = len(pts); a = np.zeros((n, 4), np.float32); a[:, :3] = pts
    v = registration.odometer_vgicp(0); v.setResolution(1.0); v.setCorrespondenceRandomness(20); v.setNeighbourReuse(2)
    variant = os.environ.get('VAR', '')
    dm, db = v.device_alloc(a.nbytes), v.device_alloc(a.nbytes); v.upload(dm, a)
    out = []
    for f in range(5):
        Tw = synth.se3(synth.rot_zyx(*(rng.uniform(-0.1, 0.1, 3))), rng.uniform(-3, 3, 3))
        q, t = bench.world_to_body(Tw)
        v.setInputTargetReframed(dm, n, 16, q, t, db)
        v.getTargetCovariances()
        st = v.stats(); out.append((st["searched_target"], st["deferred_target"], st["target_cells"]))
        if 'align' in variant:
            body = v.download(db, (n, 4)); src = body[rng.choice(n, 3000, replace=False), :3] + np.float32(0.01)
            v.setInputSource(src); v.align(np.eye(4, dtype=np.float32), want_output=False, want_fitness=True)
        if 'vox' in variant:
            v.getVoxels()
    print(kind, n, out)
