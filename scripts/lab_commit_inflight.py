"""Repro behind a fix (round 6): rgc_map_commit with a solve in flight on the map's context must be refused BEFORE it writes anything -- commits with
changing leaf sizes between rgc_align_begin and rgc_align_end, under PLANE and FROBENIUS; regression test: tests/test_gpu_routes.py::test_a_refused_commit_writes_nothing.  GPU.
    python scripts/lab_commit_inflight.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rgc_slam_amd.synth as synth
from rgc_slam_amd import registration as reg, local_map, _lib
world, base = synth.make_world_and_map(20000, seed=5)
base = base.astype(np.float32)
for method in (3, 4):
    v = reg.odometer_vgicp(0); v.setRegularizationMethod(method)
    lm = local_map.RollingLocalMap(v); lm.reset(None)
    for k in range(3):
        a = np.zeros((3000, 4), np.float32); a[:, :3] = base[k * 3000:(k + 1) * 3000]
        lm.insert(a, np.array([0, 0, 0, 1.0]), np.zeros(3))
    n0 = lm.commit(0.3); t0 = lm.target().copy()
    lm.evict(2)
    v.setInputSource(base[:2000] + np.float32(0.02))
    v.align_begin(np.eye(4, dtype=np.float32))
    out = []
    for leaf in (0.3, 0.5, 0.3):
        try:
            n = lm.commit(leaf); out.append(("OK", n))
        except _lib.RgcError as e:
            out.append(("refused", str(e)[:60]))
    T = v.align_end()
    print("method", method, out, "info", lm.info()["n_target"])
    v.close()
