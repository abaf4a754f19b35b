"""A lazy target replaced while a scan is set, then read through the getters (found by tests/fuzz/fuzz_api.py)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rgc_slam_amd.synth as synth
from rgc_slam_amd import registration as reg
world, base = synth.make_world_and_map(20000, seed=5)
base = base.astype(np.float32)
A, B = base[:5245], base[3000:11368]
src = B[::3][:1500] + np.float32(0.02)
def full(cloud):
    w = reg.odometer_vgicp(0); w.setInputTarget(cloud); c = w.getTargetCovariances(); w.close(); return c
for case in ("A, S, B  (same device buffer)", "A, S, B  (host clouds)", "S, B", "A, S, B then a solve"):
    v = reg.odometer_vgicp(0); v.setLazyTarget(3)
    d = v.device_alloc(len(base) * 16); ds = v.device_alloc(len(base) * 16)
    def tgt(c):
        if "host" in case: v.setInputTarget(c)
        else:
            a = np.zeros((len(c), 4), np.float32); a[:, :3] = c; v.upload(d, a); v.setInputTargetDevice(d, len(c), 16)
    if case.startswith("A"): tgt(A)
    v.setInputSource(src)
    tgt(B)
    if case.endswith("solve"): v.align(np.eye(4, dtype=np.float32), want_output=False)
    c = v.getTargetCovariances(); f = full(B)
    bad = np.nonzero(np.any((c != f).reshape(len(c), -1), axis=1))[0]
    print(case, ": differ from the full build's:", len(bad), "of", len(c), v.stats()["lazy_misses"])
    v.close()
