"""Repro behind a fix (round 6, found by tests/fuzz/fuzz_api.py): a sequence of re-framed / device / host targets, cleared clouds and user covariances that
left a context with a stale guard ('invalid configuration'); the sequence is cut down step by step to the calls that matter.  GPU.
    python scripts/lab_invalid_config.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rgc_slam_amd.synth as synth
from rgc_slam_amd import registration as reg
import bench
world, base = synth.make_world_and_map(36000, seed=5)
base = base.astype(np.float32)
rng = np.random.default_rng(1)
def cloud(n): return base[rng.choice(len(base), n, replace=False)]
def run(steps):
    v = reg.odometer_vgicp(0)
    bufs = []
    def dev(c):
        a = np.zeros((len(c), 4), np.float32); a[:, :3] = c; p = v.device_alloc(a.nbytes); v.upload(p, a); bufs.append(p); return p
    try:
        for st in steps:
            if st[0] == "reframed":
                c = cloud(st[1]); d = dev(c); s = v.device_alloc(len(c) * 16); bufs.append(s)
                Tw = synth.se3(synth.rot_zyx(rng.uniform(-3, 3), 0.01, 0.01), rng.uniform(-20, 20, 3) * np.array([1, 1, 0.02])); q, t = bench.world_to_body(Tw)
                v.setInputTargetReframed(d, len(c), 16, q, t, s)
            elif st[0] == "tgt_dev":
                c = cloud(st[1]); v.setInputTargetDevice(dev(c), len(c), 16)
            elif st[0] == "tgt_host": v.setInputTarget(cloud(st[1]))
            elif st[0] == "src_dev":
                c = cloud(st[1]); v.setInputSourceDevice(dev(c), len(c), 16)
            elif st[0] == "clear_tgt": v.clearTarget()
            elif st[0] == "clear_src": v.clearSource()
            elif st[0] == "getcov": v.getTargetCovariances()
            elif st[0] == "setcov": v.setTargetCovariances(v.getTargetCovariances())
        return "ok"
    except Exception as e:
        return "FAILED at %s: %s" % (st, str(e)[:120])
    finally:
        v.close()
full = [("reframed", 25408), ("reframed", 2397), ("reframed", 27640), ("setcov",), ("clear_src",), ("src_dev", 2462), ("clear_tgt",), ("clear_src",), ("tgt_dev", 16268), ("getcov",)]
print("full:", run(full))
print("no setcov:", run([s for s in full if s[0] != "setcov"]))
print("one reframed:", run([("reframed", 27640), ("clear_tgt",), ("tgt_dev", 16268), ("getcov",)]))
print("one reframed, no clear:", run([("reframed", 27640), ("tgt_dev", 16268), ("getcov",)]))
print("reframed small then dev:", run([("reframed", 2397), ("tgt_dev", 16268), ("getcov",)]))
print("host then dev:", run([("tgt_host", 27640), ("tgt_dev", 16268), ("getcov",)]))
