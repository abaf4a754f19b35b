"""Soak: contexts created and destroyed in a loop, clouds of changing sizes through every entry point; device memory must return."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import rgc_slam_amd.synth as synth
from rgc_slam_amd import registration, mapping, loop_closure, wire, local_map
world, tgt = synth.make_world_and_map(100000)
scan = synth.make_scan_n(world, np.eye(4), 20000)["xyz"]
free0 = torch.cuda.mem_get_info()[0]
rng = np.random.default_rng(0)
for it in range(40):
    v = registration.odometer_vgicp(0)
    nt, ns = int(rng.integers(2000, 100000)), int(rng.integers(500, 20000))
    v.setInputTarget(tgt[:nt]); v.setInputSource(scan[:ns])
    v.align(np.eye(4, dtype=np.float32), want_output=False, want_fitness=True)
    if it % 5 == 0:
        v.setResolution(0.5 + 0.1 * (it % 3))
        v.align(np.eye(4, dtype=np.float32), want_output=False)
    if it % 4 == 0:   # f2: the resident map through inserts, evictions, a re-base and commits, then destroyed with the context
        m = local_map.RollingLocalMap(v)
        m.reset([1.0, 2.0, 0.0])
        s4 = np.zeros((ns, 4), np.float32); s4[:, :3] = scan[:ns]
        for k in range(6):
            m.insert(s4, [0, 0, 0, 1.0], [1.0 + 0.2 * k, 2.0, 0.0])
            m.evict(3)
            m.commit(0.3)
        m.rebase([1.5, 2.0, 0.0]); m.commit(0.3)
        v.align(np.eye(4, dtype=np.float32), want_output=False)
    v.close()
    icp = loop_closure.IterativeClosestPoint(0)
    icp.setMaximumIterations(5); icp.setInputSource(scan[:ns]); icp.setInputTarget(tgt[:nt]); icp.align(); icp.close()
torch.cuda.synchronize()
free1 = torch.cuda.mem_get_info()[0]
print("device memory delta after 40 create/destroy cycles: %.1f MiB" % ((free0 - free1) / 2**20))
# ROCm 7.2 itself keeps ~1.5 MiB of device memory per hipStreamCreateWithFlags / hipStreamDestroy pair (measured with a bare HIP
# program: 146 MiB per 100 streams), i.e. ~3.7 MiB per context pair here; anything beyond that would be ours
assert abs(free0 - free1) < 40 * 2 * 2.2 * 2**20
print("soak ok")
