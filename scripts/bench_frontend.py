"""Front-end (A1-A8) timing on the MI355X against the CPU oracle, one synthetic VLP-16 sweep (~23 k points).

gpu_ms: rgc_frontend_device on a sweep that is already in HBM with the ring-major cloud left there -- what the chained frame body of the
odometry node calls (rgc::OdometryNode, device_chain): the features, the flags and the ground fit come down, nothing else crosses PCIe.
host_in_host_out_ms: rgc_frontend, sweep and every output in page-locked host memory (rgc_host_alloc): input up, cloud + features +
ground list down.  python_mirror_ms: rgc_slam_amd.frontend.ScanRegistration.laserCloudHandler as the parity tests call it (numpy buffers
allocated per call: what round 1-3 printed as gpu_ms)."""
import sys, os, time, json, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401
import rgc_slam_amd.synth as synth
from rgc_slam_amd import frontend, _lib
from oracle import oracle
w = synth.make_world(seed=synth.SEED)
sc = synth.make_scan(w, np.eye(4), n_az=1800, seed=synth.SEED + 3)
xyzi = np.concatenate([sc["xyz"], sc["intensity"][:, None]], axis=1).astype(np.float32)
n = len(xyzi)
fe = frontend.ScanRegistration(device=0)
L, h = fe._L, fe._h
reps = 100
def timed(fn):
    for _ in range(5): fn()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    return (time.perf_counter() - t0) / reps
t_py = timed(lambda: fe.laserCloudHandler(xyzi, diagnostics=False))
out_py = fe.laserCloudHandler(xyzi, diagnostics=False)
# the library calls themselves, buffers allocated once (page-locked: rgc_host_alloc)
ns = fe.params.n_scans
fcap, gcap = ns * 6 * 41, 10 * n
def pinned(nbytes):
    p = C.c_void_p()
    assert L.rgc_host_alloc(nbytes, C.byref(p)) == 0
    return p
f32 = C.POINTER(C.c_float)
h_in = pinned(16 * n); C.memmove(h_in, xyzi.ctypes.data, 16 * n)
h_cloud, h_sharp, h_flat, h_inten, h_ground = pinned(16 * n), pinned(20 * fcap), pinned(20 * fcap), pinned(20 * fcap), pinned(16 * gcap)
def make_out(cloud, ground):
    o = _lib.FeOut()
    o.cloud = C.cast(h_cloud, f32) if cloud else None
    o.cloud_cap = n
    o.sharp, o.flat, o.inten, o.feat_cap = C.cast(h_sharp, f32), C.cast(h_flat, f32), C.cast(h_inten, f32), fcap
    o.ground_pts = C.cast(h_ground, f32) if ground else None
    o.ground_cap = gcap if ground else 0
    return o
o_host, o_dev = make_out(True, True), make_out(False, False)
def run_host():
    assert L.rgc_frontend(h, h_in, n, 16, C.byref(fe.params), C.byref(o_host)) == 0
t_host = timed(run_host)
d_in = C.c_void_p()
assert L.rgc_device_alloc(h, 16 * n, C.byref(d_in)) == 0
assert L.rgc_upload(h, d_in, h_in, 16 * n) == 0 and L.rgc_synchronize(h) == 0
def run_dev():
    assert L.rgc_frontend_device(h, d_in, n, 16, C.byref(fe.params), C.byref(o_dev)) == 0
t_dev = timed(run_dev)
assert o_dev.n_sharp == len(out_py["sharp"]) and o_dev.n_flat == len(out_py["flat"]) and o_host.n_sharp == o_dev.n_sharp and o_dev.n_cloud == out_py["n_cloud"]
t0 = time.perf_counter()
for _ in range(5):
    ref = oracle.frontend(xyzi)
t_cpu = (time.perf_counter() - t0) / 5
print(json.dumps({"workload": f"front-end A1-A8, one VLP-16 sweep of {n} points", "gpu_ms": round(1e3 * t_dev, 3),
                  "what_gpu_ms_is": "rgc_frontend_device: sweep in HBM, ring-major cloud left in HBM (the node's chained frame body); features, flags and ground fit come down",
                  "host_in_host_out_ms": round(1e3 * t_host, 3), "python_mirror_ms": round(1e3 * t_py, 3), "cpu_oracle_ms_1_thread": round(1e3 * t_cpu, 3),
                  "n_sharp": int(o_dev.n_sharp), "n_flat": int(o_dev.n_flat)}))
