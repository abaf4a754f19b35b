"""Does a context give back what it took?  The process's resident set over hundreds of context life cycles, by what the context was used for
(scripts/lab_leaks.py).  Found: nothing of the library's -- a bare create / destroy costs 0.1 KiB, every use 0.0-0.7 KiB per life cycle; the
front-end's first ~200 life cycles grow the resident set by 0.9 MB each (the runtime's pool behind hipHostMalloc / hipHostFree filling up: the
same loop stays flat for the next thousand)."""
import sys, os, ctypes as C, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rgc_slam_amd.synth as synth
from rgc_slam_amd import registration as reg, odometry, frontend, local_map
def rss(): return int(open("/proc/self/statm").read().split()[1]) * 4096
world, base = synth.make_world_and_map(30000, seed=5)
base = base.astype(np.float32); src = base[::3][:6000] + np.float32(0.02)
a = np.zeros((len(base), 4), np.float32); a[:, :3] = base
sc = synth.make_scan(world, np.eye(4), n_az=600, seed=3)
raw = np.concatenate([sc["xyz"], sc["intensity"][:, None]], axis=1).astype(np.float32)
I = np.eye(4, dtype=np.float32)
def plain():
    v = reg.odometer_vgicp(0); v.setInputTarget(base); v.setInputSource(src); v.align(I, want_output=False); v.close()
def plain_same_ctx(v=[None]):
    if v[0] is None: v[0] = reg.odometer_vgicp(0)
    v[0].setInputTarget(base); v[0].setInputSource(src); v[0].align(I, want_output=False)
def reframed():
    v = reg.odometer_vgicp(0); d, s = v.device_alloc(a.nbytes), v.device_alloc(a.nbytes); v.upload(d, a)
    v.setInputTargetReframed(d, len(a), 16, np.array([0, 0, 0.1, 0.995]), np.zeros(3), s); v.device_free(d); v.device_free(s); v.close()
def general():
    v = reg.odometer_vgicp(0); v.setRegularizationMethod(1); v.setInputTarget(base[:5000]); v.setInputSource(src[:1000]); v.align(I, want_output=False); v.close()
def pre():
    p = odometry.Preprocessor(0); p.voxelGridFilter(a, 0.3); p.close()
def fe():
    f = frontend.ScanRegistration(16); f.laserCloudHandler(raw); f.close()
def lmap():
    v = reg.odometer_vgicp(0); lm = local_map.RollingLocalMap(v); lm.reset(None); lm.insert(a[:4000], np.array([0, 0, 0, 1.0]), np.zeros(3)); lm.commit(0.3); v.close()
for name, fn in (("set + align, a context per round", plain), ("set + align on ONE context", plain_same_ctx), ("re-framed target", reframed), ("general route", general), ("leaf filter", pre), ("front-end", fe), ("resident map", lmap)):
    for _ in range(10): fn()
    gc.collect(); r0 = rss()
    for _ in range(200): fn()
    gc.collect(); r1 = rss()
    print("%-36s %7.1f KiB per round" % (name, (r1 - r0) / 200 / 1024))
def fe_same(f=[None]):
    if f[0] is None: f[0] = frontend.ScanRegistration(16)
    f[0].laserCloudHandler(raw)
hip = C.CDLL("libamdhip64.so")
def pinned_1mb():
    p = C.c_void_p(); hip.hipHostMalloc(C.byref(p), C.c_size_t(900 * 1024), 0); hip.hipHostFree(p)
def fe_create_only():
    f = frontend.ScanRegistration(16); f.close()
for name, fn in (("front-end on ONE context", fe_same), ("0.9 MB pinned, allocated and freed (HIP alone)", pinned_1mb), ("front-end context created and closed, unused", fe_create_only)):
    for _ in range(10): fn()
    gc.collect(); r0 = rss()
    for _ in range(200): fn()
    gc.collect(); r1 = rss()
    print("%-50s %7.1f KiB per round" % (name, (r1 - r0) / 200 / 1024))
from rgc_slam_amd import _lib
L = _lib.load()
def fe_raw():
    h = C.c_void_p(); L.rgc_create(0, None, C.byref(h))
    prm = _lib.FeParams(16, 0.5, 80.0, 1); o = _lib.FeOut()
    sh = np.zeros((len(raw), 5), np.float32); fp = C.POINTER(C.c_float)
    o.sharp = sh.ctypes.data_as(fp); o.flat = sh.ctypes.data_as(fp); o.inten = sh.ctypes.data_as(fp); o.feat_cap = len(raw)
    rc = L.rgc_frontend(h, raw.ctypes.data_as(fp), len(raw), 16, C.byref(prm), C.byref(o))
    assert rc == 0, rc
    L.rgc_destroy(h)
def fe_raw_cloud():
    h = C.c_void_p(); L.rgc_create(0, None, C.byref(h))
    prm = _lib.FeParams(16, 0.5, 80.0, 1); o = _lib.FeOut()
    sh = np.zeros((len(raw), 5), np.float32); cl = np.zeros((len(raw), 4), np.float32); fp = C.POINTER(C.c_float)
    o.sharp = sh.ctypes.data_as(fp); o.flat = sh.ctypes.data_as(fp); o.inten = sh.ctypes.data_as(fp); o.feat_cap = len(raw)
    o.cloud = cl.ctypes.data_as(fp); o.cloud_cap = len(raw)
    rc = L.rgc_frontend(h, raw.ctypes.data_as(fp), len(raw), 16, C.byref(prm), C.byref(o))
    assert rc == 0, rc
    L.rgc_destroy(h)
for name, fn in (("front-end through the C-ABI, feature clouds only", fe_raw), ("... and the ring-major cloud", fe_raw_cloud)):
    for _ in range(10): fn()
    gc.collect(); r0 = rss()
    for _ in range(200): fn()
    gc.collect(); r1 = rss()
    print("%-50s %7.1f KiB per round" % (name, (r1 - r0) / 200 / 1024))
print("front-end through the C-ABI, five more blocks of 200 lifecycles:")
for blk in range(5):
    gc.collect(); r0 = rss()
    for _ in range(200): fe_raw()
    gc.collect(); r1 = rss()
    print("   block %d: %7.1f KiB per round, rss %.0f MiB" % (blk, (r1 - r0) / 200 / 1024, r1 / 2**20))
