"""Everything on ONE context, interleaved at random (rgc::OdometryNode drives one context too, in one fixed order): the front-end, the leaf
filter (blocking and in its two halves), de-skew, re-framing, the registration's clouds / solves / getters / aligned cloud, the wire
unpacking -- each call's result against the same call on a context of its own.  Shared staging buffers, streams and scratch are what
this is after.      python tests/fuzz/fuzz_one_context.py [trials] [seed] [operations per trial]"""
import sys, os, json, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import rgc_slam_amd.synth as synth
from rgc_slam_amd import registration as reg, odometry, frontend, wire, _lib

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 20
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
n_ops = int(sys.argv[3]) if len(sys.argv) > 3 else 40
rep = {"trials": 0, "operations": {}, "compared": 0, "failures": []}
t0 = time.time()


def share(obj, owner):
    """obj gives up its own context and works on owner's"""
    obj._L.rgc_destroy(obj._h)
    obj._h = owner._h
    obj.close = lambda: None
    return obj


def fe_equal(a, b):
    keys = ("cloud", "sharp", "flat", "inten", "ground_pts", "ring_count", "label", "picked", "curvature")
    return a["n_cloud"] == b["n_cloud"] and a["ground_valid"] == b["ground_valid"] and all(np.array_equal(a[k], b[k]) for k in keys if k in a and k in b) and \
        np.array_equal(a["groundparam"], b["groundparam"])


for trial in range(trials):
    rng = np.random.default_rng(seed0 * 179424673 + trial)
    world = synth.make_world(half_extent=40.0, seed=int(rng.integers(1, 1 << 30)))
    v = reg.odometer_vgicp(0)
    pre, fe, wr = share(odometry.Preprocessor(0), v), share(frontend.ScanRegistration(16), v), share(wire.Wire(0), v)
    r_v, r_pre, r_fe, r_wr = reg.odometer_vgicp(0), odometry.Preprocessor(0), frontend.ScanRegistration(16), wire.Wire(0)   # contexts of their own
    tgt = src = None
    vg_open = None
    tag = {"trial": trial}
    L = v._L

    def sweep():
        T = synth.se3(synth.rot_zyx(rng.uniform(-np.pi, np.pi), rng.normal(0, 0.01), rng.normal(0, 0.01)), rng.uniform(-5, 5, 3) * np.array([1, 1, 0.01]))
        sc = synth.make_scan(world, T, n_az=int(rng.integers(200, 1500)), seed=int(rng.integers(1, 1 << 30)))
        return sc
    try:
        for op_i in range(n_ops):
            op = str(rng.choice(["frontend", "voxelgrid", "vg_begin", "vg_end", "deskew", "transform", "set_target", "set_source", "align", "getters", "aligned", "unpack"]))
            tag.update(op=op, op_i=op_i)
            rep["operations"][op] = rep["operations"].get(op, 0) + 1
            sc = sweep()
            xyzi = np.concatenate([sc["xyz"], (sc["ring"] + 0.1 * sc["rel_time"])[:, None].astype(np.float32)], axis=1).astype(np.float32)
            if op == "frontend":
                raw = np.concatenate([sc["xyz"], sc["intensity"][:, None]], axis=1).astype(np.float32)
                a, b = fe.laserCloudHandler(raw), r_fe.laserCloudHandler(raw)
                rep["compared"] += 1
                if not fe_equal(a, b):
                    rep["failures"].append(dict(tag, error="front-end differs on the shared context"))
            elif op == "voxelgrid":
                if vg_open is not None:
                    continue
                leaf = float(rng.choice([0.2, 0.3, 0.5]))
                a, b = pre.voxelGridFilter(xyzi, leaf), r_pre.voxelGridFilter(xyzi, leaf)
                rep["compared"] += 1
                if not np.array_equal(a, b):
                    rep["failures"].append(dict(tag, error="leaf filter differs on the shared context"))
            elif op == "vg_begin":
                if vg_open is not None:
                    continue
                leaf = float(rng.choice([0.2, 0.3]))
                d_in, d_out = v.device_alloc(xyzi.nbytes), v.device_alloc(xyzi.nbytes)
                v.upload(d_in, xyzi)
                rc = L.rgc_voxelgrid_begin(v._h, C.c_void_p(d_in), len(xyzi), 16, C.c_float(leaf), C.c_void_p(d_out))
                if rc != 0:
                    rep["failures"].append(dict(tag, error="rgc_voxelgrid_begin refused: %s" % L.rgc_last_error(v._h).decode()[:120]))
                    v.device_free(d_in); v.device_free(d_out)
                else:
                    vg_open = (d_in, d_out, xyzi.copy(), leaf)
            elif op == "vg_end":
                n = C.c_int(0)
                rc = L.rgc_voxelgrid_end(v._h, C.byref(n))
                if vg_open is None:
                    if rc == 0:
                        rep["failures"].append(dict(tag, error="rgc_voxelgrid_end without a begin worked"))
                    continue
                d_in, d_out, cloud, leaf = vg_open
                vg_open = None
                if rc != 0:
                    rep["failures"].append(dict(tag, error="rgc_voxelgrid_end refused: %s" % L.rgc_last_error(v._h).decode()[:120]))
                else:
                    got = v.download(d_out, (n.value, 4))
                    rep["compared"] += 1
                    if not np.array_equal(got, r_pre.voxelGridFilter(cloud, leaf)):
                        rep["failures"].append(dict(tag, error="the two-halves leaf filter differs"))
                v.device_free(d_in); v.device_free(d_out)
            elif op == "deskew":
                R = synth.rot_zyx(*(rng.normal(0, 0.02, 3)))
                qw = np.sqrt(1 + np.trace(R)) / 2
                q = np.array([(R[2, 1] - R[1, 2]) / (4 * qw), (R[0, 2] - R[2, 0]) / (4 * qw), (R[1, 0] - R[0, 1]) / (4 * qw), qw]); t = rng.normal(0, 0.2, 3)
                rep["compared"] += 1
                if not np.array_equal(pre.adjustDistortion(xyzi, q, t), r_pre.adjustDistortion(xyzi, q, t)):
                    rep["failures"].append(dict(tag, error="de-skew differs on the shared context"))
            elif op == "transform":
                q = rng.normal(0, 1, 4); q /= np.linalg.norm(q); t = rng.uniform(-30, 30, 3)
                rep["compared"] += 1
                if not np.array_equal(pre.transformPointCloud(xyzi, q, t), r_pre.transformPointCloud(xyzi, q, t)):
                    rep["failures"].append(dict(tag, error="re-framing differs on the shared context"))
            elif op == "set_target":
                c = r_pre.voxelGridFilter(xyzi, 0.3)[:, :3].copy()
                if len(c) >= 20:
                    v.setInputTarget(c); r_v.setInputTarget(c); tgt = c
            elif op == "set_source":
                c = r_pre.voxelGridFilter(xyzi, 0.2)[:, :3].copy()
                if len(c) >= 20:
                    v.setInputSource(c); r_v.setInputSource(c); src = c
            elif op == "align":
                if tgt is None or src is None:
                    continue
                g = np.eye(4, dtype=np.float32)
                v.align(g, want_output=False, want_fitness=True); r_v.align(g, want_output=False, want_fitness=True)
                rep["compared"] += 1
                a, b = v.getFinalTransformation(), r_v.getFinalTransformation()
                if not (np.array_equal(a, b, equal_nan=True) and v.nr_iterations == r_v.nr_iterations):
                    rep["failures"].append(dict(tag, error="solve differs on the shared context", dT=float(np.nanmax(np.abs(a - b))), it=[int(v.nr_iterations), int(r_v.nr_iterations)]))
            elif op == "getters":
                if tgt is not None:
                    rep["compared"] += 1
                    if not np.array_equal(v.getTargetCovariances(), r_v.getTargetCovariances()):
                        rep["failures"].append(dict(tag, error="target covariances differ on the shared context"))
                if src is not None:
                    rep["compared"] += 1
                    if not np.array_equal(v.getSourceCovariances(), r_v.getSourceCovariances()):
                        rep["failures"].append(dict(tag, error="source covariances differ on the shared context"))
            elif op == "aligned":
                if src is None:
                    continue
                T = synth.se3(synth.rot_zyx(*(rng.normal(0, 0.1, 3))), rng.normal(0, 1, 3)).astype(np.float32)
                fp = C.POINTER(C.c_float)
                o1, o2 = np.empty((len(src), 3), np.float32), np.empty((len(src), 3), np.float32)
                rc1 = L.rgc_get_aligned(v._h, T.ctypes.data_as(fp), o1.ctypes.data_as(fp), 12)
                rc2 = L.rgc_get_aligned(r_v._h, T.ctypes.data_as(fp), o2.ctypes.data_as(fp), 12)
                rep["compared"] += 1
                if rc1 != rc2 or (rc1 == 0 and not np.array_equal(o1, o2)):
                    rep["failures"].append(dict(tag, error="the aligned cloud differs on the shared context", rc=[rc1, rc2]))
            elif op == "unpack":
                n = int(rng.integers(10, 5000))
                dt = np.dtype({"names": ["x", "y", "z", "intensity", "ring"], "formats": ["<f4", "<f4", "<f4", "<f4", "<u2"], "offsets": [0, 4, 8, 16, 20], "itemsize": 32})
                a = np.zeros(n, dt)
                for kf in ("x", "y", "z", "intensity"):
                    a[kf] = rng.normal(0, 20, n).astype(np.float32)
                a["ring"] = rng.integers(0, 16, n)
                lay = wire.layout(32, dict(x=(0, 7), y=(4, 7), z=(8, 7), intensity=(16, 7), ring=(20, 4)))
                x1, r1, _ = wr.unpack(a.tobytes(), n, lay, want_ring=True)
                rep["compared"] += 1
                if not (np.array_equal(x1, np.stack([a["x"], a["y"], a["z"], a["intensity"]], axis=1)) and np.array_equal(r1, a["ring"].astype(np.int32))):
                    rep["failures"].append(dict(tag, error="unpacking differs on the shared context"))
            if len(rep["failures"]) > 10:
                break
        if vg_open is not None:
            n = C.c_int(0); L.rgc_voxelgrid_end(v._h, C.byref(n)); v.device_free(vg_open[0]); v.device_free(vg_open[1])
    except Exception as e:
        import traceback
        rep["failures"].append(dict(tag, error="exception: %r" % (e,), where=traceback.format_exc()[-500:]))
    for o in (r_v, r_pre, r_fe, r_wr):
        o.close()
    pre._h = fe._h = wr._h = None
    v.close()
    rep["trials"] += 1
    if len(rep["failures"]) > 10:
        break
rep["wall_s"] = round(time.time() - t0, 1)
print(json.dumps(rep))
