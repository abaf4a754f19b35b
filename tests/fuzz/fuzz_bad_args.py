"""One bad argument at a time: each of the registration's and the pre-stages' entry points called correctly once, then with every argument in
turn replaced by something it must not be -- NULL, 0, -1, a huge count, a stride that is no stride, NaN / inf for a number -- on a context
that holds clouds.  A status or a result, never a crash; and the context still solves afterwards, with the same result as before.
    python tests/fuzz/fuzz_bad_args.py"""
import sys, os, json, ctypes as C, faulthandler
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
faulthandler.enable()
import numpy as np
from rgc_slam_amd import _lib, registration as reg
import rgc_slam_amd.synth as synth

L0 = _lib.load()
raw = C.CDLL(_lib.LIB_PATH)
world, base = synth.make_world_and_map(8000, seed=3)
tgt = np.ascontiguousarray(base, np.float32)
src = np.ascontiguousarray(base[::3] + np.float32(0.01), np.float32)
v = reg.odometer_vgicp(0)
v.setInputTarget(tgt); v.setInputSource(src)
v.align(np.eye(4, dtype=np.float32), want_output=False)
T_ref = v.getFinalTransformation().copy()
h = v._h
vp, ci, cf, cd = C.c_void_p, C.c_int, C.c_float, C.c_double
nt, ns = len(tgt), len(src)
T16f = np.eye(4, dtype=np.float32); T16d = np.eye(4)
H36, b6, cost, fit = np.zeros(36), np.zeros(6), np.zeros(1), np.zeros(1)
cov_t, nrm_t = np.zeros(nt * 9), np.zeros(nt * 3)
out_s = np.zeros((ns, 3), np.float32)
xyzi = np.zeros((nt, 4), np.float32); xyzi[:, :3] = tgt
vg_out, n_out = np.zeros((nt, 4), np.float32), np.zeros(1, np.int32)
q4, t3 = np.array([0, 0, 0, 1.0]), np.zeros(3)
deskew_buf = xyzi.copy()
cap = 4096
vx_c, vx_n, vx_m, vx_v, vx_cnt = np.zeros(cap * 3, np.int32), np.zeros(cap, np.int32), np.zeros(cap * 3), np.zeros(cap * 9), np.zeros(1, np.int32)
d_buf = v.device_alloc(xyzi.nbytes); v.upload(d_buf, xyzi)
msg = np.zeros((nt, 8), np.float32); msg[:, :3] = tgt
lay = _lib.Pc2Layout(); lay.point_step = 32
for i_, (o_, t_) in enumerate(((0, 7), (4, 7), (8, 7), (16, 7), (-1, 0), (-1, 0))):
    lay.offset[i_], lay.datatype[i_] = o_, t_
ring_out, time_out = np.zeros(nt, np.int32), np.zeros(nt, np.float32)
msg_out = np.zeros(nt * 48, np.uint8)


class IcpParams(C.Structure):
    _fields_ = [("max_iterations", ci), ("max_correspondence_distance", cd), ("transformation_epsilon", cd), ("euclidean_fitness_epsilon", cd)]


class IcpResult(C.Structure):
    _fields_ = [("iterations", ci), ("converged", ci), ("state", ci), ("n_correspondences", ci), ("fitness", cd)]


icp_p, icp_r = IcpParams(20, 2.0, 1e-6, 1e-6), IcpResult()
feat = np.zeros((500, 4), np.float32); feat[:, :3] = tgt[:500]
fac8, n_valid, kid, n_ev = np.zeros(500 * 8), np.zeros(1, np.int32), np.zeros(1, np.int32), np.zeros(1, np.int32)
big_out = np.zeros((4 * nt + 64, 4), np.float32)
fe_p = _lib.FeParams(16, 0.5, 80.0, 1)
fe_o = _lib.FeOut()      # (the optional per-point outputs NULL; the feature clouds need room)
fe_sharp, fe_flat, fe_inten = (np.zeros((nt, 5), np.float32) for _ in range(3))
fp_ = C.POINTER(C.c_float)
fe_o.sharp, fe_o.flat, fe_o.inten, fe_o.feat_cap = fe_sharp.ctypes.data_as(fp_), fe_flat.ctypes.data_as(fp_), fe_inten.ctypes.data_as(fp_), nt
d_out = v.device_alloc(xyzi.nbytes)
P = lambda a: a.ctypes.data


def cases():
    # (name, argtypes, good arguments, which positions to corrupt (position 0 is the context))
    yield "rgc_set_target", [vp, vp, ci, ci], [h, P(tgt), nt, 12]
    yield "rgc_set_source", [vp, vp, ci, ci], [h, P(src), ns, 12]
    yield "rgc_set_target_device", [vp, vp, ci, ci], [h, d_buf, nt, 16]
    yield "rgc_set_source_device", [vp, vp, ci, ci], [h, d_buf, nt, 16]
    yield "rgc_set_target_reframed", [vp, vp, ci, ci, vp, vp, vp], [h, d_buf, nt, 16, P(q4), P(t3), d_out]
    yield "rgc_linearize", [vp, vp, vp, vp, vp], [h, P(T16d), P(H36), P(b6), P(cost)]
    yield "rgc_compute_error", [vp, vp, vp], [h, P(T16d), P(cost)]
    yield "rgc_fitness", [vp, vp, vp], [h, P(T16f), P(fit)]
    yield "rgc_get_aligned", [vp, vp, vp, ci], [h, P(T16f), P(out_s), 12]
    yield "rgc_get_target_covariances", [vp, vp, vp], [h, P(cov_t), P(nrm_t)]
    yield "rgc_set_target_covariances", [vp, vp, ci], [h, P(cov_t), nt]
    yield "rgc_get_voxels", [vp, ci, vp, vp, vp, vp, vp], [h, cap, P(vx_c), P(vx_n), P(vx_m), P(vx_v), P(vx_cnt)]
    yield "rgc_align", [vp, vp, vp, vp, vp, vp, vp, vp], [h, P(T16f), P(np.zeros(16, np.float32)), P(H36), P(fit), P(np.zeros(1, np.int32)), P(np.zeros(1, np.int32)), P(np.zeros(1, np.int32))]
    yield "rgc_align_begin", [vp, vp, ci], [h, P(T16f), 1]
    # (the last argument of these three says which memory the pointers are in: a caller who gets THAT wrong hands device code a host address --
    #  nothing a library can check for the price of a call; left alone)
    yield "rgc_voxelgrid", [vp, vp, ci, ci, cf, vp, vp, ci], [h, P(xyzi), nt, 16, 0.3, P(vg_out), P(n_out), 0], (7,)
    yield "rgc_deskew", [vp, vp, ci, ci, vp, vp, ci], [h, P(deskew_buf), nt, 16, P(q4), P(t3), 0], (6,)
    yield "rgc_transform_cloud", [vp, vp, ci, ci, vp, vp, vp, ci], [h, P(xyzi), nt, 16, P(q4), P(t3), P(vg_out), 0], (7,)
    # ---- the rows either side: wire, loop-closure ICP, the mapping node's registration, the resident map ----
    yield "rgc_pc2_unpack", [vp, vp, ci, vp, vp, vp, vp, ci], [h, P(msg), nt, C.addressof(lay), P(vg_out), P(ring_out), P(time_out), 0], (7,)
    yield "rgc_pc2_pack", [vp, ci, vp, ci, ci, vp], [h, 0, P(xyzi), nt, 0, P(msg_out)], (4,)
    yield "rgc_icp_align", [vp, vp, ci, vp, ci, ci, vp, vp, vp], [h, P(src), ns, P(tgt), nt, 12, C.addressof(icp_p), P(np.zeros(16, np.float32)), C.addressof(icp_r)]
    yield "rgc_mapreg_set_maps", [vp, vp, ci, vp, ci, ci], [h, P(xyzi), nt, P(xyzi), nt, 16]
    yield "rgc_mapreg_associate", [vp, ci, vp, ci, vp, vp, vp, vp], [h, 1, P(feat), len(feat), P(q4), P(t3), P(fac8), P(n_valid)]
    yield "rgc_map_reset", [vp, vp], [h, P(t3)]
    yield "rgc_map_insert", [vp, vp, ci, ci, vp, vp, ci, vp], [h, P(xyzi), nt, 16, P(q4), P(t3), 0, P(kid)], (6,)
    yield "rgc_map_evict", [vp, ci, vp, cd, vp], [h, 2, P(t3), 5.0, P(n_ev)]
    yield "rgc_map_rebase", [vp, vp], [h, P(t3)]
    yield "rgc_map_commit", [vp, cf, vp], [h, 0.3, P(n_out)]
    yield "rgc_map_download", [vp, ci, vp, ci, vp], [h, 0, P(big_out), len(big_out), P(n_out)]
    yield "rgc_frontend", [vp, vp, ci, ci, vp, vp], [h, P(xyzi), nt, 16, C.addressof(fe_p), C.addressof(fe_o)]
    yield "rgc_set_target_lazy", [vp, ci], [h, 2]
    yield "rgc_set_knn_reuse", [vp, ci], [h, 1]
    yield "rgc_set_regularization_method", [vp, ci], [h, 3]
    yield "rgc_set_voxel_accumulation_mode", [vp, ci], [h, 0]


def restore():
    """bring the context back to the reference state (a refused call may have cleared a cloud: that is its right)"""
    n = C.c_int(0)
    raw.rgc_align_end.argtypes = [vp] * 7
    raw.rgc_align_end(h, None, None, None, None, None, None)     # (if a begin went through: collect it)
    v.setLazyTarget(0); v.setNeighbourReuse(2); v.setRegularizationMethod(3); v.setVoxelAccumulationMode(0)
    v.setInputTarget(tgt); v.setInputSource(src)
    v.linearize(np.eye(4))     # (rgc_compute_error works on the correspondences the last linearisation froze)


rep = {"functions": 0, "calls": 0, "bad_calls_that_returned_ok": [], "failures": []}
for case in cases():
    name, types, good = case[:3]
    leave = case[3] if len(case) > 3 else ()
    fn = getattr(raw, name)
    fn.argtypes = types; fn.restype = ci
    rep["functions"] += 1
    restore()
    r0 = fn(*good)
    rep["calls"] += 1
    if r0 != 0:
        rep["failures"].append(dict(function=name, error="the good call failed", status=int(r0), message=L0.rgc_last_error(h).decode()[:120]))
    for pos in range(len(good)):
        if pos in leave:
            continue
        t = types[pos]
        if t is vp:
            bads = [None]
        elif t is ci:
            bads = [0, -1, 1 << 30, 7] if pos else []
        else:
            bads = [0.0, -1.0, float("nan"), float("inf")]
        for bad in bads:
            args = list(good); args[pos] = bad
            restore()
            print("calling", name, "argument", pos, "=", bad, flush=True, file=sys.stderr)
            r = fn(*args)
            rep["calls"] += 1
            if r == 0 and not (t is ci and bad == good[pos]):
                rep["bad_calls_that_returned_ok"].append("%s(arg %d = %s)" % (name, pos, bad))
restore()
v.align(np.eye(4, dtype=np.float32), want_output=False)
rep["same_result_afterwards"] = bool(np.array_equal(v.getFinalTransformation(), T_ref))
if not rep["same_result_afterwards"]:
    rep["failures"].append(dict(error="the context does not solve as before"))
v.device_free(d_buf); v.device_free(d_out); v.close()
print(json.dumps(rep))
