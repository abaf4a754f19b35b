"""Host-side mirror of the registration operator the RGC-SLAM odometer programs against
(``fast_gicp::FastVGICP<PointXYZI,PointXYZI>`` as driven at /root/reference/rgc_slam/src/RGC_odometer.cpp:998-1011),
backed by the HIP library through the C-ABI (include/rgc_hip.h).  Same method names, argument meaning and
error behaviour (no exceptions from the solver itself: "lm not converged" is a flag, hasConverged() like PCL);
API misuse and HIP failures raise RgcError.

This is a thin Python stand-in for the C++ adaptor (cpp/fast_vgicp_hip.hpp) so that the parity tests read like
the reference's call site.  It never computes anything on the CPU.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import DIRECT1, DIRECT7, DIRECT27, Params, RgcError, Stats


def _f32c(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a


class NeighborSearchMethod:  # include/fast_gicp/gicp/gicp_settings.hpp:8
    DIRECT27, DIRECT7, DIRECT1 = DIRECT27, DIRECT7, DIRECT1


class FastVGICP:
    """``vgicp = FastVGICP(); vgicp.setResolution(1.0); ... vgicp.setInputTarget(t); vgicp.setInputSource(s);
    aligned = vgicp.align(T2); score = vgicp.getFitnessScore(); T = vgicp.getFinalTransformation()``"""

    def __init__(self, device: int = 0):
        self._L = _lib.load()
        # constructor defaults of LsqRegistration / FastGICP / FastVGICP
        # (lsq_registration_impl.hpp:9-22, fast_gicp_impl.hpp:9-24, fast_vgicp_impl.hpp:18-25)
        self._p = _lib.default_params(max_iterations=64, translation_eps=5e-4)
        h = C.c_void_p()
        rc = self._L.rgc_create(device, C.byref(self._p), C.byref(h))
        if rc != 0:
            raise RgcError(rc, self._L.rgc_status_string(rc).decode() + " (rgc_create: is a HIP device visible?)")
        self._h = h
        self._final = np.eye(4, dtype=np.float32)
        self._H = np.eye(6)
        self._converged = False
        self._lm_failed = False
        self._iterations = 0
        self._fitness = None
        self._n_src = self._n_tgt = 0
        self._keep = {}

    # -- lifetime --------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            self._L.rgc_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            raise RgcError(rc, self._L.rgc_last_error(self._h).decode() or self._L.rgc_status_string(rc).decode())

    def _push(self):
        self._chk(self._L.rgc_set_params(self._h, C.byref(self._p)))

    # -- setters (pcl::Registration / LsqRegistration / FastGICP / FastVGICP) ----------------------
    def setResolution(self, r):                      # fast_vgicp_impl.hpp:32-34
        self._p.voxel_res = float(r); self._push()

    def setMaximumIterations(self, n):               # pcl::Registration
        self._p.max_iterations = int(n); self._push()

    def setTransformationEpsilon(self, e):           # pcl::Registration; used by is_converged (:82-91)
        self._p.translation_eps = float(e); self._push()

    def setRotationEpsilon(self, e):                 # lsq_registration_impl.hpp:27-29
        self._p.rotation_eps = float(e); self._push()

    def setInitialLambdaFactor(self, f):             # lsq_registration_impl.hpp:32-34
        self._p.lm_init_lambda_factor = float(f); self._push()

    def setCorrespondenceRandomness(self, k):        # fast_gicp_impl.hpp:41-43
        self._p.k_correspondences = int(k); self._push()

    def setNeighborSearchMethod(self, m):            # fast_vgicp_impl.hpp:37-39
        self._p.neighbor_method = int(m); self._push()

    # accepted and ignored, exactly like the reference for FastVGICP
    def setMaxCorrespondenceDistance(self, d):       # unused by FastVGICP (only kd-tree FastGICP, fast_gicp_impl.hpp:136)
        self._max_corr_dist = float(d)

    def setEuclideanFitnessEpsilon(self, e):         # no-op in LsqRegistration
        self._fitness_eps = float(e)

    def setRANSACIterations(self, n):                # no-op
        self._ransac = int(n)

    def setNumThreads(self, n):                      # fast_gicp_impl.hpp:29-37: CPU threads; nothing to set on the GPU
        self._num_threads = int(n)

    def setDebugPrint(self, on):                     # lsq_registration.hpp:53 (the LM trace on stdout, impl :59, :147): accepted, nothing is printed
        self._debug_print = bool(on)

    # -- clouds ----------------------------------------------------------------------------------
    def setInputTarget(self, cloud):                 # fast_vgicp_impl.hpp:56-63
        a = _f32c(cloud)
        if a.ndim != 2 or a.shape[1] < 3:
            raise RgcError(_lib.ERR_INVALID, "cloud must be (n, >=3) float32")
        self._chk(self._L.rgc_set_target(self._h, a.ctypes.data, a.shape[0], a.strides[0]))
        self._n_tgt = a.shape[0]
        self._fitness = None

    def setInputSource(self, cloud):                 # fast_gicp_impl.hpp:72-80
        a = _f32c(cloud)
        if a.ndim != 2 or a.shape[1] < 3:
            raise RgcError(_lib.ERR_INVALID, "cloud must be (n, >=3) float32")
        self._chk(self._L.rgc_set_source(self._h, a.ctypes.data, a.shape[0], a.strides[0]))
        self._n_src = a.shape[0]
        self._fitness = None

    def setInputTargetDevice(self, d_ptr: int, n: int, stride_bytes: int):
        self._chk(self._L.rgc_set_target_device(self._h, C.c_void_p(d_ptr), n, stride_bytes))
        self._n_tgt = n
        self._fitness = None

    def setInputSourceDevice(self, d_ptr: int, n: int, stride_bytes: int):
        self._chk(self._L.rgc_set_source_device(self._h, C.c_void_p(d_ptr), n, stride_bytes))
        self._n_src = n
        self._fitness = None

    # -- the operator ------------------------------------------------------------------------------
    def align(self, guess=None, want_output=True, want_fitness=False):
        """pcl::Registration::align(output, guess): returns the transformed source cloud (n,3) float32
        (or None if want_output=False)."""
        g = np.eye(4, dtype=np.float32) if guess is None else np.ascontiguousarray(guess, dtype=np.float32).reshape(4, 4)
        fin = np.empty(16, np.float32)
        H = np.empty(36)
        fit = C.c_double(0)
        it, conv, fail = C.c_int(0), C.c_int(0), C.c_int(0)
        fp, dp = C.POINTER(C.c_float), C.POINTER(C.c_double)
        self._chk(self._L.rgc_align(self._h, g.ctypes.data_as(fp), fin.ctypes.data_as(fp), H.ctypes.data_as(dp),
                                    C.byref(fit) if want_fitness else None, C.byref(it), C.byref(conv), C.byref(fail)))
        self._final = fin.reshape(4, 4)
        self._H = H.reshape(6, 6)
        self._iterations, self._converged, self._lm_failed = it.value, bool(conv.value), bool(fail.value)
        self._fitness = fit.value if want_fitness else None
        if not want_output:
            return None
        out = np.empty((self._n_src, 3), np.float32)
        self._chk(self._L.rgc_get_aligned(self._h, fin.ctypes.data_as(fp), out.ctypes.data_as(fp), 12))
        return out

    def shareTargetFrom(self, owner: "FastVGICP"):
        """Register to the target `owner` has prepared (setInputTarget*, or a committed RollingLocalMap) without preparing or copying it
        (rgc_share_target).  Share again whenever the owner prepares a new target."""
        self._chk(self._L.rgc_share_target(self._h, owner._h))
        self._n_tgt = getattr(owner, "_n_tgt", None)

    def setLazyTarget(self, margin_cells: int):
        """rgc_set_target_lazy: targets set from now on get covariances and voxels only within margin_cells voxels of where the scan falls
        at the solve's guess (0: everywhere, the default); results are the full build's bit for bit (a look-up outside repeats the solve)."""
        self._chk(self._L.rgc_set_target_lazy(self._h, int(margin_cells)))

    REUSE_NONE, REUSE_SEEDS, REUSE_LISTS = 0, 1, 2
    REG_NONE, REG_MIN_EIG, REG_NORMALIZED_MIN_EIG, REG_PLANE, REG_FROBENIUS = range(5)     # fast_gicp::RegularizationMethod, gicp_settings.hpp:6
    VOXEL_ADDITIVE, VOXEL_ADDITIVE_WEIGHTED, VOXEL_MULTIPLICATIVE = range(3)               # VoxelAccumulationMode, gicp_settings.hpp:10

    def setRegularizationMethod(self, method: int):
        """FastGICP::setRegularizationMethod (fast_gicp_impl.hpp:46-48).  PLANE (the odometer's) runs on the tuned kernels, every other
        method on the general route (rgc_hip.h).  Select BEFORE setting the clouds: a change drops them."""
        self._chk(self._L.rgc_set_regularization_method(self._h, int(method)))

    def setVoxelAccumulationMode(self, mode: int):
        """FastVGICP::setVoxelAccumulationMode (fast_vgicp_impl.hpp:41-43); MULTIPLICATIVE runs on the general route"""
        self._chk(self._L.rgc_set_voxel_accumulation_mode(self._h, int(mode)))

    def setNeighbourReuse(self, mode: int):
        """rgc_set_knn_reuse: what the context keeps between the targets setInputTargetReframed prepares -- REUSE_NONE (every target is
        searched like a map the library has not seen: what a caller whose map's point set changes every frame pays anyway), REUSE_SEEDS
        (the last search's k-th distances), REUSE_LISTS (seeds + neighbour lists of a verified-unchanged map; the default).  Results do
        not depend on it, bit for bit."""
        self._chk(self._L.rgc_set_knn_reuse(self._h, int(mode)))

    def getNeighbourReuse(self) -> int:
        m = C.c_int(0)
        self._chk(self._L.rgc_get_knn_reuse(self._h, C.byref(m)))
        return int(m.value)

    def holdSourceUntilTargetOf(self, other: "FastVGICP"):
        """the next setInputSource* here starts on the GPU when `other`'s target preparation (as enqueued so far) is done (rgc_hip.h)"""
        self._chk(self._L.rgc_hold_source_until_target_of(self._h, other._h))

    def align_begin(self, guess=None, want_fitness=False):
        """First half of align(): enqueue the solve and return (rgc_align_begin).  align_end() collects the result; in between the
        caller may prepare the next frame on ANOTHER FastVGICP (PipelinedVGICP below does)."""
        g = np.eye(4, dtype=np.float32) if guess is None else np.ascontiguousarray(guess, dtype=np.float32).reshape(4, 4)
        self._chk(self._L.rgc_align_begin(self._h, g.ctypes.data_as(C.POINTER(C.c_float)), 1 if want_fitness else 0))
        self._pending_fitness = want_fitness

    def align_end(self):
        """Second half of align(): waits for the solve; getFinalTransformation() etc. are valid afterwards."""
        want_fitness = getattr(self, "_pending_fitness", False)
        fin = np.empty(16, np.float32)
        H = np.empty(36)
        fit = C.c_double(0)
        it, conv, fail = C.c_int(0), C.c_int(0), C.c_int(0)
        fp, dp = C.POINTER(C.c_float), C.POINTER(C.c_double)
        self._chk(self._L.rgc_align_end(self._h, fin.ctypes.data_as(fp), H.ctypes.data_as(dp), C.byref(fit) if want_fitness else None,
                                        C.byref(it), C.byref(conv), C.byref(fail)))
        self._final = fin.reshape(4, 4)
        self._H = H.reshape(6, 6)
        self._iterations, self._converged, self._lm_failed = it.value, bool(conv.value), bool(fail.value)
        self._fitness = fit.value if want_fitness else None
        return self._final.copy()

    def align_end_reframe(self, nxt, Tw, d_map: int, n: int, stride_bytes: int, d_scratch: int):
        """align_end(), the world pose composed (Tw <- Tw * T, in place: a C-contiguous float64 4x4) and the NEXT frame's target -- the map at
        d_map re-expressed in the new body frame -- enqueued on context `nxt` in ONE call (rgc_align_end_reframe): a dependent sequence's
        turn-around between a frame's result and the next frame's first launch without Python in it.  Returns the frame's motion."""
        want_fitness = getattr(self, "_pending_fitness", False)
        fin = np.empty(16, np.float32)
        H = np.empty(36)
        fit = C.c_double(0)
        it, conv, fail = C.c_int(0), C.c_int(0), C.c_int(0)
        fp, dp = C.POINTER(C.c_float), C.POINTER(C.c_double)
        assert Tw.dtype == np.float64 and Tw.flags["C_CONTIGUOUS"] and Tw.size == 16
        self._chk(self._L.rgc_align_end_reframe(self._h, nxt._h, Tw.ctypes.data_as(dp), C.c_void_p(d_map), n, stride_bytes, C.c_void_p(d_scratch),
                                                fin.ctypes.data_as(fp), H.ctypes.data_as(dp), C.byref(fit) if want_fitness else None,
                                                C.byref(it), C.byref(conv), C.byref(fail)))
        self._final = fin.reshape(4, 4)
        self._H = H.reshape(6, 6)
        self._iterations, self._converged, self._lm_failed = it.value, bool(conv.value), bool(fail.value)
        self._fitness = fit.value if want_fitness else None
        nxt._n_tgt = n
        return self._final.copy()

    def alignedToDevice(self, d_out, stride_bytes=16, T=None):
        """the `output` cloud of align() written to a device buffer (no copy to the host, no synchronisation)"""
        t = np.ascontiguousarray(self._final if T is None else T, dtype=np.float32).reshape(16)
        self._chk(self._L.rgc_get_aligned_device(self._h, t.ctypes.data_as(C.POINTER(C.c_float)), C.c_void_p(d_out), stride_bytes))

    def getFinalTransformation(self):
        return self._final.copy()

    def getFinalHessian(self):                       # lsq_registration_impl.hpp:43-45
        return self._H.copy()

    def hasConverged(self):
        return self._converged

    @property
    def lm_failed(self):                             # "lm not converged!!" lsq_registration_impl.hpp:69-72
        return self._lm_failed

    @property
    def nr_iterations(self):                         # number of outer iterations executed
        return self._iterations

    def getFitnessScore(self):                       # RGC_odometer.cpp:1010
        if self._fitness is None:
            f = C.c_double(0)
            t = np.ascontiguousarray(self._final, dtype=np.float32)
            self._chk(self._L.rgc_fitness(self._h, t.ctypes.data_as(C.POINTER(C.c_float)), C.byref(f)))
            self._fitness = f.value
        return self._fitness

    def fitnessAt(self, T):
        """getFitnessScore's quantity at an arbitrary pose: mean squared 1-NN distance of the transformed source."""
        f = C.c_double(0)
        t = np.ascontiguousarray(T, dtype=np.float32).reshape(16)
        self._chk(self._L.rgc_fitness(self._h, t.ctypes.data_as(C.POINTER(C.c_float)), C.byref(f)))
        return f.value

    def evaluateCost(self, relative_pose, want_H=False):   # lsq_registration_impl.hpp:48-50
        T = np.ascontiguousarray(np.asarray(relative_pose, dtype=np.float32).astype(np.float64)).reshape(16)
        cost = C.c_double(0)
        dp = C.POINTER(C.c_double)
        if want_H:
            H, b = np.empty(36), np.empty(6)
            self._chk(self._L.rgc_linearize(self._h, T.ctypes.data_as(dp), H.ctypes.data_as(dp), b.ctypes.data_as(dp), C.byref(cost)))
            return cost.value, H.reshape(6, 6), b
        self._chk(self._L.rgc_linearize(self._h, T.ctypes.data_as(dp), None, None, C.byref(cost)))
        return cost.value

    # the two protected virtuals of LsqRegistration (lsq_registration.hpp:68-69), exposed for the parity tests
    def linearize(self, T):
        T = np.ascontiguousarray(T, dtype=np.float64).reshape(16)
        cost = C.c_double(0)
        H, b = np.empty(36), np.empty(6)
        dp = C.POINTER(C.c_double)
        self._chk(self._L.rgc_linearize(self._h, T.ctypes.data_as(dp), H.ctypes.data_as(dp), b.ctypes.data_as(dp), C.byref(cost)))
        return cost.value, H.reshape(6, 6), b

    def compute_error(self, T):
        T = np.ascontiguousarray(T, dtype=np.float64).reshape(16)
        cost = C.c_double(0)
        self._chk(self._L.rgc_compute_error(self._h, T.ctypes.data_as(C.POINTER(C.c_double)), C.byref(cost)))
        return cost.value

    @property
    def num_correspondences(self):
        n = C.c_int(0)
        self._chk(self._L.rgc_num_correspondences(self._h, C.byref(n)))
        return n.value

    # -- per-stage results ---------------------------------------------------------------------------
    def getSourceCovariances(self):                  # fast_gicp.hpp:63-65 (3x3 block of the reference's Matrix4d)
        cov = np.empty((self._n_src, 3, 3))
        self._chk(self._L.rgc_get_source_covariances(self._h, cov.ctypes.data_as(C.POINTER(C.c_double)), None))
        return cov

    def getTargetCovariances(self):                  # fast_gicp.hpp:67-69
        cov = np.empty((self._n_tgt, 3, 3))
        self._chk(self._L.rgc_get_target_covariances(self._h, cov.ctypes.data_as(C.POINTER(C.c_double)), None))
        return cov

    def setSourceCovariances(self, cov):             # fast_gicp_impl.hpp:93-95 (plane-regularised covariances only, see rgc_hip.h)
        c = np.ascontiguousarray(cov, dtype=np.float64).reshape(-1, 9)
        self._chk(self._L.rgc_set_source_covariances(self._h, c.ctypes.data_as(C.POINTER(C.c_double)), c.shape[0]))

    def setTargetCovariances(self, cov):             # fast_gicp_impl.hpp:98-100
        c = np.ascontiguousarray(cov, dtype=np.float64).reshape(-1, 9)
        self._chk(self._L.rgc_set_target_covariances(self._h, c.ctypes.data_as(C.POINTER(C.c_double)), c.shape[0]))

    def swapSourceAndTarget(self):                   # fast_vgicp_impl.hpp:46-53
        self._chk(self._L.rgc_swap_source_and_target(self._h))
        self._n_src, self._n_tgt = self._n_tgt, self._n_src

    def clearSource(self):                           # fast_gicp_impl.hpp:60-63
        self._chk(self._L.rgc_clear_source(self._h))

    def clearTarget(self):                           # fast_gicp_impl.hpp:66-69
        self._chk(self._L.rgc_clear_target(self._h))

    def getSourceNormals(self):
        n = np.empty((self._n_src, 3))
        self._chk(self._L.rgc_get_source_covariances(self._h, None, n.ctypes.data_as(C.POINTER(C.c_double))))
        return n

    def getTargetNormals(self):
        n = np.empty((self._n_tgt, 3))
        self._chk(self._L.rgc_get_target_covariances(self._h, None, n.ctypes.data_as(C.POINTER(C.c_double))))
        return n

    def getVoxels(self):
        """Gaussian voxel map sorted by (cx,cy,cz): dict(coords, num, mean, cov)."""
        cnt = C.c_int(0)
        self._chk(self._L.rgc_get_voxels(self._h, 0, None, None, None, None, C.byref(cnt)))
        V = cnt.value
        coords, num = np.empty((V, 3), np.int32), np.empty(V, np.int32)
        mean, cov = np.empty((V, 3)), np.empty((V, 3, 3))
        ip, dp = C.POINTER(C.c_int), C.POINTER(C.c_double)
        self._chk(self._L.rgc_get_voxels(self._h, V, coords.ctypes.data_as(ip), num.ctypes.data_as(ip), mean.ctypes.data_as(dp),
                                         cov.ctypes.data_as(dp), C.byref(cnt)))
        o = np.lexsort((coords[:, 2], coords[:, 1], coords[:, 0]))
        return dict(coords=coords[o], num=num[o], mean=mean[o], cov=cov[o])

    def stats(self) -> dict:
        s = Stats()
        self._chk(self._L.rgc_get_stats(self._h, C.byref(s)))
        return {k: getattr(s, k) for k, _ in Stats._fields_}

    # -- profiling plumbing ----------------------------------------------------------------------------
    def profile_enable(self, on=True):
        self._chk(self._L.rgc_profile_enable(self._h, 1 if on else 0))

    def profile_select(self, kinds=None):
        """Only time the named kinds (e.g. ["knn_cov_target"]); None = all."""
        mask = 0xFFFFFFFF
        if kinds is not None:
            names = [self._L.rgc_profile_name(k).decode() for k in range(9)]
            mask = 0
            for k in kinds:
                mask |= 1 << names.index(k)
        self._chk(self._L.rgc_profile_select(self._h, mask))

    def profile_reset(self):
        self._chk(self._L.rgc_profile_reset(self._h))

    def profile(self) -> dict:
        out = {}
        for kind in range(_lib.K_COUNT):
            n, ms, pts = C.c_longlong(0), C.c_double(0), C.c_longlong(0)
            self._chk(self._L.rgc_profile_get(self._h, kind, C.byref(n), C.byref(ms), C.byref(pts)))
            out[self._L.rgc_profile_name(kind).decode()] = dict(launches=n.value, total_ms=ms.value, points=pts.value)
        return out

    def synchronize(self):
        self._chk(self._L.rgc_synchronize(self._h))

    # device memory helpers (clouds resident in HBM)
    def device_alloc(self, nbytes: int) -> int:
        p = C.c_void_p()
        self._chk(self._L.rgc_device_alloc(self._h, nbytes, C.byref(p)))
        return p.value

    def device_free(self, ptr: int):
        self._chk(self._L.rgc_device_free(self._h, C.c_void_p(ptr)))

    def transformCloudDevice(self, d_in: int, n: int, stride_bytes: int, q_xyzw, t, d_out: int):
        """B9 on device-resident clouds (RGC_odometer.cpp:1495-1514): q * p + t in fp64 -> n x 4 floats at d_out, stream-ordered."""
        q = np.ascontiguousarray(q_xyzw, dtype=np.float64)
        tt = np.ascontiguousarray(t, dtype=np.float64)
        dp = C.POINTER(C.c_double)
        self._chk(self._L.rgc_transform_cloud(self._h, C.c_void_p(d_in), n, stride_bytes, q.ctypes.data_as(dp), tt.ctypes.data_as(dp),
                                              C.c_void_p(d_out), 1))

    def setInputTargetReframed(self, d_map: int, n: int, stride_bytes: int, q_xyzw, t, d_scratch: int):
        """B9 + setInputTarget on the device in one call (rgc_set_target_reframed): the sub-map at d_map re-expressed by (q, t) into
        d_scratch and prepared as the target, its bounding box derived from the map's and the transform (RGC_odometer.cpp:1248-1256, 998-1007)"""
        qt = getattr(self, "_qt7", None)
        if qt is None:
            qt = self._qt7 = (C.c_double * 7)()   # (the call sits on a sequence's critical path: no numpy temporaries)
            self._qt7_q = C.cast(qt, C.POINTER(C.c_double))
            self._qt7_t = C.cast(C.byref(qt, 4 * C.sizeof(C.c_double)), C.POINTER(C.c_double))
        qt[0], qt[1], qt[2], qt[3] = float(q_xyzw[0]), float(q_xyzw[1]), float(q_xyzw[2]), float(q_xyzw[3])
        qt[4], qt[5], qt[6] = float(t[0]), float(t[1]), float(t[2])
        self._chk(self._L.rgc_set_target_reframed(self._h, C.c_void_p(d_map), n, stride_bytes, self._qt7_q, self._qt7_t, C.c_void_p(d_scratch)))
        self._n_tgt = n

    def download(self, ptr: int, shape, dtype=np.float32) -> np.ndarray:
        out = np.empty(shape, dtype)
        self._chk(self._L.rgc_download(self._h, out.ctypes.data, C.c_void_p(ptr), out.nbytes))
        return out

    def upload(self, ptr: int, arr: np.ndarray):
        a = np.ascontiguousarray(arr)
        self._chk(self._L.rgc_upload(self._h, C.c_void_p(ptr), a.ctypes.data, a.nbytes))
        self._chk(self._L.rgc_synchronize(self._h))

    def upload_async(self, ptr: int, arr: np.ndarray):
        """rgc_upload without the synchronisation: enqueued on the context's main stream; `arr` (C-contiguous, ideally page-locked) must
        stay alive and unchanged until the stream has passed it"""
        assert arr.flags["C_CONTIGUOUS"]
        self._chk(self._L.rgc_upload(self._h, C.c_void_p(ptr), arr.ctypes.data, arr.nbytes))


def odometer_vgicp(device: int = 0) -> FastVGICP:
    """A FastVGICP configured exactly as the odometer does (RGC_odometer.cpp:998-1006)."""
    v = FastVGICP(device)
    v.setResolution(1.0)                 # down_simple_vgicp, :308,1000
    v.setMaximumIterations(25)           # :1001
    v.setMaxCorrespondenceDistance(2)    # :1002
    v.setTransformationEpsilon(1e-6)     # :1003
    v.setEuclideanFitnessEpsilon(1e-6)   # :1004
    v.setRANSACIterations(0)             # :1005
    v.setNumThreads(14)                  # :1006
    return v


class PipelinedVGICP:
    """Scan-to-map registration of a SEQUENCE on `depth` contexts taking turns.  A frame is cloud preparation (grids, kNN covariances,
    voxel map: throughput-bound, most of the frame) followed by the LM solve (a chain of short launches that leaves the chip mostly
    idle) -- and only the solve needs the previous frame's pose.  So while frame i is being solved on one context the clouds of frames
    i + 1 .. i + depth - 1 are being prepared on the others; every frame runs the same kernels on the same inputs as FastVGICP.align()
    one frame at a time: results are identical, the frames just overlap on the GPU (align_begin / align_end, include/rgc_hip.h)."""

    def __init__(self, device: int = 0, make=odometer_vgicp, depth: int = 2, contexts=None):
        """contexts: FastVGICP objects to use as the first contexts (e.g. the one a RollingLocalMap lives on); they are not closed here"""
        if depth < 2:
            raise ValueError("depth >= 2")
        given = list(contexts or [])
        self._owned = [make(device) for _ in range(depth - len(given))]
        self.v = given + self._owned

    def close(self):
        for v in self._owned:
            v.close()

    def synchronize(self):
        for v in self.v:
            v.synchronize()

    def share_target(self):
        """The other contexts register to the target context 0 holds (setInputTarget* or a committed RollingLocalMap on self.v[0]) without
        preparing it again: for sequences whose map does not change every frame; then set_clouds of run() sets only the source.
        Call again after context 0 prepared a new target."""
        for w in self.v[1:]:
            w.shareTargetFrom(self.v[0])

    def run(self, n_frames, set_clouds, guess0, want_fitness=False, next_guess=None, on_result=None):
        """set_clouds(i, v): set target and source of frame i on the FastVGICP `v` (device-resident or host clouds).
        The guess of frame 0 is guess0; that of frame i + 1 is next_guess(i, T_i) (default: T_i, frame i's final transformation).
        on_result(i, v) is called when frame i is done (v holds its results until it is given frame i + depth).  Returns the final
        transformations."""
        out = []
        D = len(self.v)
        for j in range(min(D - 1, n_frames)):
            set_clouds(j, self.v[j % D])
        g = guess0
        for i in range(n_frames):
            cur = self.v[i % D]
            cur.align_begin(g, want_fitness)
            j = i + D - 1                       # the context of frame i - 1 is free: the frame D - 1 ahead goes there
            if j < n_frames:
                set_clouds(j, self.v[j % D])    # ... and is prepared on the GPU while this frame (and the next) are solved
            T = cur.align_end()
            if on_result is not None:
                on_result(i, cur)
            g = T if next_guess is None else next_guess(i, T)
            out.append(T)
        return out
