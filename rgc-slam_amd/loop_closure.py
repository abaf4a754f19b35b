"""Host-side mirror of the loop-closure ICP: pcl::IterativeClosestPoint as configured and used at
src/RGC_mapping.cpp:2050-2088, behind rgc_icp_align (include/rgc_hip.h).

    icp = IterativeClosestPoint(device=0)
    icp.setMaxCorrespondenceDistance(poseGraphSearchRadius * 2); icp.setMaximumIterations(100)
    icp.setTransformationEpsilon(1e-6); icp.setEuclideanFitnessEpsilon(1e-6)
    icp.setInputSource(latestKeyFrameCloud); icp.setInputTarget(nearHistoryKeyFrameCloud)
    icp.align()
    ok = icp.hasConverged() and icp.getFitnessScore() <= historyKeyframeFitnessScore
    T_drift = icp.getFinalTransformation()

No CPU fallback: without librgc_hip.so / an MI355X this raises."""
import ctypes as C

import numpy as np

from . import _lib

STATES = ("not_converged", "iterations", "transform", "abs_mse", "rel_mse", "no_correspondences")


class IterativeClosestPoint:
    def __init__(self, device: int = 0):
        self._L = _lib.load()
        h = C.c_void_p()
        rc = self._L.rgc_create(device, None, C.byref(h))
        if rc:
            raise _lib.RgcError(rc, self._L.rgc_status_string(rc).decode())
        self._h = h
        self._p = _lib.IcpParams()
        self._L.rgc_default_icp_params(C.byref(self._p))
        self._src = self._tgt = None
        self._res, self._T = None, np.eye(4, dtype=np.float32)

    def close(self):
        if getattr(self, "_h", None):
            self._L.rgc_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def setMaxCorrespondenceDistance(self, d): self._p.max_correspondence_distance = float(d)
    def setMaximumIterations(self, n): self._p.max_iterations = int(n)
    def setTransformationEpsilon(self, e): self._p.transformation_epsilon = float(e)
    def setEuclideanFitnessEpsilon(self, e): self._p.euclidean_fitness_epsilon = float(e)
    def setRANSACIterations(self, n): pass  # 0 in the reference (:2056): no outlier rejection stage

    def setInputSource(self, cloud): self._src = np.ascontiguousarray(cloud, dtype=np.float32)
    def setInputTarget(self, cloud): self._tgt = np.ascontiguousarray(cloud, dtype=np.float32)

    def align(self):
        if self._src is None or self._tgt is None:
            raise ValueError("setInputSource / setInputTarget first")
        if self._src.shape[1] != self._tgt.shape[1]:
            raise ValueError("source and target must have the same point layout")
        T = np.zeros(16, np.float32)
        res = _lib.IcpResult()
        fp = C.POINTER(C.c_float)
        rc = self._L.rgc_icp_align(self._h, self._src.ctypes.data_as(fp), self._src.shape[0], self._tgt.ctypes.data_as(fp), self._tgt.shape[0],
                                   4 * self._src.shape[1], C.byref(self._p), T.ctypes.data_as(fp), C.byref(res))
        if rc:
            raise _lib.RgcError(rc, self._L.rgc_last_error(self._h).decode() or self._L.rgc_status_string(rc).decode())
        self._T, self._res = T.reshape(4, 4), res
        return self._T

    def hasConverged(self): return bool(self._res.converged)
    def getFitnessScore(self): return float(self._res.fitness)
    def getFinalTransformation(self): return self._T
    @property
    def nr_iterations(self): return int(self._res.iterations)
    @property
    def convergence_state(self): return STATES[self._res.state]
