"""ctypes binding of librgc_hip.so (the C-ABI declared in include/rgc_hip.h).

The library is the product: if it is missing or cannot be loaded this module raises -- there is no CPU
fallback anywhere in this package.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RGC_HIP_LIB") or os.path.join(_HERE, "librgc_hip.so")  # RGC_HIP_LIB: developer override (A/B builds)

DIRECT27, DIRECT7, DIRECT1 = 0, 1, 2

K_GRID, K_KNN_COV, K_VOXEL, K_LINEARIZE, K_ERROR, K_FITNESS, K_KNN_COV_SRC, K_KNN_COOP, K_KNN_COOP_SRC, K_COUNT = range(10)

OK = 0
ERR_INVALID, ERR_HIP, ERR_TOO_FEW_POINTS, ERR_GRID_TOO_LARGE, ERR_NO_INPUT, ERR_NONFINITE, ERR_UNSUPPORTED = -1, -2, -3, -4, -5, -6, -7


class Params(C.Structure):
    _fields_ = [("voxel_res", C.c_double), ("max_iterations", C.c_int), ("lm_max_iterations", C.c_int),
                ("rotation_eps", C.c_double), ("translation_eps", C.c_double), ("lm_init_lambda_factor", C.c_double),
                ("k_correspondences", C.c_int), ("neighbor_method", C.c_int), ("max_cells", C.c_longlong)]


class Stats(C.Structure):
    _fields_ = [("n_source", C.c_int), ("n_target", C.c_int), ("n_voxels", C.c_int), ("n_corr", C.c_int),
                ("outer_iterations", C.c_int), ("n_linearize", C.c_int), ("n_error", C.c_int),
                ("target_cells", C.c_longlong), ("source_cells", C.c_longlong),
                ("deferred_target", C.c_int), ("deferred_source", C.c_int), ("source_crowding", C.c_double),
                ("lazy_misses", C.c_int), ("searched_target", C.c_int)]


class FuseIn(C.Structure):
    _fields_ = [("q_lidar_xyzw", C.c_double * 4), ("t_lidar", C.c_double * 3), ("fitness", C.c_double), ("use_ground", C.c_int),
                ("ground_last", C.c_double * 11), ("ground_cur", C.c_double * 11), ("q_w_curr_f_xyzw", C.c_double * 4),
                ("ground_cov", C.c_double), ("use_imu", C.c_int), ("q_imu_xyzw", C.c_double * 4), ("max_iterations", C.c_int)]


class ImuFilter(C.Structure):
    _fields_ = [("ba", C.c_double * 3), ("bg", C.c_double * 3), ("dropped", C.c_int), ("count", C.c_int), ("t_last", C.c_double),
                ("roll", C.c_double), ("pitch", C.c_double), ("yaw", C.c_double), ("roll_last", C.c_double), ("pitch_last", C.c_double),
                ("Rwi", C.c_double * 9), ("mf_buf", (C.c_double * 201) * 3), ("mf_count", C.c_int * 3)]


class GroundGate(C.Structure):
    _fields_ = [("gflag", C.c_int), ("changegroundflag", C.c_int), ("q_w_curr_delta", C.c_double * 4), ("n_history", C.c_int),
                ("history", (C.c_double * 4) * 64)]


class FeParams(C.Structure):
    _fields_ = [("n_scans", C.c_int), ("min_range", C.c_double), ("max_range", C.c_double), ("use_intensity", C.c_int)]


class FeOut(C.Structure):
    _fields_ = [("cloud", C.POINTER(C.c_float)), ("cloud_cap", C.c_int), ("n_cloud", C.c_int), ("sharp", C.POINTER(C.c_float)),
                ("flat", C.POINTER(C.c_float)), ("inten", C.POINTER(C.c_float)), ("feat_cap", C.c_int), ("n_sharp", C.c_int),
                ("n_sharp_own", C.c_int), ("n_flat", C.c_int), ("n_inten", C.c_int), ("ground_pts", C.POINTER(C.c_float)),
                ("ground_cap", C.c_int), ("n_ground", C.c_int), ("groundparam", C.c_double * 11), ("ground_valid", C.c_int),
                ("ring_count", C.c_int * 64), ("curvature", C.POINTER(C.c_float)), ("curvature2", C.POINTER(C.c_float)),
                ("inten_curvature", C.POINTER(C.c_float)), ("label", C.POINTER(C.c_int)), ("inten_label", C.POINTER(C.c_int)),
                ("picked", C.POINTER(C.c_int)), ("ground_marked", C.POINTER(C.c_int))]


class MapregReport(C.Structure):
    _fields_ = [("initial_cost", C.c_double), ("final_cost", C.c_double), ("iterations", C.c_int), ("successful", C.c_int),
                ("n_edge_cur", C.c_int), ("n_plane_cur", C.c_int), ("n_edge_last", C.c_int), ("n_plane_last", C.c_int)]


class IcpParams(C.Structure):
    _fields_ = [("max_iterations", C.c_int), ("max_correspondence_distance", C.c_double), ("transformation_epsilon", C.c_double),
                ("euclidean_fitness_epsilon", C.c_double)]


class IcpResult(C.Structure):
    _fields_ = [("iterations", C.c_int), ("converged", C.c_int), ("state", C.c_int), ("n_correspondences", C.c_int), ("fitness", C.c_double)]


class Pc2Layout(C.Structure):
    _fields_ = [("point_step", C.c_int), ("offset", C.c_int * 6), ("datatype", C.c_int * 6), ("is_bigendian", C.c_int), ("strict", C.c_int)]


class Pc2Field(C.Structure):
    _fields_ = [("name", C.c_char * 16), ("offset", C.c_int), ("datatype", C.c_int), ("count", C.c_int)]


class MapregGround(C.Structure):
    _fields_ = [("last_v1", C.c_double * 3), ("last_v2", C.c_double * 3), ("last_norm", C.c_double * 3), ("last_distance", C.c_double),
                ("cur_norm", C.c_double * 3), ("cur_distance", C.c_double), ("q_history", C.c_double * 4), ("last_q", C.c_double * 4),
                ("last_t", C.c_double * 3), ("p_var", C.c_double)]


class MapregImu(C.Structure):
    _fields_ = [("delta_q", C.c_double * 4), ("imu_cov", C.c_double), ("pitch_cur", C.c_double), ("roll_cur", C.c_double),
                ("pitch_last", C.c_double), ("roll_last", C.c_double), ("pr_var", C.c_double)]


class MapInfo(C.Structure):
    _fields_ = [("n_keyframes", C.c_int), ("n_points", C.c_longlong), ("n_target", C.c_int), ("revision", C.c_ulonglong),
                ("oldest_id", C.c_int), ("newest_id", C.c_int), ("origin", C.c_double * 3)]


class RgcError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__(f"rgc_hip status {status}: {msg}")
        self.status = status


# every symbol include/rgc_hip.h declares (checked by tests/test_abi.py against the header text)
SYMBOLS = [
    "rgc_default_params", "rgc_create", "rgc_destroy", "rgc_set_params", "rgc_get_params", "rgc_last_error",
    "rgc_status_string", "rgc_version", "rgc_set_target", "rgc_set_source", "rgc_set_target_device",
    "rgc_set_source_device", "rgc_linearize", "rgc_compute_error", "rgc_num_correspondences", "rgc_align", "rgc_align_begin", "rgc_align_end", "rgc_align_end_reframe", "rgc_share_target", "rgc_hold_source_until_target_of", "rgc_set_target_lazy", "rgc_set_knn_reuse", "rgc_get_knn_reuse", "rgc_set_regularization_method", "rgc_set_voxel_accumulation_mode",
    "rgc_fitness", "rgc_get_aligned", "rgc_get_aligned_device", "rgc_get_source_covariances", "rgc_get_target_covariances", "rgc_set_source_covariances", "rgc_set_target_covariances",
    "rgc_clear_source", "rgc_clear_target", "rgc_swap_source_and_target", "rgc_get_voxels",
    "rgc_get_stats", "rgc_device_alloc", "rgc_device_free", "rgc_host_alloc", "rgc_host_free", "rgc_upload", "rgc_download", "rgc_synchronize",
    "rgc_stream", "rgc_default_fe_params", "rgc_frontend", "rgc_extract_pose", "rgc_imu_preintegrate", "rgc_imu_filter_init", "rgc_imu_filter_push", "rgc_ground_gate_init", "rgc_ground_gate_remember", "rgc_ground_gate_step", "rgc_default_fuse_in", "rgc_fuse_pose", "rgc_compose_pose",
    "rgc_R2ypr", "rgc_ypr2R", "rgc_deskew", "rgc_voxelgrid", "rgc_voxelgrid_begin", "rgc_voxelgrid_end", "rgc_transform_cloud", "rgc_set_target_reframed", "rgc_frontend_device", "rgc_frontend_cloud_device", "rgc_default_icp_params", "rgc_icp_align", "rgc_pc2_unpack", "rgc_pc2_pack", "rgc_pc2_point_fields", "rgc_tum_line", "rgc_pcd_write", "rgc_mapreg_set_maps", "rgc_mapreg_associate", "rgc_mapreg_optimize", "rgc_map_reset", "rgc_map_insert", "rgc_map_evict", "rgc_map_rebase", "rgc_map_commit", "rgc_map_get_info", "rgc_map_download", "rgc_profile_enable", "rgc_profile_select", "rgc_profile_reset", "rgc_profile_get", "rgc_profile_name",
]

_lib = None


class _PartialLibrary:
    """RGC_HIP_LIB_PARTIAL=1: LIB_PATH names a library built from a SUBSET of the sources -- csrc/rgc_host.cpp alone under
    -fsanitize=address,undefined (tests/test_sanitizers.py: the scalar host stages need no HIP).  A symbol it lacks becomes a stand-in
    that takes the prototype declarations below and raises when called; the product library is never loaded this way."""

    class _Missing:
        def __init__(self, name):
            object.__setattr__(self, "_name", name)

        def __setattr__(self, k, v):
            pass

        def __call__(self, *a, **kw):
            raise RuntimeError(f"{self._name} is not in the partial library {LIB_PATH}")

    def __init__(self, real):
        self._real = real
        self._missing = {}

    def __getattr__(self, name):
        try:
            return getattr(self._real, name)
        except AttributeError:
            return self._missing.setdefault(name, _PartialLibrary._Missing(name))


_seq = None


def load_seq():
    """librgc_seq.so: the C++ host layer's dependent frame loop behind one call (cpp/dependent_sequence_c.cpp).  It is linked against the
    in-tree librgc_hip.so; with RGC_HIP_LIB naming another build there is no frame loop for it (None): contexts of one library build
    must not be driven through another's code."""
    global _seq
    if _seq is not None:
        return _seq
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "librgc_seq.so")
    if os.path.abspath(LIB_PATH) != os.path.join(os.path.dirname(os.path.abspath(__file__)), "librgc_hip.so") or not os.path.exists(path):
        return None
    load()
    S = C.CDLL(path)
    vp, fp, dp, ip = C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_double), C.POINTER(C.c_int)
    S.rgc_seq_run_dependent.argtypes = [vp, vp, vp, C.c_int, C.c_int, vp, vp, C.POINTER(vp), ip, C.c_int, C.c_int, dp, fp, C.c_int, fp, dp, dp, ip, dp]
    S.rgc_seq_run_dependent.restype = C.c_int
    _seq = S
    return S


def load():
    """Load librgc_hip.so and declare prototypes.  Raises if the library is missing (build with
    `python rgc-slam_amd/build.py` or __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} not found: the HIP extension is required (no CPU fallback); "
                          f"run `python rgc-slam_amd/build.py`")
    L = C.CDLL(LIB_PATH)
    if os.environ.get("RGC_HIP_LIB_PARTIAL") == "1":
        L = _PartialLibrary(L)
    vp, fp, dp, ip = C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_double), C.POINTER(C.c_int)
    L.rgc_default_params.argtypes = [C.POINTER(Params)]
    L.rgc_default_params.restype = None
    L.rgc_create.argtypes = [C.c_int, C.POINTER(Params), C.POINTER(vp)]
    L.rgc_destroy.argtypes = [vp]
    L.rgc_destroy.restype = None
    L.rgc_set_params.argtypes = [vp, C.POINTER(Params)]
    L.rgc_get_params.argtypes = [vp, C.POINTER(Params)]
    L.rgc_last_error.argtypes = [vp]
    L.rgc_last_error.restype = C.c_char_p
    L.rgc_status_string.argtypes = [C.c_int]
    L.rgc_status_string.restype = C.c_char_p
    L.rgc_version.restype = C.c_char_p
    for f in (L.rgc_set_target, L.rgc_set_source, L.rgc_set_target_device, L.rgc_set_source_device):
        f.argtypes = [vp, vp, C.c_int, C.c_int]
    L.rgc_linearize.argtypes = [vp, dp, dp, dp, dp]
    L.rgc_compute_error.argtypes = [vp, dp, dp]
    L.rgc_num_correspondences.argtypes = [vp, ip]
    L.rgc_align.argtypes = [vp, fp, fp, dp, dp, ip, ip, ip]
    L.rgc_align_begin.argtypes = [vp, fp, C.c_int]
    L.rgc_align_end.argtypes = [vp, fp, dp, dp, ip, ip, ip]
    L.rgc_share_target.argtypes = [vp, vp]
    L.rgc_fitness.argtypes = [vp, fp, dp]
    L.rgc_get_aligned.argtypes = [vp, fp, fp, C.c_int]
    L.rgc_get_aligned_device.argtypes = [vp, fp, vp, C.c_int]
    L.rgc_get_source_covariances.argtypes = [vp, dp, dp]
    L.rgc_set_source_covariances.argtypes = [vp, dp, C.c_int]
    L.rgc_set_target_covariances.argtypes = [vp, dp, C.c_int]
    L.rgc_hold_source_until_target_of.argtypes = [vp, vp]
    L.rgc_set_target_lazy.argtypes = [vp, C.c_int]
    L.rgc_set_knn_reuse.argtypes = [vp, C.c_int]
    L.rgc_set_regularization_method.argtypes = [vp, C.c_int]
    L.rgc_set_voxel_accumulation_mode.argtypes = [vp, C.c_int]
    L.rgc_get_knn_reuse.argtypes = [vp, ip]
    L.rgc_clear_source.argtypes = [vp]
    L.rgc_clear_target.argtypes = [vp]
    L.rgc_swap_source_and_target.argtypes = [vp]
    L.rgc_get_target_covariances.argtypes = [vp, dp, dp]
    L.rgc_get_voxels.argtypes = [vp, C.c_int, ip, ip, dp, dp, ip]
    L.rgc_get_stats.argtypes = [vp, C.POINTER(Stats)]
    L.rgc_align_end_reframe.argtypes = [vp, vp, dp, vp, C.c_int, C.c_int, vp, C.POINTER(C.c_float), dp, dp, ip, ip, ip]
    L.rgc_device_alloc.argtypes = [vp, C.c_size_t, C.POINTER(vp)]
    L.rgc_device_free.argtypes = [vp, vp]
    L.rgc_host_alloc.argtypes = [C.c_size_t, C.POINTER(C.c_void_p)]
    L.rgc_host_free.argtypes = [vp]
    L.rgc_upload.argtypes = [vp, vp, vp, C.c_size_t]
    L.rgc_download.argtypes = [vp, vp, vp, C.c_size_t]
    L.rgc_synchronize.argtypes = [vp]
    L.rgc_stream.argtypes = [vp]
    L.rgc_stream.restype = vp
    L.rgc_default_fe_params.argtypes = [C.POINTER(FeParams)]
    L.rgc_default_fe_params.restype = None
    L.rgc_frontend.argtypes = [vp, vp, C.c_int, C.c_int, C.POINTER(FeParams), C.POINTER(FeOut)]
    L.rgc_frontend_device.argtypes = [vp, vp, C.c_int, C.c_int, C.POINTER(FeParams), C.POINTER(FeOut)]
    L.rgc_extract_pose.argtypes = [fp, dp, dp]
    L.rgc_imu_preintegrate.argtypes = [dp, dp, dp, C.c_int, C.c_double, C.c_double, dp, dp, dp, dp]
    L.rgc_imu_filter_init.argtypes = [C.POINTER(ImuFilter)]
    L.rgc_imu_filter_init.restype = None
    L.rgc_imu_filter_push.argtypes = [C.POINTER(ImuFilter), C.c_double, dp, dp, dp, dp]
    L.rgc_ground_gate_init.argtypes = [C.POINTER(GroundGate)]
    L.rgc_ground_gate_init.restype = None
    L.rgc_ground_gate_remember.argtypes = [C.POINTER(GroundGate)]
    L.rgc_ground_gate_remember.restype = None
    L.rgc_ground_gate_step.argtypes = [C.POINTER(GroundGate), dp, dp, dp, dp, dp, dp, dp]
    L.rgc_default_fuse_in.argtypes = [C.POINTER(FuseIn)]
    L.rgc_default_fuse_in.restype = None
    L.rgc_fuse_pose.argtypes = [C.POINTER(FuseIn), dp, dp, ip]
    L.rgc_compose_pose.argtypes = [dp, dp, dp, dp, dp, C.c_int, dp, dp, dp, dp]
    L.rgc_R2ypr.argtypes = [dp, dp]
    L.rgc_R2ypr.restype = None
    L.rgc_ypr2R.argtypes = [dp, dp]
    L.rgc_ypr2R.restype = None
    L.rgc_deskew.argtypes = [vp, vp, C.c_int, C.c_int, dp, dp, C.c_int]
    L.rgc_voxelgrid.argtypes = [vp, vp, C.c_int, C.c_int, C.c_float, vp, ip, C.c_int]
    L.rgc_voxelgrid_begin.argtypes = [vp, vp, C.c_int, C.c_int, C.c_float, vp]
    L.rgc_voxelgrid_end.argtypes = [vp, ip]
    L.rgc_transform_cloud.argtypes = [vp, vp, C.c_int, C.c_int, dp, dp, vp, C.c_int]
    L.rgc_set_target_reframed.argtypes = [vp, vp, C.c_int, C.c_int, dp, dp, vp]
    L.rgc_default_icp_params.argtypes = [C.POINTER(IcpParams)]
    L.rgc_default_icp_params.restype = None
    L.rgc_icp_align.argtypes = [vp, fp, C.c_int, fp, C.c_int, C.c_int, C.POINTER(IcpParams), fp, C.POINTER(IcpResult)]
    L.rgc_pc2_unpack.argtypes = [vp, vp, C.c_int, C.POINTER(Pc2Layout), vp, vp, vp, C.c_int]
    L.rgc_pc2_pack.argtypes = [vp, C.c_int, vp, C.c_int, C.c_int, vp]
    L.rgc_pc2_point_fields.argtypes = [C.c_int, C.POINTER(Pc2Field), C.c_int, ip]
    L.rgc_tum_line.argtypes = [C.c_double, dp, dp, C.c_char_p, C.c_int]
    L.rgc_pcd_write.argtypes = [C.c_char_p, fp, C.c_int, C.c_int]
    L.rgc_mapreg_set_maps.argtypes = [vp, fp, C.c_int, fp, C.c_int, C.c_int]
    L.rgc_mapreg_associate.argtypes = [vp, C.c_int, fp, C.c_int, dp, dp, dp, ip]
    L.rgc_mapreg_optimize.argtypes = [vp, fp, C.c_int, fp, C.c_int, fp, C.c_int, fp, C.c_int, C.POINTER(MapregGround), C.POINTER(MapregGround),
                                      C.POINTER(MapregImu), dp, C.POINTER(MapregReport), ip]
    L.rgc_frontend_cloud_device.argtypes = [vp, C.POINTER(C.c_void_p), ip]
    L.rgc_map_reset.argtypes = [vp, dp]
    L.rgc_map_insert.argtypes = [vp, vp, C.c_int, C.c_int, dp, dp, C.c_int, ip]
    L.rgc_map_evict.argtypes = [vp, C.c_int, dp, C.c_double, ip]
    L.rgc_map_rebase.argtypes = [vp, dp]
    L.rgc_map_commit.argtypes = [vp, C.c_float, ip]
    L.rgc_map_get_info.argtypes = [vp, C.POINTER(MapInfo)]
    L.rgc_map_download.argtypes = [vp, C.c_int, vp, C.c_int, ip]
    L.rgc_profile_enable.argtypes = [vp, C.c_int]
    L.rgc_profile_select.argtypes = [vp, C.c_uint]
    L.rgc_profile_reset.argtypes = [vp]
    L.rgc_profile_get.argtypes = [vp, C.c_int, C.POINTER(C.c_longlong), dp, C.POINTER(C.c_longlong)]
    L.rgc_profile_name.argtypes = [C.c_int]
    L.rgc_profile_name.restype = C.c_char_p
    _lib = L
    return L


def default_params(**kw) -> Params:
    p = Params()
    load().rgc_default_params(C.byref(p))
    for k, v in kw.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, v)
    return p
