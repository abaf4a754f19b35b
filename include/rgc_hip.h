/*
 * rgc_hip.h -- C-ABI of librgc_hip.so: the MI355X (gfx950) scan-to-map registration path of RGC-SLAM.
 *
 * Plain C: opaque context, plain pointers and sizes, int status codes.  No C++/PCL/Eigen/ROS/torch types.
 * Every entry point names the reference interface it replaces; paths are relative to
 * /root/reference/rgc_slam/ (ROBOT-WSC/RGC-SLAM @2024_10_08).
 *
 * Ownership: the caller owns every host buffer for the duration of the call only (inputs are copied to
 *   the device before the call returns); the context owns all device memory, its HIP stream and events.
 * Threading: a context is NOT thread-safe and is bound to one HIP device; use one context per
 *   sequence / GPU, driven by one host thread (the reference drives registration from the single
 *   ICP_thread, src/RGC_odometer.cpp:408).  Every entry point that reaches the HIP runtime selects the context's
 *   device first (hipSetDevice) and leaves it current on return: a process may hold contexts on several GPUs and
 *   call each from a thread of its own, whatever device that thread had current (tests/test_abi.py audits it).
 * Errors: 0 = OK, negative = rgc_status; nothing aborts or throws across this boundary (the reference
 *   prints "lm not converged!!" and carries on, lsq_registration_impl.hpp:69-72; here that is the
 *   `lm_failed` output).  rgc_last_error() returns a human-readable message for the last failure.
 * There is NO CPU fallback: if no HIP device / kernel image is available every call fails with
 *   RGC_ERR_HIP.
 *
 * Matrices: 4x4 poses and 6x6 Hessians are ROW-MAJOR.  The 6-vector order is [rotation(3), translation(3)]
 *   with a LEFT-multiplied perturbation, exactly as LsqRegistration (lsq_registration_impl.hpp:139-143).
 */
#ifndef RGC_HIP_H
#define RGC_HIP_H

#include <stddef.h>

#if defined(__GNUC__) || defined(__clang__)
#define RGC_API __attribute__((visibility("default")))
#else
#define RGC_API
#endif

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rgc_ctx rgc_ctx;

typedef enum rgc_status {
  RGC_OK = 0,
  RGC_ERR_INVALID = -1,         /* bad argument / call order                                   */
  RGC_ERR_HIP = -2,             /* HIP runtime error (no device, OOM, launch failure, ...)     */
  RGC_ERR_TOO_FEW_POINTS = -3,  /* cloud has fewer than k_correspondences points (undefined in
                                   the reference, fast_gicp_impl.hpp:256-259)      */
  RGC_ERR_GRID_TOO_LARGE = -4,  /* bounding box / resolution needs more cells than max_cells   */
  RGC_ERR_NO_INPUT = -5,        /* source or target not set                                    */
  RGC_ERR_NONFINITE = -6,       /* input contains NaN/Inf coordinates                          */
  RGC_ERR_UNSUPPORTED = -7      /* the request has no meaning under the selected settings (e.g.
                                   unit normals asked for under a RegularizationMethod other
                                   than PLANE: rgc_set_regularization_method)                  */
} rgc_status;
/* What every entry point checks of its arguments before it touches anything (tests/fuzz/fuzz_null_args.py, fuzz_bad_args.py call each of them
 * with nothing, and with one bad argument at a time): a required pointer that is NULL, a negative count, a count above 2^27 points (2^24
 * for a PointCloud2 message), a stride that is not a multiple of 4 between the point type's minimum and 4096 bytes, a leaf size or a pose
 * that is not finite -- RGC_ERR_INVALID / RGC_ERR_NONFINITE, and the context is as it was.  What it cannot check and takes the caller's
 * word for: that a buffer holds as many points as the count says, and that a pointer called "device" (on_device != 0, the *_device
 * entries) is one -- unless RGC_CHECK_POINTERS=1 is in the environment when the context is created: then every such pointer is looked up first
 * (device memory of the context's GPU, an allocation with room for the count) and a host pointer or a short buffer is RGC_ERR_INVALID.
 * A HIP error is reported once, by the call it happened in (RGC_ERR_HIP, rgc_last_error); it does not surface again. */

/* enum order = fast_gicp::NeighborSearchMethod, include/fast_gicp/gicp/gicp_settings.hpp:8 */
typedef enum rgc_neighbor_method { RGC_DIRECT27 = 0, RGC_DIRECT7 = 1, RGC_DIRECT1 = 2 } rgc_neighbor_method;

/* Parameter block = the setters the odometer calls on fast_gicp::FastVGICP (src/RGC_odometer.cpp:998-1006)
 * plus the constructor defaults they leave untouched.  Defaults: rgc_default_params(). */
typedef struct rgc_params {
  double voxel_res;             /* setResolution(1.0)            RGC_odometer.cpp:308,1000; fast_vgicp_impl.hpp:32-34 */
  int    max_iterations;        /* setMaximumIterations(25)      RGC_odometer.cpp:1001                                */
  int    lm_max_iterations;     /* lm_max_iterations_ = 10       lsq_registration_impl.hpp:17                          */
  double rotation_eps;          /* setRotationEpsilon (2e-3)     lsq_registration_impl.hpp:12,27-29                    */
  double translation_eps;       /* setTransformationEpsilon(1e-6) RGC_odometer.cpp:1003                               */
  double lm_init_lambda_factor; /* setInitialLambdaFactor (1e-9) lsq_registration_impl.hpp:18,32-34                    */
  int    k_correspondences;     /* setCorrespondenceRandomness (20) fast_gicp_impl.hpp:16,41-43; 2..32 supported       */
  int    neighbor_method;       /* setNeighborSearchMethod (RGC_DIRECT1) fast_vgicp_impl.hpp:23,37-39                  */
  long long max_cells;          /* cap on dense grid cells per cloud (default 1<<29)                                  */
} rgc_params;

RGC_API void rgc_default_params(rgc_params* p);

/* FastGICP::setRegularizationMethod (include/fast_gicp/gicp/fast_gicp.hpp:52, impl/fast_gicp_impl.hpp:46-48; enum order =
 * fast_gicp::RegularizationMethod, gicp_settings.hpp:6) and FastVGICP::setVoxelAccumulationMode (fast_vgicp.hpp:57,
 * impl/fast_vgicp_impl.hpp:41-43; enum order = VoxelAccumulationMode, gicp_settings.hpp:10).  The odometer calls neither
 * (src/RGC_odometer.cpp:998-1006 leaves PLANE, fast_gicp_impl.hpp:20, and ADDITIVE, fast_vgicp_impl.hpp:24).
 * Every value is implemented.  PLANE with ADDITIVE / ADDITIVE_WEIGHTED (one and the same AdditiveGaussianVoxel in the vendored FastVGICP,
 * fast_vgicp_voxel.hpp:137-141; the mode is looked at nowhere else) runs on the tuned kernels, which keep a point's covariance as the unit
 * normal of I - 0.999 n n^T (24 bytes).  NONE, MIN_EIG, NORMALIZED_MIN_EIG, FROBENIUS (fast_gicp_impl.hpp:262-293) and
 * VoxelAccumulationMode::MULTIPLICATIVE (fast_vgicp_voxel.hpp:76-99) run on a GENERAL route: a regularised 3x3 per point (48 bytes), every
 * point's exact k-NN through the cooperative search, a plain voxel pass, and the LM loop driven from the host over a linearisation that takes
 * the full source covariance -- the same entry points and the reference's arithmetic, about four times slower at c1 size (it is not the odometer's
 * path and has not been tuned).  On that route rgc_align_begin solves at once and rgc_align_end hands the result over; rgc_set_target_lazy and
 * rgc_set_knn_reuse have no effect; rgc_get_*_covariances returns no normals (RGC_ERR_UNSUPPORTED if asked); rgc_set_*_covariances takes any
 * symmetric 3x3.  RGC_ERR_UNSUPPORTED is otherwise unused by these calls; an out-of-range value is RGC_ERR_INVALID.
 * The reference computes covariances at align() under the method selected THEN (fast_gicp_impl.hpp:103-112); this library computes them when
 * a cloud is set, so changing the method (or additive <-> multiplicative) afterwards DROPS the clouds: set them again.  Select before setting. */
typedef enum rgc_regularization_method { RGC_REG_NONE = 0, RGC_REG_MIN_EIG = 1, RGC_REG_NORMALIZED_MIN_EIG = 2, RGC_REG_PLANE = 3, RGC_REG_FROBENIUS = 4 } rgc_regularization_method;
typedef enum rgc_voxel_accumulation_mode { RGC_VOXEL_ADDITIVE = 0, RGC_VOXEL_ADDITIVE_WEIGHTED = 1, RGC_VOXEL_MULTIPLICATIVE = 2 } rgc_voxel_accumulation_mode;
RGC_API int rgc_set_regularization_method(rgc_ctx* ctx, int method);
RGC_API int rgc_set_voxel_accumulation_mode(rgc_ctx* ctx, int mode);

/* FastVGICP construction / destruction (a stack local re-created per frame at RGC_odometer.cpp:998;
 * here the context is long-lived and re-used, device buffers grow on demand). */
RGC_API int  rgc_create(int hip_device, const rgc_params* params /* NULL = defaults */, rgc_ctx** out);
RGC_API void rgc_destroy(rgc_ctx* ctx);
/* setResolution / setCorrespondenceRandomness / setMaximumIterations ... in one struct.  A new voxel_res or k_correspondences prepares the
 * clouds the context holds again, from their inputs (a device cloud must still be where it was set from); refused with a solve in flight. */
RGC_API int  rgc_set_params(rgc_ctx* ctx, const rgc_params* params);
RGC_API int  rgc_get_params(const rgc_ctx* ctx, rgc_params* params);
RGC_API const char* rgc_last_error(const rgc_ctx* ctx);
RGC_API const char* rgc_status_string(int status);
RGC_API const char* rgc_version(void);

/* setInputTarget (fast_vgicp_impl.hpp:56-63) / setInputSource (fast_gicp_impl.hpp:72-80).
 * xyz: first float of point 0; point i starts at (char*)xyz + i*stride_bytes (x,y,z consecutive floats;
 * stride 16 or 32 = pcl::PointXYZI padding welcome).  Setting a cloud drops its covariances (and, for the
 * target, the voxel map) exactly like the reference; the exact-kNN covariances (fast_gicp_impl.hpp:241-298),
 * and for the target the Gaussian voxel map (fast_vgicp_voxel.hpp:129-156), are computed on the device
 * (enqueued immediately on the context's stream).  n <= 2^27 points per cloud (RGC_ERR_INVALID beyond: the kernels
 * address the sorted 16-byte points with 32-bit byte offsets).
 * From its second cloud on a context re-uses the previous cloud's (widened) grid instead of measuring the bounding box first
 * -- the one host round trip of these calls; results are unaffected.  A cloud that does not fit is prepared again, transparently,
 * when the first call that consumes it finds out; by the same token RGC_ERR_NONFINITE / RGC_ERR_GRID_TOO_LARGE for such a cloud
 * may be returned by that consuming call (rgc_align, rgc_linearize, a getter ...) instead of by rgc_set_*.  RGC_SPEC_GRID=0 in
 * the environment restores the immediate check. */
RGC_API int rgc_set_target(rgc_ctx* ctx, const float* xyz, int n, int stride_bytes);
RGC_API int rgc_set_source(rgc_ctx* ctx, const float* xyz, int n, int stride_bytes);
/* same, but xyz is DEVICE memory on the context's device (cloud already resident in HBM).
 * Streams: the target is prepared on rgc_stream(ctx); the SOURCE is prepared on a second, internal stream so that it overlaps the
 * (much larger) target's preparation, and rgc_align joins the two.  That internal stream is ordered after everything that was
 * enqueued on rgc_stream(ctx) before this frame's rgc_set_target* call (or before rgc_set_source* itself when no target preparation
 * is in flight): a source buffer written by rgc_upload, by the front-end or by the caller's own kernels on rgc_stream(ctx) is safe
 * to hand over without synchronising, provided those writes were enqueued before rgc_set_target*.  Writes enqueued between
 * rgc_set_target* and rgc_set_source_device must be complete (synchronised) first. */
RGC_API int rgc_set_target_device(rgc_ctx* ctx, const float* d_xyz, int n, int stride_bytes);
RGC_API int rgc_set_source_device(rgc_ctx* ctx, const float* d_xyz, int n, int stride_bytes);

/* LsqRegistration::linearize (lsq_registration.hpp:68; fast_vgicp_impl.hpp:119-180): rebuilds the voxel
 * correspondences and Mahalanobis matrices at T, returns cost and (if both non-NULL) H, b. */
RGC_API int rgc_linearize(rgc_ctx* ctx, const double T[16], double H[36], double b[6], double* cost);
/* LsqRegistration::compute_error (lsq_registration.hpp:69; fast_vgicp_impl.hpp:183-204): cost at T with the
 * correspondences and Mahalanobis matrices FROZEN at the last rgc_linearize. */
RGC_API int rgc_compute_error(rgc_ctx* ctx, const double T[16], double* cost);
RGC_API int rgc_num_correspondences(rgc_ctx* ctx, int* n_corr);

/* pcl::Registration::align(out, guess) -> LsqRegistration::computeTransformation
 * (lsq_registration_impl.hpp:53-79, LM step :125-172) + getFinalTransformation + getFinalHessian (:43-45)
 * + getFitnessScore (RGC_odometer.cpp:1009-1011).  Any output pointer may be NULL.
 *   final_T    : float 4x4 (final_transformation_ = x0.cast<float>(), :77)
 *   final_H    : 6x6 of the last accepted LM step (identity if none, :21,167)
 *   fitness    : mean squared 1-NN distance source->target (computed only if non-NULL)
 *   iterations : outer iterations executed;  converged: hasConverged();  lm_failed: "lm not converged!!" */
RGC_API int rgc_align(rgc_ctx* ctx, const float guess[16], float final_T[16], double final_H[36], double* fitness,
              int* iterations, int* converged, int* lm_failed);
/* rgc_align in two halves (rgc_align == begin + end).  rgc_align_begin enqueues the solve behind the clouds' preparation and
 * returns without waiting; rgc_align_end waits for it and returns what rgc_align returns.  Between the two the caller may do
 * anything that does not touch THIS context -- in particular prepare the next frame's clouds on a second context, so that
 * the preparation of frame i + 1 runs on the GPU while frame i is being solved (rgc_slam_amd.registration.PipelinedVGICP,
 * rgc::OdometryNode's replay).  want_fitness != 0 chains getFitnessScore behind the solve (rgc_align_end's `fitness` then costs
 * nothing extra).  Errors that a speculative grid defers (rgc_set_* above) may surface in rgc_align_end.
 * BETWEEN the two halves the context refuses (RGC_ERR_INVALID, "a solve is in flight") everything that would touch the solve's inputs:
 * new clouds, clearing or swapping them, another begin or a blocking rgc_align, the settings that re-route a cloud, the getters,
 * rgc_map_commit (but for the no-op commit of an unchanged map).  The same on the general covariance route, where rgc_align_begin runs the
 * solve at once and keeps its result for rgc_align_end. */
RGC_API int rgc_align_begin(rgc_ctx* ctx, const float guess[16], int want_fitness);
RGC_API int rgc_align_end(rgc_ctx* ctx, float final_T[16], double final_H[36], double* fitness, int* iterations, int* converged,
                          int* lm_failed);
/* rgc_align_end on `solve`, then -- without going back to the caller -- the two steps a frame loop WITHOUT a fusion stage does with the
 * result: the world pose composed in fp64, world_T <- world_T * final_T (t_w_curr / q_w_curr, src/RGC_odometer.cpp:1201-1203), and the
 * next frame's target enqueued on `next` (the same context or the other one of a pair taking turns): the sub-map d_map re-expressed in
 * the new body frame, q = inverse rotation of world_T, t = -q * translation (:1250-1255), exactly rgc_set_target_reframed(next, d_map,
 * n, stride_bytes, q, t, d_scratch).  The host's turn-around between a frame's result and the next frame's first launch is on the
 * critical path of a dependent sequence; here it is a few microseconds of C instead of the caller's pose arithmetic and a second call.
 * world_T: 16 doubles, row-major 4x4, in: the world pose before this frame, out: after it.  Outputs as rgc_align_end.
 * On two contexts (next != solve) with the fitness chained to the solve, the next target is enqueued as soon as the solve's final POSE is
 * known -- the deciding launch posts it before it computes the score -- and the call then waits for the score: same results, the score's
 * ~25 us no longer in front of the next frame.
 * Failure is all or nothing for the caller's arguments: `next` (alive, no solve in flight), d_map / n / stride_bytes / d_scratch are
 * checked as rgc_set_target_reframed checks them BEFORE the solve is consumed -- such an error leaves the solve pending and world_T
 * untouched, and the call may be repeated with corrected arguments.  (A HIP or allocation failure inside the next target's preparation
 * is reported after the outputs and world_T have been written; rgc_last_error names the failing call.) */
RGC_API int rgc_align_end_reframe(rgc_ctx* solve, rgc_ctx* next, double world_T[16], const float* d_map, int n, int stride_bytes,
                                  float* d_scratch, float final_T[16], double final_H[36], double* fitness, int* iterations,
                                  int* converged, int* lm_failed);
/* ctx registers its scans to the target `owner` has prepared (rgc_set_target*, or rgc_map_commit: the resident local map), without
 * preparing or copying it: ctx's target becomes a non-owning alias of the owner's device buffers.  Two contexts can then take turns
 * on a sequence whose map does not change every frame (the next scan is prepared on one while the current one is solved on the
 * other).  The owner must stay alive and must not be destroyed while ctx uses the target; when the owner prepares a new target
 * (rgc_set_target*, a commit that rebuilds) ctx's next rgc_align* fails with RGC_ERR_INVALID until the target is shared again.
 * Synchronises both contexts.  No reference counterpart (the reference builds one FastVGICP per frame).
 * A borrowed target is the OWNER's, prepared under the owner's settings: rgc_set_params on ctx with another voxel_res or k (which prepares a
 * context's own clouds again) drops the alias -- share again once the owner holds a target under those settings; swapping, clearing or
 * setting covariances on a borrowed target is refused or drops it likewise. */
RGC_API int rgc_share_target(rgc_ctx* ctx, rgc_ctx* owner);
/* pcl::Registration::getFitnessScore() for an arbitrary pose */
RGC_API int rgc_fitness(rgc_ctx* ctx, const float T[16], double* fitness);
/* the `output` cloud of align(): pcl::transformPointCloud(*input_, output, final_transformation_)
 * (lsq_registration_impl.hpp:78) in the caller's point order; out stride in bytes (>= 12). */
RGC_API int rgc_get_aligned(rgc_ctx* ctx, const float T[16], float* out_xyz, int stride_bytes);
/* the same cloud left on the device: d_out_xyz is a device pointer (n_source * stride_bytes bytes); enqueued on the context's
 * stream, no synchronisation, no copy -- for callers that consume the aligned cloud on the GPU (the odometer's sub-map insert) */
RGC_API int rgc_get_aligned_device(rgc_ctx* ctx, const float T[16], float* d_out_xyz, int stride_bytes);

/* Lazy target.  Only the voxels the solve LOOKS UP enter its cost (update_correspondences, impl/fast_vgicp_impl.hpp:73-116: one voxel per
 * source point with DIRECT1), and a voxel's covariance needs the 20-NN covariances of that voxel's points only -- yet the reference builds
 * every covariance of the map every frame (setInputTarget, src/RGC_odometer.cpp:1007), and so does this library by default.  With
 * margin_cells > 0 a target set afterwards gets its grid at rgc_set_target* and the rest at rgc_align / rgc_align_begin, for the cells
 * within margin_cells voxels (Chebyshev) of a voxel the scan falls into at the guess: 8 % of the c-main map at a margin of 2.  Every
 * look-up of the solve is checked; one that lands on an occupied voxel outside the built part (the pose moved more than the margin
 * covers) makes rgc_align_end complete the map and solve again, and any other consumer of the target (getters, the fine seam,
 * rgc_share_target, a second solve on the same target) completes it first: results are those of the full build, bit for bit, always.
 * 0 (default): off.  rgc_stats::lazy_misses counts the solves that had to be repeated. */
RGC_API int rgc_set_target_lazy(rgc_ctx* ctx, int margin_cells);

/* Scheduling hint for two contexts taking turns on a dependent sequence -- every frame's target is the sub-map re-framed by the previous
 * pose (src/RGC_odometer.cpp:1248-1256), so only the NEXT scan can be prepared ahead: the next rgc_set_source* on ctx starts on the
 * GPU only when the target preparation other has enqueued last (rgc_set_target*) has finished, i.e. under other's solve instead of
 * beside its map's kNN launch.  Results are unaffected. */
RGC_API int rgc_hold_source_until_target_of(rgc_ctx* ctx, rgc_ctx* other);

/* getSourceCovariances / getTargetCovariances analogue (fast_gicp.hpp): PLANE-regularised 3x3 covariances
 * (row-major, n*9 doubles) and/or unit normals (n*3 doubles, sign arbitrary), caller point order. */
RGC_API int rgc_get_source_covariances(rgc_ctx* ctx, double* cov9 /* may be NULL */, double* normals /* may be NULL */);
RGC_API int rgc_get_target_covariances(rgc_ctx* ctx, double* cov9, double* normals);
/* FastGICP::setSourceCovariances / setTargetCovariances (include/fast_gicp/gicp/fast_gicp.hpp:58-61, impl/fast_gicp_impl.hpp:93-100):
 * n*9 doubles, row-major, caller point order, for the cloud set before.  Only covariances of the plane-regularised form
 * I - 0.999 n n^T (what this path and the reference's odometer produce, fast_gicp_impl.hpp:280-293) are representable here: any other
 * matrix is RGC_ERR_INVALID.  The target's voxel map is rebuilt from them. */
RGC_API int rgc_set_source_covariances(rgc_ctx* ctx, const double* cov9, int n);
RGC_API int rgc_set_target_covariances(rgc_ctx* ctx, const double* cov9, int n);
/* FastGICP::clearSource / clearTarget (fast_gicp.hpp:56-57, fast_gicp_impl.hpp:60-69) and FastVGICP::swapSourceAndTarget
 * (fast_vgicp_impl.hpp:46-53: the clouds change roles, the voxel map is rebuilt from the new target; covariances the caller set with
 * rgc_set_source/target_covariances travel with their cloud, like the reference's swap of source_covs_ / target_covs_ -- computed
 * ones are the same function of the cloud in either role). */
RGC_API int rgc_clear_source(rgc_ctx* ctx);
RGC_API int rgc_clear_target(rgc_ctx* ctx);
RGC_API int rgc_swap_source_and_target(rgc_ctx* ctx);
/* Gaussian voxel map dump (fast_vgicp_voxel.hpp:105-122): up to cap voxels, unordered.
 * coords 3*cap ints, num cap ints, mean 3*cap doubles, cov9 9*cap doubles; *count = total voxels. */
RGC_API int rgc_get_voxels(rgc_ctx* ctx, int cap, int* coords, int* num, double* mean, double* cov9, int* count);

/* ---- stages either side of the operator in the odometer's per-frame body (vg_ICP::ICP_thread) ----
 * on_device != 0: every cloud pointer of the call is device memory on the context's device. */
/* B2  vg_ICP::adjustDistortion (src/RGC_odometer.cpp:1441-1481), in place.  Points are x,y,z,intensity with
 * intensity = ring + 0.1 * relTime as the front-end encodes it (scanRegistration.cpp:207-210).
 * q_last_curr_xyzw / t_last_curr: the motion guess (IMU pre-integration or previous delta, :929,993-996).
 * on_device: returns once the kernel is enqueued on rgc_stream(ctx); later calls on this context, rgc_download and work the caller
 * enqueues on that stream see the de-skewed sweep (the frame body's next stage is the leaf filter, which reads it there). */
RGC_API int rgc_deskew(rgc_ctx* ctx, float* xyzi, int n, int stride_bytes, const double q_last_curr_xyzw[4],
                       const double t_last_curr[3], int on_device);
/* B3  pcl::VoxelGrid<PointXYZI>::filter with setLeafSize(leaf,leaf,leaf) (src/RGC_odometer.cpp:976-991).
 * out_xyzi: n*4 floats capacity (x,y,z,intensity centroids, 16-byte stride, ordered by leaf index); *n_out = leaves.
 * If the leaf grid would overflow int the input is returned unfiltered, like PCL.
 * (A context remembers the leaf box of the last cloud per leaf size and filters the next one on it without measuring its bounding
 * box first; the output does not depend on the box, and a cloud that leaves it is filtered again on its own.) */
RGC_API int rgc_voxelgrid(rgc_ctx* ctx, const float* xyzi, int n, int stride_bytes, float leaf, float* out_xyzi, int* n_out,
                          int on_device);
/* The same filter for a DEVICE cloud in two halves: begin enqueues it (on the leaf box kept from the previous cloud of this leaf size) and
 * returns, end waits and returns the point count -- repeating the filter when the kept box did not hold the cloud, so d_xyzi and d_out must
 * stay untouched in between.  The odometer's sub-map filter (src/RGC_odometer.cpp:985-991) depends only on the previous frame's pose: begun
 * when that frame ends and ended after the next sweep's own filter, it is off the frame's critical path.  Other rgc_voxelgrid calls may run
 * in between; one begin may be open per context. */
RGC_API int rgc_voxelgrid_begin(rgc_ctx* ctx, const float* d_xyzi, int n, int stride_bytes, float leaf, float* d_out);
RGC_API int rgc_voxelgrid_end(rgc_ctx* ctx, int* n_out);
/* B9  vg_ICP::transformPointCloud(cloud, q, t) (src/RGC_odometer.cpp:1495-1514): q * p + t in fp64, stored fp32,
 * intensity copied; out_xyzi: n*4 floats.  on_device: both pointers are device memory and the call returns once the kernel is enqueued
 * on rgc_stream(ctx), like rgc_deskew. */
RGC_API int rgc_transform_cloud(rgc_ctx* ctx, const float* xyzi, int n, int stride_bytes, const double q_xyzw[4], const double t[3],
                                float* out_xyzi, int on_device);

/* B9 + setInputTarget in one call, for a sub-map that lives on the device (src/RGC_odometer.cpp:1248-1256 then :998-1007): d_xyzi (n
 * points, fixed between calls) re-expressed by q * p + t into d_scratch (n*4 floats, device) and prepared as the registration's target
 * -- grid, exact-kNN covariances, voxel map, like rgc_set_target_device -- without a host round trip: the re-framed cloud's bounding
 * box is derived from the input's (measured on the first call with a given d_xyzi, n) and the transform.  What a dependent sequence
 * does every frame with the pose the previous frame returned.
 * Seeds (round 5): while consecutive calls name the same (d_xyzi, n), the map's exact 20-NN search (fast_gicp_impl.hpp:241-298) starts
 * from what the last search of each point found -- its k-th neighbour distance, kept per ORIGINAL point (4 bytes each) and widened by the
 * fp32 rounding of the coordinates in two frames: only candidates under that bound are looked at, no running top-k is kept.  The result
 * does NOT depend on what the seeds hold: a search that does not find exactly its k neighbours under the bound (a buffer rewritten in
 * place, a different cloud at the same address) is repeated without it, so "fixed between calls" is what makes the call fast, not what
 * makes it right.  Covariances are bit-identical with and without seeds (RGC_KNN_SEEDS=0 in the environment switches them off).
 * Neighbour lists (round 5): on top of the seeds the library keeps, per point, the 20 neighbours its last exact search found and its OWN
 * COPY of the map (112 bytes per point in all).  Every call compares d_xyzi with that copy, bit for bit, in the pass that re-frames it;
 * when nothing differs (and |q|^2 is within 1e-9 of 1: any quaternion normalised in fp64), a point whose list carries a certificate -- the gap behind its 20th
 * neighbour is wider than the fp32 rounding of the coordinates in any two frames can bridge -- takes its neighbours from the list instead
 * of searching (about 99 % of a map; rgc_stats::searched_target counts the rest); one differing coordinate and the call searches every
 * point again and rebuilds the lists.  The result is the search's, bit for bit, either way (RGC_KNN_CACHE=0 switches the lists off).
 * A lazy target (rgc_set_target_lazy) keeps seeds but no lists.
 * WHEN THIS HELPS, AND WHEN IT CANNOT: seeds and lists belong to point IDENTITIES.  They serve a caller that keeps its sub-map in the world
 * frame on the device and hands the same points over every frame -- i.e. one that has dropped the reference's per-frame body-frame leaf filter
 * (pcl::VoxelGrid over the re-framed sub-map, src/RGC_odometer.cpp:985-991: its centroids are a new point set every frame) and filters once
 * per keyframe instead (rgc_map_commit does).  A caller that keeps the reference's frame body (rgc::OdometryNode does) never hits them: its
 * target is a map the library has not seen, every frame, and costs the full search.  rgc_set_knn_reuse(ctx, RGC_REUSE_NONE) gives exactly
 * that cost for any map, and is what bench.py's `value` is timed with.
 * Preconditions: d_scratch must not overlap d_xyzi (RGC_ERR_INVALID: the input is read while the output is written, and a buffer has
 * one bounding-box hint); a d_scratch that is 16-byte aligned (anything hipMalloc / rgc_device_alloc returns) takes the fused path
 * -- re-framing inside the preparation's counting pass -- any other 4-byte aligned address the re-framing runs as its own launch.
 * A pose that is none -- a non-finite q or t (the NaN a diverged solve hands on through rgc_align_end_reframe), a zero quaternion -- is refused
 * with RGC_ERR_NONFINITE before anything is touched. */
RGC_API int rgc_set_target_reframed(rgc_ctx* ctx, const float* d_xyzi, int n, int stride_bytes, const double q_xyzw[4], const double t[3],
                                    float* d_scratch);
/* What a context keeps between the targets rgc_set_target_reframed prepares (no reference counterpart: the reference keeps nothing, it
 * builds a FastVGICP per frame, src/RGC_odometer.cpp:998).  Results never depend on the mode, bit for bit.
 *   RGC_REUSE_NONE   nothing: every target is searched like a map the library has not seen (no per-point state, no copy of the map)
 *   RGC_REUSE_SEEDS  the k-th distances of the last search (4 bytes per point)
 *   RGC_REUSE_LISTS  seeds + neighbour lists + the library's copy of the map (112 bytes per point; the default).  If the device cannot hold
 *                    them the context drops to RGC_REUSE_SEEDS by itself and carries on (rgc_get_knn_reuse tells).
 * Takes effect with the next rgc_set_target_reframed; lowering the mode frees the buffers it no longer needs (synchronises the context).
 * RGC_KNN_SEEDS=0 / RGC_KNN_CACHE=0 in the environment set a context's INITIAL mode to NONE / SEEDS (rgc_create). */
typedef enum rgc_knn_reuse { RGC_REUSE_NONE = 0, RGC_REUSE_SEEDS = 1, RGC_REUSE_LISTS = 2 } rgc_knn_reuse;
RGC_API int rgc_set_knn_reuse(rgc_ctx* ctx, int mode);
RGC_API int rgc_get_knn_reuse(const rgc_ctx* ctx, int* mode);

/* ---- A1-A8  ScanRegistration::laserCloudHandler (src/scanRegistration.cpp:89-730): range/NaN filter, ring + rel-time
 * assignment, curvature stencils, ground marking + weighted-PCA ground plane, occlusion mask, per-ring 6-sector
 * selection of sharp / flat / intensity features.  Input: the raw sensor cloud x,y,z,intensity in firing order
 * (what pcl::fromROSMsg yields from /velodyne_points, :107-108).  All output buffers are caller-allocated HOST memory.
 * Outputs = the node's published topics (:689-727): cloud = /velodyne_cloud_2 (ring-major, intensity = ring +
 * 0.1*relTime), sharp = /laser_cloud_sharp, inten = /laser_cloud_inten, flat = /laser_cloud_flat (features are
 * x,y,z,intensity,normal_x with normal_x the weight of :501,554,609), ground_pts = /laser_cloud_ground,
 * groundparam = /ground_param (ground_msg/groundparam.msg field order).  Per-point diagnostics may be NULL. */
typedef struct rgc_fe_params {
  int    n_scans;        /* scan_line: 16 / 32 / 64 (:57,69)          */
  double min_range;      /* minimum_range 0.5 (:59)                    */
  double max_range;      /* maxmum_range 80 (launch/run.launch:13)     */
  int    use_intensity;  /* USE_intensity (:58,645)                    */
} rgc_fe_params;
typedef struct rgc_fe_out {
  float* cloud; int cloud_cap; int n_cloud;                 /* cloud_cap points * 4 floats (NULL: not downloaded) */
  float* sharp; float* flat; float* inten; int feat_cap;    /* feat_cap features * 5 floats each                  */
  int n_sharp, n_sharp_own, n_flat, n_inten;                /* n_sharp includes the appended intensity corners    */
  float* ground_pts; int ground_cap; int n_ground;          /* ground_cap * 4 floats (may be NULL); n_ground total */
  double groundparam[11]; int ground_valid;                 /* ground_valid = 0 <=> "groundsize0" (:354-357)      */
  int ring_count[64];
  float *curvature, *curvature2, *inten_curvature;          /* cloud_cap each, optional                           */
  int *label, *inten_label, *picked, *ground_marked;        /* cloud_cap each, optional                           */
} rgc_fe_out;
RGC_API void rgc_default_fe_params(rgc_fe_params* p);
RGC_API int rgc_frontend(rgc_ctx* ctx, const float* xyzi, int n, int stride_bytes, const rgc_fe_params* params, rgc_fe_out* out);
/* the same with the sweep already resident on the device (a pointer from rgc_device_alloc, e.g. filled by
 * rgc_pc2_unpack(..., out_on_device = 1)): the message bytes are the only thing that crosses PCIe */
RGC_API int rgc_frontend_device(rgc_ctx* ctx, const float* d_xyzi, int n, int stride_bytes, const rgc_fe_params* params, rgc_fe_out* out);
/* the ring-major cloud of the LAST front-end call where it lies on the device (n x {x, y, z, ring + 0.1 relTime}, 16-byte stride), so
 * that de-skew / VoxelGrid / setInputSource can follow without a PCIe round trip (pass out->cloud = NULL to skip its download).
 * The pointer belongs to the context and is valid until the next front-end call; rgc_deskew(..., on_device = 1) may modify it in place. */
RGC_API int rgc_frontend_cloud_device(rgc_ctx* ctx, float** d_cloud, int* n);

/* ---- scalar host stages of the frame body (no GPU needed; kept in the same library so the adaptor is complete) ----
 * Quaternions are x,y,z,w.  ground[11] = ground_msg/groundparam order: norm xyz, vector1 xyz, vector2 xyz, distance,
 * source (ground_msg/msg/groundparam.msg:2-12; ground_s, include/rgc_slam/utility.h:382-389). */
/* C9  result extraction (src/RGC_odometer.cpp:1011-1016): Matrix4f -> (q_last_curr_l, t_last_curr_l) */
RGC_API int rgc_extract_pose(const float T[16], double q_xyzw[4], double t[3]);
/* B1  gyro-only delta-q guess + mid-point delta-p/delta-v (src/RGC_odometer.cpp:883-931,1418-1438) */
RGC_API int rgc_imu_preintegrate(const double* stamps, const double* gyr3, const double* acc3 /* may be NULL */, int n,
                                 double prev_time, double cur_time, double dq_xyzw[4], double dq2_xyzw[4] /* NULL ok */,
                                 double dp[3] /* NULL ok */, double dv[3] /* NULL ok */);
/* B1  the IMU side of the odometer's callbacks: vg_ICP::imu_callback (src/RGC_odometer.cpp:444-486: the first 100 messages are
 * dropped, biases removed) and ComplementaryFilter (:545-625: median filters over 201 / 41 / 41 accelerometer samples --
 * Mid_Filter, include/rgc_slam/utility.h -- complementary roll / pitch, integrated yaw) -> IMU.Rwi, the attitude the frame body
 * initialises the pose from (:866-871) and blends pitch / roll towards (:1206-1214).  Plain host code, no GPU. */
typedef struct rgc_imu_filter {
  double ba[3], bg[3];          /* imu_s::ba / bg (utility.h:253-254), subtracted from every sample */
  int    dropped;               /* of the first 100 messages (:446-451) */
  int    count;                 /* IMU.count: samples accepted */
  double t_last;                /* imu_last.t */
  double roll, pitch, yaw;      /* rad */
  double roll_last, pitch_last; /* imu_last.roll / pitch */
  double Rwi[9];                /* IMU.Rwi, row-major */
  double mf_buf[3][201];        /* Mid_Filter::data_buf of accx_MF(201), accy_MF(41), accz_MF(41) (:39) */
  int    mf_count[3];
} rgc_imu_filter;
RGC_API void rgc_imu_filter_init(rgc_imu_filter* f);
/* one sensor_msgs/Imu message.  Returns 1 and the bias-free sample (what the callback pushes into accBuf / gyrBuf for
 * rgc_imu_preintegrate) when the message is accepted, 0 while the first 100 are being dropped, < 0 on error. */
RGC_API int rgc_imu_filter_push(rgc_imu_filter* f, double stamp, const double acc[3], const double gyr[3], double acc_out[3], double gyr_out[3]);

/* B7  the ground-change detector in front of the fusion (src/RGC_odometer.cpp:1034-1087): a plane mismatch while the IMU pitches
 * (>= 0.02 twice and |pitch of delta_q_imu| > 0.5 deg) switches the ground factor off for 25 frames, after which the new plane is
 * re-associated with a remembered one (pitch / roll within 4 deg) or remembered itself.  Plain host code. */
typedef struct rgc_ground_gate {
  int    gflag;                 /* :328 */
  int    changegroundflag;      /* :327, starts at 25 */
  double q_w_curr_delta[4];     /* :20, x y z w */
  int    n_history;
  double history[64][4];        /* histoary_pose (:298); the oldest entry is overwritten beyond 64 */
} rgc_ground_gate;
RGC_API void rgc_ground_gate_init(rgc_ground_gate* g);
RGC_API void rgc_ground_gate_remember(rgc_ground_gate* g);  /* histoary_pose.push_back(q_w_curr_delta) of the first sub-map frame (:969) */
/* one frame: returns the new gflag (the ground factor is used when USE_GROUND && gflag == 0, :1088) and q_w_curr_f (:1086-1087) */
RGC_API int rgc_ground_gate_step(rgc_ground_gate* g, const double ground_last[11], const double ground_cur[11], const double q_lidar_xyzw[4],
                                 const double t_lidar[3], const double dq_imu_xyzw[4] /* NULL: identity */, const double q_w_curr_xyzw[4],
                                 double q_w_curr_f_xyzw[4]);

/* B7  the pose-fusion problem the reference builds for Ceres (src/RGC_odometer.cpp:1025-1032,1088-1119,1188-1193;
 * factors src/lidarFactor.hpp:132-172,228-265,311-350) */
typedef struct rgc_fuse_in {
  double q_lidar_xyzw[4];    /* q_last_curr_l (:1016)                                              */
  double t_lidar[3];         /* t_last_curr_l (:1015)                                              */
  double fitness;            /* vgicp_source = getFitnessScore() (:1010)                           */
  int    use_ground;         /* USE_GROUND && gflag == 0 (:1088)                                   */
  double ground_last[11];    /* ground_last                                                        */
  double ground_cur[11];     /* ground_cur                                                         */
  double q_w_curr_f_xyzw[4]; /* q_w_curr_delta^-1 * q_w_curr, normalised (:1086-1087)              */
  double ground_cov;         /* 0.2 (:1092)                                                        */
  int    use_imu;            /* USE_IMU && imuflag == 1 (:1104)                                    */
  double q_imu_xyzw[4];      /* delta_q_imu; its weight follows :1107-1116                         */
  int    max_iterations;     /* 6 (:1190)                                                          */
} rgc_fuse_in;
RGC_API void rgc_default_fuse_in(rgc_fuse_in* in);
RGC_API int rgc_fuse_pose(const rgc_fuse_in* in, double q_xyzw[4], double t[3], int* iterations /* NULL ok */);
/* B8  pose composition + 0.95/0.05 pitch-roll blend in DEGREES (src/RGC_odometer.cpp:1194-1214);
 * R_imu_wl = IMU.Rwi * R_il, row-major, only read when use_imu */
RGC_API int rgc_compose_pose(const double q_w_curr[4], const double t_w_curr[3], const double q_fused[4], const double t_fused[3],
                             const double t_lidar[3], int use_imu, const double R_imu_wl[9], double q_w_out[4], double t_w_out[3],
                             double t_last_curr_out[3] /* NULL ok */);
/* Utility::R2ypr / ypr2R (include/rgc_slam/utility.h:105-147): degrees, Rz*Ry*Rx, row-major 3x3 */
RGC_API void rgc_R2ypr(const double R[9], double ypr_deg[3]);
RGC_API void rgc_ypr2R(const double ypr_deg[3], double R[9]);

/* per-align statistics (reported by bench.py) */
typedef struct rgc_stats {
  int n_source, n_target, n_voxels, n_corr;
  int outer_iterations, n_linearize, n_error;
  long long target_cells, source_cells;
  int deferred_target, deferred_source; /* queries handled by the cooperative kNN kernel */
  double source_crowding;               /* mean number of points in a scan point's own kNN-grid cell (sum count^2 / n) */
  int lazy_misses;                      /* lazy target: solves repeated on the completed map since the context was created */
  int searched_target;                  /* queries of the target's last preparation whose k neighbours were SEARCHED: n_target, or -- a map handed
                                         * over again by rgc_set_target_reframed, unchanged -- the few whose neighbour list carries no certificate */
} rgc_stats;
RGC_API int rgc_get_stats(rgc_ctx* ctx, rgc_stats* out);

/* ---- device plumbing for callers that keep clouds resident in HBM (bench, ROS adaptor) ---- */
RGC_API int  rgc_device_alloc(rgc_ctx* ctx, size_t bytes, void** d_ptr);
RGC_API int  rgc_device_free(rgc_ctx* ctx, void* d_ptr);
/* Page-locked HOST memory (hipHostMalloc, visible to every device) for a caller's staging buffers: a copy between the device and
 * pageable memory is staged by the runtime in a blocking hop, and the first touch of fresh pageable pages costs milliseconds (the
 * host-staged node's one 8.8 ms frame in round 3); out of these buffers the copies of rgc_set_*, rgc_voxelgrid, rgc_frontend run at
 * PCIe rate.  No context needed; rgc_host_free(NULL) is a no-op. */
RGC_API int  rgc_host_alloc(size_t bytes, void** h_ptr);
RGC_API int  rgc_host_free(void* h_ptr);
RGC_API int  rgc_upload(rgc_ctx* ctx, void* d_dst, const void* h_src, size_t bytes);   /* async on ctx stream */
RGC_API int  rgc_download(rgc_ctx* ctx, void* h_dst, const void* d_src, size_t bytes); /* synchronous          */
RGC_API int  rgc_synchronize(rgc_ctx* ctx);
RGC_API void* rgc_stream(rgc_ctx* ctx); /* the context's main hipStream_t (everything except the source's preparation, see rgc_set_source_device) */

/* ---- f3: wire and disk formats at the edges of the path ----
 * sensor_msgs/PointCloud2 <-> device arrays without PCL: pcl::fromROSMsg (src/scanRegistration.cpp:107-108) and pcl::toROSMsg
 * (:689-727, src/RGC_odometer.cpp:1348).  Fields in the order x, y, z, intensity, ring, time; offset < 0 = the message has
 * no such field (-> 0, ring -> -1).  datatype = sensor_msgs/PointField constants (INT8 1 ... FLOAT32 7, FLOAT64 8).
 * strict != 0 reproduces fromROSMsg<PointXYZI>'s field matching: a field whose datatype is not FLOAT32 is NOT mapped (stays
 * 0); strict = 0 converts any numeric type. */
typedef struct rgc_pc2_layout {
  int point_step;
  int offset[6];
  int datatype[6];
  int is_bigendian;
  int strict;
} rgc_pc2_layout;
/* data: width*height points of point_step bytes (host).  xyzi_out: n x 4 floats (host, or device if out_on_device);
 * ring_out / time_out: n ints / floats, nullable (same residency). */
RGC_API int rgc_pc2_unpack(rgc_ctx* ctx, const void* data, int n_points, const rgc_pc2_layout* layout, float* xyzi_out, int* ring_out,
                           float* time_out, int out_on_device);
/* kind 0: PointXYZI message (point_step 32), in = n x 4 {x,y,z,intensity}; kind 1: PointXYZINormal message (point_step 48),
 * in = n x 5 {x,y,z,intensity,normal_x}.  data_out: n * point_step bytes (host). */
RGC_API int rgc_pc2_pack(rgc_ctx* ctx, int kind, const float* in, int n, int in_on_device, void* data_out);
typedef struct rgc_pc2_field { char name[16]; int offset; int datatype; int count; } rgc_pc2_field;
/* the PointField table pcl::toROSMsg emits for that point type; returns the number of fields (<= cap), *point_step set */
RGC_API int rgc_pc2_point_fields(int kind, rgc_pc2_field* out, int cap, int* point_step);
/* one line of the TUM trajectory file f_save_pose_evo (src/RGC_odometer.cpp:1315-1316): "stamp tx ty tz qx qy qz qw\n" with
 * std::fixed, 6 decimals for the stamp and 9 for the rest.  Returns the length written (excluding the NUL) or < 0. */
RGC_API int rgc_tum_line(double stamp, const double t[3], const double q_xyzw[4], char* buf, int cap);
/* key-frame cloud to a .pcd file (src/RGC_odometer.cpp:1353-1354 writes ASCII): binary = 0 -> DATA ascii like
 * pcl::io::savePCDFileASCII (precision 8), binary = 1 -> DATA binary, 16 bytes per point (FIELDS x y z intensity). */
RGC_API int rgc_pcd_write(const char* path, const float* xyzi, int n, int binary);

/* ---- f4: loop-closure ICP = pcl::IterativeClosestPoint<PointType2,PointType2> as configured and used at
 * src/RGC_mapping.cpp:2050-2069: setMaxCorrespondenceDistance, setMaximumIterations(100), setTransformationEpsilon(1e-6),
 * setEuclideanFitnessEpsilon(1e-6), setRANSACIterations(0); setInputSource(latestKeyFrameCloud), setInputTarget(
 * nearHistoryKeyFrameCloud), align() with the identity guess, hasConverged(), getFitnessScore(), getFinalTransformation(). */
typedef struct rgc_icp_params {
  int max_iterations;                 /* 100 */
  double max_correspondence_distance; /* poseGraphSearchRadius * 2 */
  double transformation_epsilon;      /* 1e-6: translation^2 <= eps and cos(angle) >= 1 - eps */
  double euclidean_fitness_epsilon;   /* 1e-6: relative change of the correspondence MSE */
} rgc_icp_params;
enum { RGC_ICP_NOT_CONVERGED = 0, RGC_ICP_ITERATIONS = 1, RGC_ICP_TRANSFORM = 2, RGC_ICP_ABS_MSE = 3, RGC_ICP_REL_MSE = 4,
       RGC_ICP_NO_CORRESPONDENCES = 5 };   /* pcl::registration::DefaultConvergenceCriteria::ConvergenceState */
typedef struct rgc_icp_result {
  int iterations, converged, state, n_correspondences;
  double fitness;                     /* getFitnessScore(): mean squared nearest-neighbour distance at the final transformation */
} rgc_icp_result;
RGC_API void rgc_default_icp_params(rgc_icp_params* p);
/* source / target: host AoS, x,y,z first, same stride.  final_T: row-major 4x4 float (getFinalTransformation()). */
RGC_API int rgc_icp_align(rgc_ctx* ctx, const float* source, int n_source, const float* target, int n_target, int stride_bytes,
                          const rgc_icp_params* params, float final_T[16], rgc_icp_result* result);

/* ---- f1: scan-to-map FEATURE registration of the mapping node, src/RGC_mapping.cpp:1069-1358 ----
 * Replaces, per mapping frame: kdtreeCornerFromMap/kdtreeSurfFromMap->setInputCloud (:1073-1074), the four association
 * loops (:1092-1282: pointAssociateToMap, 5-NN, PCA line test / QR plane fit) and ceres::Solve over para_q/para_t and
 * para_q_last/para_t_last with LidarEdgeFactor / LidarPlaneNormFactor under HuberLoss(0.1) (:1078-1341,
 * src/lidarFactor.hpp:9-51,91-121), twice (:1076).  The ground block (:1314-1340, Ground_DeltaFactor_goable) and the IMU
 * block (:1285-1312, RelativeRFactor + two PitchRollFactor) are optional inputs.
 * Features are n x 4 floats {x, y, z, weight} (PointXYZINormal's x,y,z,normal_x); quaternions are x,y,z,w. */
typedef struct rgc_mapreg_report {
  double initial_cost, final_cost;  /* ceres Summary: 1/2 sum rho(|r|^2) before / after the solve */
  int iterations, successful;       /* LM iterations run (<= 6) and accepted steps */
  int n_edge_cur, n_plane_cur, n_edge_last, n_plane_last; /* residual blocks: corner_num, surf_num, cornerLast_num, surfLast_num */
} rgc_mapreg_report;
/* one Ground_DeltaFactor_goable (src/lidarFactor.hpp:352-403) as created at RGC_mapping.cpp:1326-1337, NULL loss */
typedef struct rgc_mapreg_ground {
  double last_v1[3], last_v2[3], last_norm[3], last_distance; /* g_last: vector_1, vector_2, vector_norm, distance */
  double cur_norm[3], cur_distance;                          /* g_cur: vector_norm, distance */
  double q_history[4];                                       /* histoary_q = q_w_curr_f (x,y,z,w) */
  double last_q[4], last_t[3];                               /* last_q_q, last_t_t: the fixed previous pose */
  double p_var;                                              /* ground_cov (0.2) */
} rgc_mapreg_ground;
/* the IMU block of RGC_mapping.cpp:1285-1312 (USE_IMU == 1 && map_update != 0), NULL loss: RelativeRFactor::Create(delta_q_imu,
 * imu_cov) on (para_q_last, para_q) (src/lidarFactor.hpp:174-226) and PitchRollFactor::Create(pitch, roll, 0.02) on para_q and
 * on para_q_last (:434-468).  The caller computes imu_cov (0.004 / 0.4, :1288-1293) and the pitch / roll targets in radians
 * (ypr of IMUTemp.Rwi * R_il and of IMULast.Rwi * R_il, :1299-1310) as the reference does. */
typedef struct rgc_mapreg_imu {
  double delta_q[4];             /* delta_q_imu (x,y,z,w) */
  double imu_cov;                /* q_var of the RelativeRFactor */
  double pitch_cur, roll_cur;    /* pl_tmp, rl_tmp */
  double pitch_last, roll_last;  /* pl_last, rl_last */
  double pr_var;                 /* q_var of both PitchRollFactor (0.02) */
} rgc_mapreg_imu;
/* laserCloudCornerFromMapDS / laserCloudSurfFromMapDS (host AoS, x,y,z first; at least 5 points each) */
RGC_API int rgc_mapreg_set_maps(rgc_ctx* ctx, const float* corner_map, int n_corner, const float* surf_map, int n_surf, int stride_bytes);
/* association only (kind 0 = edge against the corner map, 1 = plane against the surf map): factors8 (nullable) receives
 * n x 8 doubles: edge {point_a[3], point_b[3], var, valid}, plane {norm[3], negative_OA_dot_norm, 0, 0, var, valid} */
RGC_API int rgc_mapreg_associate(rgc_ctx* ctx, int kind, const float* feat_xyzw, int n, const double q_xyzw[4], const double t[3],
                                 double* factors8, int* n_valid);
/* poses: q_w_curr[4] t_w_curr[3] q_w_last[4] t_w_last[3], in/out.  *gate_failed = 1 (poses untouched) when the size gate of
 * :1069 is not met.  report: one entry per pass of the two-pass loop (nullable).  ground_cur / ground_last (nullable): the
 * ground block on (para_q, para_t) resp. (para_q_last, para_t_last) -- pass them when the reference's condition at :1314 holds.
 * imu (nullable): the IMU block -- pass it when the condition at :1285 holds. */
RGC_API int rgc_mapreg_optimize(rgc_ctx* ctx, const float* corner_cur, int n_ccur, const float* surf_cur, int n_scur,
                                const float* corner_last, int n_clast, const float* surf_last, int n_slast,
                                const rgc_mapreg_ground* ground_cur, const rgc_mapreg_ground* ground_last,
                                const rgc_mapreg_imu* imu, double poses[14], rgc_mapreg_report report[2], int* gate_failed);


/* ---- f2: rolling local map resident on the device ----
 * The odometer keeps a deque of <= slipwide (3) keyframe clouds in the WORLD frame (src/RGC_odometer.cpp:1237-1247), and every
 * frame re-expresses all of them in the new body frame (:1248-1256), VoxelGrid-filters the concatenation (:985-991) and hands it
 * to a fresh FastVGICP (:998,1007): 3 kd-trees, N_t covariances and the voxel map are rebuilt per FRAME even when no keyframe
 * changed.  Here the keyframes stay in HBM in a map frame (world minus an origin that the caller keeps near the sensor so that
 * fp32 coordinates stay small), the registration runs in that frame (guess = T_w_curr * T_last_curr, result = the new world pose)
 * and the target (filter + grid + covariances + voxels) is rebuilt, on the device, only when a keyframe was inserted or evicted.
 * Numerics differ from the reference by the frame the 0.3 m leaf lattice and the 1 m voxel lattice are aligned to (map axes
 * instead of the previous body axes); the parity definition of this row: DESIGN.md section 6 (f2) and EXPERIMENTS.md "6e". */
typedef struct rgc_map_info {
  int n_keyframes;
  long long n_points;            /* world-frame points held (before the leaf filter) */
  int n_target;                  /* points of the committed target, -1 if the map changed since the last commit */
  unsigned long long revision;   /* bumped by every insert / evict / rebase / reset */
  int oldest_id, newest_id;      /* -1 when empty */
  double origin[3];
} rgc_map_info;
/* drops every keyframe; origin (nullable = 0,0,0): the world point the map frame is centred on */
RGC_API int rgc_map_reset(rgc_ctx* ctx, const double origin[3]);
/* surroundingCloud.push_back(transformPointCloud(cloud, q_w_curr, t_w_curr)) (:1237): q * p + (t - origin) in fp64, stored fp32,
 * intensity copied.  xyzi: x,y,z,intensity (stride >= 16), host or (on_device) device memory.  *keyframe_id (nullable): its id. */
RGC_API int rgc_map_insert(rgc_ctx* ctx, const float* xyzi, int n, int stride_bytes, const double q_w_xyzw[4], const double t_w[3],
                           int on_device, int* keyframe_id);
/* eviction: first every keyframe whose pose is farther than radius from center (skipped when center is NULL or radius <= 0), then
 * the oldest ones until at most max_keyframes remain (pop_front, :1242-1247; skipped when max_keyframes <= 0) */
RGC_API int rgc_map_evict(rgc_ctx* ctx, int max_keyframes, const double center[3], double radius, int* n_evicted);
/* moves the map frame's origin (every stored point shifts by old - new, fp64 -> fp32) */
RGC_API int rgc_map_rebase(rgc_ctx* ctx, const double new_origin[3]);
/* makes the map the registration target: pcl::VoxelGrid(leaf) over the keyframes in insertion order (:985-991) + setInputTarget
 * (:1007), all on the device.  A no-op when nothing changed since the last commit and the target is still bound (rgc_set_target*
 * unbinds it).  Poses passed to rgc_align / rgc_linearize are then map-frame poses (world translation minus origin).
 * With a solve in flight on the context anything but that no-op is refused before a byte is written (the filter's output buffer is the one
 * the bound target was set from); a commit that fails further down leaves the context WITHOUT a target rather than with one whose input
 * has been overwritten. */
RGC_API int rgc_map_commit(rgc_ctx* ctx, float leaf, int* n_target);
RGC_API int rgc_map_get_info(rgc_ctx* ctx, rgc_map_info* out);
/* which = 0: the stored keyframe points, 1: the committed target; up to cap points (x,y,z,intensity) to the host, *n = total */
RGC_API int rgc_map_download(rgc_ctx* ctx, int which, float* out_xyzi, int cap, int* n);

/* ---- in-library kernel timing with HIP events on the context's stream (bench.py roofline) ---- */
enum {
  RGC_K_GRID = 0,      /* bbox + count + scan + scatter + rank/gather                         */
  RGC_K_KNN_COV = 1,   /* exact kNN + covariance + normal, bulk kernel on the TARGET (dominant) */
  RGC_K_VOXEL = 2,     /* Gaussian voxel map reduction                                        */
  RGC_K_LINEARIZE = 3, /* correspondences + Mahalanobis + H/b/cost                            */
  RGC_K_ERROR = 4,     /* frozen-correspondence cost                                          */
  RGC_K_FITNESS = 5,   /* 1-NN fitness                                                        */
  RGC_K_KNN_COV_SRC = 6, /* the same kernel on the (small) source cloud, separate instantiation */
  RGC_K_KNN_COOP = 7,    /* cooperative kernel (one wave per deferred query), target               */
  RGC_K_KNN_COOP_SRC = 8,/* cooperative kernel, source                                             */
  RGC_K_COUNT = 9
};
RGC_API int rgc_profile_enable(rgc_ctx* ctx, int on);
/* restrict the event regions to the kinds whose bit is set (bit k = kind k; default: all).  Each region costs two
 * hipEventRecord calls, so a timed run that only needs one kernel's duration selects just that kind. */
RGC_API int rgc_profile_select(rgc_ctx* ctx, unsigned kind_mask);
RGC_API int rgc_profile_reset(rgc_ctx* ctx);
/* launches = timed regions of that kind; total_ms = summed hipEventElapsedTime; last_n = points of the last region */
RGC_API int rgc_profile_get(rgc_ctx* ctx, int kind, long long* launches, double* total_ms, long long* total_points);
RGC_API const char* rgc_profile_name(int kind);

#ifdef __cplusplus
}
#endif
#endif /* RGC_HIP_H */
