// fast_vgicp_hip.hpp -- header-only C++ adaptor over the C-ABI (include/rgc_hip.h) exposing the calls the RGC-SLAM
// odometer makes on fast_gicp::FastVGICP (/root/reference/rgc_slam/src/RGC_odometer.cpp:998-1011), with the same
// method names and argument meaning, WITHOUT requiring PCL/Eigen at build time:
//
//   rgc::FastVGICPHip vgicp;                          // fast_gicp::FastVGICP<pcl::PointXYZI, pcl::PointXYZI> vgicp;
//   vgicp.setResolution(1.0); vgicp.setMaximumIterations(25); vgicp.setMaxCorrespondenceDistance(2);
//   vgicp.setTransformationEpsilon(1e-6); vgicp.setEuclideanFitnessEpsilon(1e-6); vgicp.setRANSACIterations(0);
//   vgicp.setNumThreads(14);
//   vgicp.setInputTarget(target); vgicp.setInputSource(source);   // any cloud with ->points / ->size() (pcl::PointCloud::Ptr)
//   vgicp.align(*aligned, T2);                                     // T2: anything with operator()(row, col) (Eigen::Matrix4f)
//   double score = vgicp.getFitnessScore();  auto T = vgicp.getFinalTransformation<Eigen::Matrix4f>();
//
// Errors: like the reference, the solver itself never throws ("lm not converged" is lmFailed(), hasConverged() as in PCL) and NO SETTER
// throws: the reference's setters return void and accept anything.  A setter the library refuses (an invalid parameter) leaves its status
// in lastSetterStatus() / lastSetterError() and the requested value pending, so that the NEXT call that would compute something --
// setInputTarget / setInputSource / align -- fails instead of running under other settings: those, like HIP failures, throw
// std::runtime_error carrying rgc_last_error().  No CPU fallback.
#pragma once
#include <cmath>
#include <cstddef>
#include <cstring>
#include <stdexcept>
#include <string>
#include <utility>
#include <memory>
#include <vector>

#include "../../include/rgc_hip.h"

namespace rgc {

enum class NeighborSearchMethod { DIRECT27 = RGC_DIRECT27, DIRECT7 = RGC_DIRECT7, DIRECT1 = RGC_DIRECT1 };  // gicp_settings.hpp:8
enum class RegularizationMethod { NONE, MIN_EIG, NORMALIZED_MIN_EIG, PLANE, FROBENIUS };                    // gicp_settings.hpp:6
enum class VoxelAccumulationMode { ADDITIVE, ADDITIVE_WEIGHTED, MULTIPLICATIVE };                            // gicp_settings.hpp:10

class FastVGICPHip {
public:
  explicit FastVGICPHip(int hip_device = 0) {
    rgc_default_params(&p_);
    p_.max_iterations = 64;       // LsqRegistration() default, lsq_registration_impl.hpp:11
    p_.translation_eps = 5e-4;    // lsq_registration_impl.hpp:13
    int rc = rgc_create(hip_device, &p_, &ctx_);
    if (rc != RGC_OK) throw std::runtime_error(std::string("rgc_create: ") + rgc_status_string(rc));
    set_identity(final_);
  }
  ~FastVGICPHip() { rgc_destroy(ctx_); }
  FastVGICPHip(const FastVGICPHip&) = delete;
  FastVGICPHip& operator=(const FastVGICPHip&) = delete;

  // ---- setters: pcl::Registration / LsqRegistration / FastGICP / FastVGICP ----
  void setResolution(double r) { p_.voxel_res = r; push(); }                       // fast_vgicp_impl.hpp:32-34
  void setMaximumIterations(int n) { p_.max_iterations = n; push(); }
  void setTransformationEpsilon(double e) { p_.translation_eps = e; push(); }
  void setRotationEpsilon(double e) { p_.rotation_eps = e; push(); }               // lsq_registration_impl.hpp:27-29
  void setInitialLambdaFactor(double f) { p_.lm_init_lambda_factor = f; push(); }  // lsq_registration_impl.hpp:32-34
  void setCorrespondenceRandomness(int k) { p_.k_correspondences = k; push(); }    // fast_gicp_impl.hpp:41-43
  void setNeighborSearchMethod(NeighborSearchMethod m) { p_.neighbor_method = (int)m; push(); }
  // the odometer leaves both at the constructor's values (PLANE, fast_gicp_impl.hpp:20; ADDITIVE, fast_vgicp_impl.hpp:24), which run on the
  // tuned kernels; every other value is implemented on the library's general route (rgc_set_regularization_method in rgc_hip.h: a 3x3 per
  // point, unoptimised).  Select before setInputTarget / setInputSource: a change drops the clouds that were set under the other setting.
  void setRegularizationMethod(RegularizationMethod m) { note(rgc_set_regularization_method(ctx_, (int)m)); }
  void setVoxelAccumulationMode(VoxelAccumulationMode m) { note(rgc_set_voxel_accumulation_mode(ctx_, (int)m)); }
  int lastSetterStatus() const { return setter_status_; }                 // RGC_OK, or why the last refused setter was refused
  const std::string& lastSetterError() const { return setter_error_; }
  void setMaxCorrespondenceDistance(double) {}   // unused by FastVGICP
  void setEuclideanFitnessEpsilon(double) {}     // no-op in LsqRegistration
  void setRANSACIterations(int) {}               // no-op
  void setNumThreads(int) {}                     // CPU threads of the reference; nothing to set on the GPU
  void setDebugPrint(bool) {}                    // LsqRegistration::setDebugPrint (lsq_registration.hpp:53, impl :38-40): the LM trace on stdout (:59, :147); accepted, nothing is printed

  // ---- clouds ----
  void setInputTarget(const float* xyz, int n, int stride_bytes) { params(); chk(rgc_set_target(ctx_, xyz, n, stride_bytes)); n_tgt_ = n; fit_valid_ = false; }
  void setInputSource(const float* xyz, int n, int stride_bytes) { params(); chk(rgc_set_source(ctx_, xyz, n, stride_bytes)); n_src_ = n; fit_valid_ = false; }
  template <class CloudPtr>
  void setInputTarget(const CloudPtr& cloud) { setInputTarget(&cloud->points[0].x, (int)cloud->points.size(), (int)sizeof(cloud->points[0])); }
  template <class CloudPtr>
  void setInputSource(const CloudPtr& cloud) { setInputSource(&cloud->points[0].x, (int)cloud->points.size(), (int)sizeof(cloud->points[0])); }
  void setInputTargetDevice(const float* d_xyz, int n, int stride_bytes) { params(); chk(rgc_set_target_device(ctx_, d_xyz, n, stride_bytes)); n_tgt_ = n; fit_valid_ = false; }
  void setInputSourceDevice(const float* d_xyz, int n, int stride_bytes) { params(); chk(rgc_set_source_device(ctx_, d_xyz, n, stride_bytes)); n_src_ = n; fit_valid_ = false; }

  // The odometer's sub-map re-framing and setInputTarget in one call on device memory (RGC_odometer.cpp:1248-1256, 1007;
  // rgc_set_target_reframed): the map at d_map (x,y,z,intensity..., fixed between calls) re-expressed by q (x,y,z,w) and t into d_scratch
  // (n * 16 bytes) and prepared as the target.  A map handed over like this again and again is searched from what the last search found
  // (rgc_hip.h: seeds) -- same results, a shorter kNN launch.
  void setInputTargetReframed(const float* d_map, int n, int stride_bytes, const double q_xyzw[4], const double t[3], float* d_scratch) {
    params();
    chk(rgc_set_target_reframed(ctx_, d_map, n, stride_bytes, q_xyzw, t, d_scratch));
    n_tgt_ = n; fit_valid_ = false;
  }
  // Opt-in (rgc_set_target_lazy): covariances and voxels of a target are built only within margin_cells voxels of where the scan falls at
  // align()'s guess -- only the voxels a solve looks up enter its cost (fast_vgicp_impl.hpp:73-116).  Every look-up is checked and a solve
  // that leaves the built part is repeated on the completed map: results are the full build's bit for bit.  0 switches it off.
  void setLazyTarget(int margin_cells) { chk(rgc_set_target_lazy(ctx_, margin_cells)); }
  // What is kept between the targets setInputTargetReframed prepares (rgc_set_knn_reuse): RGC_REUSE_NONE / _SEEDS / _LISTS (default).  A frame
  // body that keeps the reference's per-frame leaf filter of the sub-map (RGC_odometer.cpp:985-991) hands over a new point set every frame
  // and gains nothing from seeds or lists; NONE then saves their 112 bytes per point.  Results never depend on the mode.
  void setNeighbourReuse(int mode) { chk(rgc_set_knn_reuse(ctx_, mode)); }
  int  getNeighbourReuse() const { int m = 0; rgc_get_knn_reuse(ctx_, &m); return m; }
  // two contexts taking turns on a dependent sequence: this context's scan preparation (enqueued now) is held back until `other`'s map
  // preparation has finished, so that it runs under other's solve instead of beside the launch other's frame is waiting for
  void holdSourceUntilTargetOf(FastVGICPHip& other) { chk(rgc_hold_source_until_target_of(ctx_, other.ctx_)); }
  // fast_gicp.hpp:55-61: the clouds change roles (fast_vgicp_impl.hpp:46-53) / are dropped / get covariances from the caller (n*9 doubles,
  // row-major, plane-regularised form only -- see rgc_set_source_covariances)
  void swapSourceAndTarget() { chk(rgc_swap_source_and_target(ctx_)); std::swap(n_src_, n_tgt_); fit_valid_ = false; }
  void clearSource() { chk(rgc_clear_source(ctx_)); n_src_ = 0; fit_valid_ = false; }
  void clearTarget() { chk(rgc_clear_target(ctx_)); n_tgt_ = 0; fit_valid_ = false; }
  void setSourceCovariances(const std::vector<double>& cov9) { chk(rgc_set_source_covariances(ctx_, cov9.data(), (int)(cov9.size() / 9))); fit_valid_ = false; }
  void setTargetCovariances(const std::vector<double>& cov9) { chk(rgc_set_target_covariances(ctx_, cov9.data(), (int)(cov9.size() / 9))); fit_valid_ = false; }

  // ---- pcl::Registration::align(output, guess) ----
  // guess: row-major float[16]
  void align(const float guess[16]) {
    int it = 0, conv = 0, fail = 0;
    params();
    chk(rgc_align(ctx_, guess, final_, hessian_, nullptr, &it, &conv, &fail));
    iterations_ = it; converged_ = conv != 0; lm_failed_ = fail != 0; fit_valid_ = false;
  }
  void align() { float I[16]; set_identity(I); align(I); }
  // align() in two halves (rgc_align_begin / rgc_align_end): between them the caller may prepare the next frame's clouds on ANOTHER
  // FastVGICPHip (rgc::PipelinedVGICP below); want_fitness chains getFitnessScore behind the solve
  void alignBegin(const float guess[16], bool want_fitness = false) {
    params();
    chk(rgc_align_begin(ctx_, guess, want_fitness ? 1 : 0));
    pending_fitness_ = want_fitness;
  }
  void alignEnd() {
    int it = 0, conv = 0, fail = 0;
    chk(rgc_align_end(ctx_, final_, hessian_, pending_fitness_ ? &fitness_ : nullptr, &it, &conv, &fail));
    iterations_ = it; converged_ = conv != 0; lm_failed_ = fail != 0; fit_valid_ = pending_fitness_;
  }
  // alignEnd() and, in the same call, the two steps a frame loop without a fusion stage takes with the result (rgc_align_end_reframe):
  // world_T <- world_T * final (fp64, row-major 4x4; RGC_odometer.cpp:1201-1203) and the NEXT frame's target on `next` (this object or
  // the other one of a pair) -- the map re-expressed in the new body frame (:1250-1255), as setInputTargetReframed would.
  void alignEndReframe(FastVGICPHip& next, double world_T[16], const float* d_map, int n, int stride_bytes, float* d_scratch) {
    int it = 0, conv = 0, fail = 0;
    chk(rgc_align_end_reframe(ctx_, next.ctx_, world_T, d_map, n, stride_bytes, d_scratch, final_, hessian_, pending_fitness_ ? &fitness_ : nullptr, &it,
                              &conv, &fail));
    iterations_ = it; converged_ = conv != 0; lm_failed_ = fail != 0; fit_valid_ = pending_fitness_;
    next.n_tgt_ = n; next.fit_valid_ = false;
  }
  // register to the target `owner` has prepared, without preparing or copying it (rgc_share_target)
  void shareTargetFrom(FastVGICPHip& owner) { chk(rgc_share_target(ctx_, owner.ctx_)); n_tgt_ = owner.n_tgt_; fit_valid_ = false; }
  // output: any cloud with .points (resized to the source size, x/y/z written); guess: operator()(r,c)
  template <class Cloud, class Mat4>
  void align(Cloud& output, const Mat4& guess) {
    float g[16];
    for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) g[r * 4 + c] = (float)guess(r, c);
    align(g);
    output.points.resize((size_t)n_src_);
    if (n_src_ > 0) chk(rgc_get_aligned(ctx_, final_, &output.points[0].x, (int)sizeof(output.points[0])));
  }
  void getAligned(float* out_xyz, int stride_bytes) { chk(rgc_get_aligned(ctx_, final_, out_xyz, stride_bytes)); }

  const float* getFinalTransformation() const { return final_; }  // row-major 4x4
  template <class Mat4>
  Mat4 getFinalTransformation() const { Mat4 m; for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) m(r, c) = final_[r * 4 + c]; return m; }
  const double* getFinalHessian() const { return hessian_; }      // row-major 6x6, lsq_registration_impl.hpp:43-45
  bool hasConverged() const { return converged_; }
  bool lmFailed() const { return lm_failed_; }                    // "lm not converged!!", lsq_registration_impl.hpp:69-72
  int  iterations() const { return iterations_; }
  double getFitnessScore() {                                      // RGC_odometer.cpp:1010
    if (!fit_valid_) { chk(rgc_fitness(ctx_, final_, &fitness_)); fit_valid_ = true; }
    return fitness_;
  }
  // LsqRegistration::evaluateCost (lsq_registration_impl.hpp:48-50); H (36) and b (6) may be null
  double evaluateCost(const float pose[16], double* H = nullptr, double* b = nullptr) {
    double T[16], cost = 0;
    for (int i = 0; i < 16; i++) T[i] = (double)pose[i];
    chk(rgc_linearize(ctx_, T, H, b, &cost));
    return cost;
  }
  // the accelerator seam of LsqRegistration (lsq_registration.hpp:68-69)
  double linearize(const double T[16], double H[36], double b[6]) { double c = 0; chk(rgc_linearize(ctx_, T, H, b, &c)); return c; }
  double compute_error(const double T[16]) { double c = 0; chk(rgc_compute_error(ctx_, T, &c)); return c; }

  std::vector<double> getSourceCovariances() { std::vector<double> c((size_t)n_src_ * 9); chk(rgc_get_source_covariances(ctx_, c.data(), nullptr)); return c; }
  std::vector<double> getTargetCovariances() { std::vector<double> c((size_t)n_tgt_ * 9); chk(rgc_get_target_covariances(ctx_, c.data(), nullptr)); return c; }
  rgc_ctx* context() { return ctx_; }
  rgc_stats stats() { rgc_stats st; chk(rgc_get_stats(ctx_, &st)); return st; }   // (lazy_misses: solves repeated on the completed map)

private:
  static void set_identity(float m[16]) { std::memset(m, 0, 16 * sizeof(float)); m[0] = m[5] = m[10] = m[15] = 1.f; }
  void chk(int rc) { if (rc != RGC_OK) throw std::runtime_error(std::string(rgc_status_string(rc)) + ": " + rgc_last_error(ctx_)); }
  // a parameter setter: the library takes the block or refuses it as a whole; refused, the requested values stay in p_ and the next
  // call that computes something offers them again (params()) -- and throws if they are still refused
  void note(int rc) { setter_status_ = rc; setter_error_ = rc == RGC_OK ? std::string() : std::string(rgc_status_string(rc)) + ": " + rgc_last_error(ctx_); }
  void push() { const int rc = rgc_set_params(ctx_, &p_); params_dirty_ = rc != RGC_OK; note(rc); }
  void params() { if (params_dirty_) { chk(rgc_set_params(ctx_, &p_)); params_dirty_ = false; } }
  int setter_status_ = RGC_OK;
  std::string setter_error_;
  bool params_dirty_ = false;
  rgc_ctx* ctx_ = nullptr;
  rgc_params p_{};
  float final_[16];
  double hessian_[36] = {0};
  double fitness_ = 0;
  bool fit_valid_ = false, converged_ = false, lm_failed_ = false, pending_fitness_ = false;
  int iterations_ = 0, n_src_ = 0, n_tgt_ = 0;
};

// A SEQUENCE of registrations on `depth` contexts taking turns (the C++ twin of rgc_slam_amd.registration.PipelinedVGICP): while frame
// i is being solved on one context the clouds of the next frames are prepared on the others -- only the solve needs the previous
// frame's pose.  Same kernels, same inputs, same order per frame: the poses are those of align() one frame at a time.
//   set_clouds(i, reg)    set target and source of frame i on `reg` (or only the source after shareTarget())
//   next_guess(i, T_i, g) write frame i + 1's guess into g (default: T_i)
//   on_result(i, reg)     frame i is done; `reg` holds its results until it is given frame i + depth
class PipelinedVGICP {
public:
  explicit PipelinedVGICP(int hip_device = 0, int depth = 2) {
    if (depth < 2) depth = 2;
    for (int k = 0; k < depth; k++) regs_.emplace_back(new FastVGICPHip(hip_device));
  }
  FastVGICPHip& context(int k) { return *regs_[(size_t)k]; }
  int depth() const { return (int)regs_.size(); }
  // the other contexts register to the target context(0) holds (a resident map): call again after context(0) prepared a new one
  void shareTarget() { for (size_t k = 1; k < regs_.size(); k++) regs_[k]->shareTargetFrom(*regs_[0]); }

  template <class SetClouds, class NextGuess, class OnResult>
  void run(int n_frames, SetClouds&& set_clouds, const float guess0[16], bool want_fitness, NextGuess&& next_guess, OnResult&& on_result) {
    const int D = depth();
    for (int j = 0; j < D - 1 && j < n_frames; j++) set_clouds(j, *regs_[(size_t)(j % D)]);
    float g[16];
    std::memcpy(g, guess0, sizeof(g));
    for (int i = 0; i < n_frames; i++) {
      FastVGICPHip& cur = *regs_[(size_t)(i % D)];
      cur.alignBegin(g, want_fitness);
      const int j = i + D - 1;                    // the context of frame i - 1 is free: the frame D - 1 ahead goes there
      if (j < n_frames) set_clouds(j, *regs_[(size_t)(j % D)]);
      cur.alignEnd();
      on_result(i, cur);
      next_guess(i, cur.getFinalTransformation(), g);
    }
  }
  template <class SetClouds, class OnResult>
  void run(int n_frames, SetClouds&& set_clouds, const float guess0[16], bool want_fitness, OnResult&& on_result) {
    run(n_frames, set_clouds, guess0, want_fitness, [](int, const float* T, float* g) { std::memcpy(g, T, 16 * sizeof(float)); }, on_result);
  }

private:
  std::vector<std::unique_ptr<FastVGICPHip>> regs_;
};

// The odometer's frame loop on device-resident clouds (RGC_odometer.cpp:976-1023, 1201-1203, 1248-1256; the C++ twin of bench.py's
// DependentSequence): frame i registers scan i to the local map re-expressed in the body frame of world pose i - 1 -- nothing of frame
// i's map can be prepared before frame i - 1 is solved.  With two registrations taking turns the next scan's preparation (it depends on
// no pose) runs under the current solve; the result, the pose composition and the next frame's target are one call (alignEndReframe).
//   rgc::DependentSequence seq(reg_a, &reg_b, d_map, n_map, 16);     // or (reg_a, nullptr, ...): one frame at a time
//   seq.run(n_frames, world_T, guess0, set_source, on_result);        // set_source(i, reg): reg.setInputSourceDevice(scan i ...)
class DependentSequence {
public:
  DependentSequence(FastVGICPHip& a, FastVGICPHip* b, const float* d_map, int n_map, int stride_bytes) : d_map_(d_map), n_(n_map), stride_(stride_bytes) {
    regs_[0] = &a; regs_[1] = b;
    for (int k = 0; k < 2; k++) {
      if (!regs_[k]) continue;
      void* p = nullptr;
      if (rgc_device_alloc(regs_[k]->context(), (size_t)n_map * 16, &p) != RGC_OK) throw std::runtime_error("rgc::DependentSequence: device allocation failed");
      scratch_[k] = (float*)p;
    }
  }
  ~DependentSequence() {
    for (int k = 0; k < 2; k++) if (scratch_[k]) rgc_device_free(regs_[k]->context(), scratch_[k]);
  }
  DependentSequence(const DependentSequence&) = delete;
  DependentSequence& operator=(const DependentSequence&) = delete;

  // world -> body of pose Tw (row-major 4x4, fp64): q = the rotation's inverse as a unit quaternion (x,y,z,w), t = -R^T t_w (:1250-1255)
  static void worldToBody(const double Tw[16], double q[4], double t[3]) {
    const double R[3][3] = {{Tw[0], Tw[4], Tw[8]}, {Tw[1], Tw[5], Tw[9]}, {Tw[2], Tw[6], Tw[10]}};   // R^T
    const double tr = R[0][0] + R[1][1] + R[2][2];
    if (tr > 0) {
      const double s4 = 2.0 * std::sqrt(tr + 1.0);
      q[0] = (R[2][1] - R[1][2]) / s4; q[1] = (R[0][2] - R[2][0]) / s4; q[2] = (R[1][0] - R[0][1]) / s4; q[3] = 0.25 * s4;
    } else {
      const int i = (R[0][0] >= R[1][1] && R[0][0] >= R[2][2]) ? 0 : (R[1][1] >= R[2][2] ? 1 : 2);
      const int j = (i + 1) % 3, k = (i + 2) % 3;
      const double s4 = 2.0 * std::sqrt(1.0 + R[i][i] - R[j][j] - R[k][k]);
      q[3] = (R[k][j] - R[j][k]) / s4; q[i] = 0.25 * s4; q[j] = (R[j][i] + R[i][j]) / s4; q[k] = (R[k][i] + R[i][k]) / s4;
    }
    const double nrm = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    for (int a = 0; a < 4; a++) q[a] /= nrm;
    for (int a = 0; a < 3; a++) t[a] = -(R[a][0] * Tw[3] + R[a][1] * Tw[7] + R[a][2] * Tw[11]);
  }

  // world_T: in the pose before the first frame, out the pose after the last.  guess0: the first frame's guess (row-major float[16]);
  // every later frame starts from the previous frame's motion.  on_result(i, reg): frame i is done, reg holds its results.
  template <class SetSource, class OnResult>
  void run(int n_frames, double world_T[16], const float guess0[16], bool want_fitness, SetSource&& set_source, OnResult&& on_result) {
    const int D = regs_[1] ? 2 : 1;
    float g[16];
    std::memcpy(g, guess0, sizeof(g));
    if (D == 2) set_source(0, *regs_[0]);
    double q[4], t[3];
    worldToBody(world_T, q, t);
    regs_[0]->setInputTargetReframed(d_map_, n_, stride_, q, t, scratch_[0]);
    for (int i = 0; i < n_frames; i++) {
      FastVGICPHip& cur = *regs_[i % D];
      FastVGICPHip& nxt = *regs_[(i + 1) % D];
      if (D == 1) set_source(i, cur);
      cur.alignBegin(g, want_fitness);
      if (D == 2 && i + 1 < n_frames) {
        nxt.holdSourceUntilTargetOf(cur);
        set_source(i + 1, nxt);
      }
      if (i + 1 < n_frames) {
        cur.alignEndReframe(nxt, world_T, d_map_, n_, stride_, scratch_[(i + 1) % D]);
      } else {
        cur.alignEnd();
        double W[16];
        const float* T = cur.getFinalTransformation();
        for (int a = 0; a < 4; a++)
          for (int b = 0; b < 4; b++) {
            double v = 0.0;
            for (int k = 0; k < 4; k++) v += world_T[a * 4 + k] * (double)T[k * 4 + b];
            W[a * 4 + b] = v;
          }
        std::memcpy(world_T, W, sizeof(W));
      }
      on_result(i, cur);
      std::memcpy(g, cur.getFinalTransformation(), sizeof(g));
    }
  }

private:
  FastVGICPHip* regs_[2] = {nullptr, nullptr};
  float* scratch_[2] = {nullptr, nullptr};
  const float* d_map_;
  int n_, stride_;
};

}  // namespace rgc
