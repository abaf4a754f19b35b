// odometry_node.hpp -- header-only C++ host side of the odometry path over the C-ABI (include/rgc_hip.h): what the two ROS
// nodes on the path do per LiDAR message, WITHOUT ROS/PCL/Eigen/Ceres at build time:
//
//   ScanRegistration::laserCloudHandler   /root/reference/rgc_slam/src/scanRegistration.cpp:89-730   (sensor_msgs::PointCloud2 in)
//   vg_ICP::ICP_thread frame body         /root/reference/rgc_slam/src/RGC_odometer.cpp:932-1322     (nav_msgs::Odometry out)
//
//   rgc::OdometryNode node(opt);
//   rgc::OdometryMsg odom; rgc::GroundMsg ground;
//   node.handlePointCloud2(msg.data.data(), msg.width * msg.height, layout, stamp, &odom, &ground);   // the subscriber callback
//   // odom -> nav_msgs::Odometry (frame_id "camera_init", child "laser_odom", pose only: RGC_odometer.cpp:1264-1275)
//   // ground -> ground_msg::groundparam (field order of ground_msg/msg/groundparam.msg:1-12)
//
// Every per-point stage is a call into librgc_hip.so (front-end, de-skew, VoxelGrid, registration, fitness, transforms); this class
// only holds the node's scalar state (poses, ground_last, the keyframe bookkeeping).  Two local-map modes:
//   resident_map = false : the reference's semantics -- keyframe deque on the host, re-framed / re-filtered / re-uploaded per frame
//                          (:1218-1256, 985-991, 1007)
//   resident_map = true  : SURVEY 8f row f2 -- keyframes stay on the device in a map frame, target rebuilt only on a keyframe change
//   device_chain = true  : the sweep never returns to the host between the stages: message bytes -> unpack kernel -> front-end ->
//                          de-skew -> VoxelGrid -> setInputSource / keyframe insert all read the previous stage's DEVICE buffer; only
//                          the message goes up and features, ground parameters and the pose come down.  With resident_map = false the
//                          reference's keyframe window lives on the device too: every frame its (at most three) world-frame clouds are
//                          re-expressed in the new body frame, concatenated, leaf-filtered and set as the target there (:1248-1256,
//                          985-991, 1007) -- the reference's semantics, bit for bit the host-staged mode's poses, without its PCIe traffic
// use_imu (launch/run.launch:18, the reference's default): imuCallback() feeds the attitude filter and the sample buffer
//   (vg_ICP::imu_callback, :444-486); the gyro's pre-integrated rotation is the registration's guess (:883-931, 993-996) and a factor
//   of the fusion (:1104-1119); pitch / roll are blended towards the filter's attitude (:1206-1214); the first `first_frames` sweeps
//   only initialise the pose from it (:857-882); a sweep without IMU coverage is dropped (:885-891).  The ground-change detector
//   (:1034-1087) runs in both modes.  Not mirrored: the two gravity-direction solves of the first frame (:1121-1186), whose results
//   do not enter the pose.
// Errors: std::runtime_error carrying rgc_last_error(); there is no CPU fallback.
#pragma once
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <exception>
#include <mutex>
#include <new>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "../../include/rgc_hip.h"

namespace rgc {

// std::allocator over page-locked host memory (rgc_host_alloc): what the node's host staging vectors are made of
template <typename T>
struct PinnedAllocator {
  using value_type = T;
  PinnedAllocator() = default;
  template <typename U> PinnedAllocator(const PinnedAllocator<U>&) {}
  T* allocate(std::size_t n) {
    void* p = nullptr;
    if (rgc_host_alloc(n * sizeof(T), &p) != RGC_OK || !p) throw std::bad_alloc();
    return static_cast<T*>(p);
  }
  void deallocate(T* p, std::size_t) noexcept { (void)rgc_host_free(p); }
  template <typename U> bool operator==(const PinnedAllocator<U>&) const { return true; }
  template <typename U> bool operator!=(const PinnedAllocator<U>&) const { return false; }
};
using HostVec = std::vector<float, PinnedAllocator<float>>;

struct OdometryMsg {             // the pose part of nav_msgs::Odometry as filled at RGC_odometer.cpp:1264-1275
  double stamp = 0;
  double position[3] = {0, 0, 0};
  double orientation_xyzw[4] = {0, 0, 0, 1};
  static const char* frame_id() { return "camera_init"; }
  static const char* child_frame_id() { return "laser_odom"; }
};
struct GroundMsg {               // ground_msg::groundparam, scanRegistration.cpp:420-430
  double param[11] = {0};
  bool valid = false;
};

class OdometryNode {
public:
  struct Options {
    int hip_device = 0;
    int scan_line = 16;                    // launch/run.launch scan_line
    double minimum_range = 0.5, maxmum_range = 80.0;
    bool use_ground = true;                // USE_GROUND
    bool resident_map = false;
    int max_keyframes = 3;                 // slipwide, RGC_odometer.cpp:299
    double evict_radius = 0.0;             // resident map only: additionally evict keyframes farther than this (0 = off)
    double rebase_distance = 50.0;         // resident map only: the map origin follows the sensor
    bool device_chain = false;             // keep the sweep -- and, without resident_map, the reference's keyframe window and its re-framed sub-map -- on the device between the stages
    bool use_imu = false;                  // USE_IMU (launch/run.launch:18)
    int first_frames = 0;                  // firstflagnum, RGC_odometer.cpp:303 (the reference: 10)
    double init_yaw = 0.0;                 // init_yaw, :358 (degrees)
    int lazy_target_margin = 0;            // > 0: the map's covariances / voxels only within this many voxels of where the sweep falls at the guess (rgc_set_target_lazy: same poses)
  };

  explicit OdometryNode(const Options& o) : opt_(o) {
    rgc_params p;
    rgc_default_params(&p);                // = the setters of RGC_odometer.cpp:998-1006 (resolution 1.0, 25 iterations, eps 1e-6)
    int rc = rgc_create(o.hip_device, &p, &ctx_);
    if (rc != RGC_OK) throw std::runtime_error(std::string("rgc_create: ") + rgc_status_string(rc));
    if (o.lazy_target_margin > 0 && (rc = rgc_set_target_lazy(ctx_, o.lazy_target_margin)) != RGC_OK)
      throw std::runtime_error(std::string("rgc_set_target_lazy: ") + rgc_last_error(ctx_));
    rgc_default_fe_params(&fe_);
    fe_.n_scans = o.scan_line; fe_.min_range = o.minimum_range; fe_.max_range = o.maxmum_range;
    rgc_imu_filter_init(&imu_);
    rgc_ground_gate_init(&gate_);
    const double ril[3] = {-1.29, -0.15, 0.65};   // R_il, :387 (degrees)
    rgc_ypr2R(ril, R_il_);
  }

  // the /mynteye/imu/data_raw subscriber (vg_ICP::imu_callback, :444-486): attitude filter + the buffer the frame body integrates
  void imuCallback(double stamp, const double acc[3], const double gyr[3]) {
    ImuSample s;
    s.t = stamp;
    if (rgc_imu_filter_push(&imu_, stamp, acc, gyr, s.acc, s.gyr) == 1) imu_buf_.push_back(s);
  }
  int groundFlag() const { return gate_.gflag; }
  ~OdometryNode() {
    for (DevBuf* b : {&d_raw_, &d_source_, &d_last_, &d_submap_, &d_target_}) if (b->p) rgc_device_free(ctx_, b->p);
    for (auto& k : d_kf_) if (k.p) rgc_device_free(ctx_, k.p);
    for (auto& k : d_kf_free_) if (k.p) rgc_device_free(ctx_, k.p);
    rgc_destroy(ctx_);
  }
  OdometryNode(const OdometryNode&) = delete;
  OdometryNode& operator=(const OdometryNode&) = delete;

  // the /velodyne_points subscriber on the raw message bytes (pcl::fromROSMsg becomes a kernel, scanRegistration.cpp:107-108)
  void handlePointCloud2(const void* data, int n_points, const rgc_pc2_layout& layout, double stamp, OdometryMsg* odom, GroundMsg* ground) {
    if (opt_.device_chain) {
      reserve(d_raw_, (size_t)16 * (size_t)(n_points > 0 ? n_points : 1));
      chk(rgc_pc2_unpack(ctx_, data, n_points, &layout, d_raw_.p, nullptr, nullptr, 1));
      process(d_raw_.p, n_points, 16, true, stamp, odom, ground);
      return;
    }
    fit(raw_, (size_t)4 * (size_t)(n_points > 0 ? n_points : 1));
    chk(rgc_pc2_unpack(ctx_, data, n_points, &layout, raw_.data(), nullptr, nullptr, 0));
    process(raw_.data(), n_points, 16, false, stamp, odom, ground);
  }

  // the same on an already converted cloud: x,y,z,intensity in firing order
  void handleCloud(const float* xyzi, int n, int stride_bytes, double stamp, OdometryMsg* odom, GroundMsg* ground) {
    if (opt_.device_chain && n > 0) {
      reserve(d_raw_, (size_t)stride_bytes * (size_t)n);
      chk(rgc_upload(ctx_, d_raw_.p, xyzi, (size_t)stride_bytes * (size_t)n));
      process(d_raw_.p, n, stride_bytes, true, stamp, odom, ground);
      return;
    }
    process(xyzi, n, stride_bytes, false, stamp, odom, ground);
  }

  // the frame body alone (device_chain): the sweep already went through a front-end -- on another context / thread, see
  // ReplayPipeline -- and lies on this GPU, ring-major with the encoded intensity; it is de-skewed in place
  void handleFrontEndOutput(float* d_full, int n_full, const double groundparam[11], bool ground_valid, double stamp, OdometryMsg* odom) {
    if (!opt_.device_chain) throw std::runtime_error("handleFrontEndOutput needs device_chain");
    if (!begin_frame(stamp)) { publish(stamp, odom); return; }
    body(d_full, n_full, groundparam, ground_valid, stamp, odom);
  }

  // what the node publishes besides the odometry (:689-727): feature clouds of the last sweep, x,y,z,intensity,normal_x
  const float* cornerPointsSharp(int* n) const { *n = n_sharp_; return sharp_.data(); }
  const float* surfPointsFlat(int* n) const { *n = n_flat_; return flat_.data(); }
  int frames() const { return frames_; }
  int keyframesInserted() const { return kf_inserted_; }
  rgc_ctx* context() { return ctx_; }

private:
  struct DevBuf { float* p = nullptr; size_t cap = 0; };
  // Host staging vectors live in PAGE-LOCKED memory (rgc_host_alloc) and grow with headroom: a copy between the device and pageable
  // memory is staged by the runtime, and the first one out of fresh pages cost 7 ms (round 3's one 8.8 ms frame of the host-staged modes).
  static void fit(HostVec& v, size_t n) {
    if (n > v.capacity()) v.reserve(n + n / 2);
    v.resize(n);
  }
  void reserve(DevBuf& b, size_t bytes) {
    if (bytes <= b.cap) return;
    if (b.p) chk(rgc_device_free(ctx_, b.p));
    b.p = nullptr; b.cap = 0;
    void* np = nullptr;
    chk(rgc_device_alloc(ctx_, bytes + bytes / 4, &np));
    b.p = (float*)np; b.cap = bytes + bytes / 4;
  }

  // in: host memory, or (on_device) device memory of this context's GPU
  void process(const float* xyzi, int n, int stride_bytes, bool on_device, double stamp, OdometryMsg* odom, GroundMsg* ground) {
    const bool chain = opt_.device_chain;
    if (!begin_frame(stamp)) { publish(stamp, odom); return; }
    // ---- ScanRegistration::laserCloudHandler ----
    const int fcap = fe_.n_scans * 6 * 41;
    fit(full_, (size_t)4 * (size_t)(n > 0 ? n : 1));
    sharp_.resize((size_t)5 * fcap); flat_.resize((size_t)5 * fcap); inten_.resize((size_t)5 * fcap);
    rgc_fe_out fo;
    std::memset(&fo, 0, sizeof(fo));
    fo.cloud = chain ? nullptr : full_.data(); fo.cloud_cap = n > 0 ? n : 1;
    fo.sharp = sharp_.data(); fo.flat = flat_.data(); fo.inten = inten_.data(); fo.feat_cap = fcap;
    chk(on_device ? rgc_frontend_device(ctx_, xyzi, n, stride_bytes, &fe_, &fo) : rgc_frontend(ctx_, xyzi, n, stride_bytes, &fe_, &fo));
    const int n_full = fo.n_cloud;
    float* d_full = nullptr;   // chain: the ring-major sweep where the front-end left it
    if (chain) { int nn = 0; chk(rgc_frontend_cloud_device(ctx_, &d_full, &nn)); }
    n_sharp_ = fo.n_sharp; n_flat_ = fo.n_flat;
    if (ground) { std::memcpy(ground->param, fo.groundparam, sizeof(ground->param)); ground->valid = fo.ground_valid != 0; }
    body(d_full, n_full, fo.groundparam, fo.ground_valid != 0, stamp, odom);
  }

  // ---- vg_ICP::ICP_thread: one frame body on a sweep that has been through the front-end (d_full: its device copy when chained,
  // else full_ holds it on the host) ----
  void body(float* d_full, int n_full, const double groundparam[11], bool ground_valid, double stamp, OdometryMsg* odom) {
    const bool chain = opt_.device_chain;
    if (n_full > 0) chk(rgc_deskew(ctx_, chain ? d_full : full_.data(), n_full, 16, q_last_curr_, t_last_curr_, chain ? 1 : 0));   // adjustDistortion, :958
    if (have_last_) {
      if (submapflag_ == 0) first_keyframe();                                                   // :963-972
      submapflag_++;
      int n_src = 0;
      bool source_set = false;
      if (chain) {
        reserve(d_source_, (size_t)16 * (size_t)(n_full > 0 ? n_full : 1));
        chk(rgc_voxelgrid(ctx_, d_full, n_full, 16, 0.2f, d_source_.p, &n_src, 1));
        // (reference semantics: the scan is prepared on its own stream while the sub-map goes through its leaf filter -- it depends
        // on neither; with the resident map the commit is usually a no-op and the order makes no difference)
        if (!opt_.resident_map) { chk(rgc_set_source_device(ctx_, d_source_.p, n_src, 16)); source_set = true; }   // :1008
      } else {
        fit(source_, (size_t)4 * n_full);
        chk(rgc_voxelgrid(ctx_, full_.data(), n_full, 16, 0.2f, source_.data(), &n_src, 0));     // :976-983, planeResolution1
      }
      float guess[16], T[16];
      double fitness = 1.0;
      if (opt_.resident_map) {
        // the guess of :993-996 moved into the map frame: T_w_curr * T_last_curr
        double qg[4], tg[3];
        qmul(q_w_, q_last_curr_, qg); qnormalize(qg);
        qrot(q_w_, t_last_curr_, tg);
        for (int a = 0; a < 3; a++) tg[a] += t_w_[a] - origin_[a];
        pose_to_mat(qg, tg, guess);
        chk(rgc_map_commit(ctx_, 0.3f, nullptr));                                              // :985-991, 1007 (only if a keyframe changed)
      } else if (chain) {   // the reference's local map, kept on the device: the re-framed keyframes -> leaf filter -> target, no PCIe crossing
        int n_tgt = 0;
        if (target_begun_) {   // the sub-map's filter was started when the previous frame ended (maintain_map): only its count is fetched here
          chk(rgc_voxelgrid_end(ctx_, &n_tgt));                                                  // :985-991
          target_begun_ = false;
        } else {
          reserve(d_target_, (size_t)16 * (size_t)(n_submap_ > 0 ? n_submap_ : 1));
          chk(rgc_voxelgrid(ctx_, d_submap_.p, n_submap_, 16, 0.3f, d_target_.p, &n_tgt, 1));     // :985-991
        }
        pose_to_mat(q_last_curr_, t_last_curr_, guess);                                         // :993-996
        chk(rgc_set_target_device(ctx_, d_target_.p, n_tgt, 16));                               // :1007
      } else {
        int n_tgt = 0;
        fit(target_, submap_.size());
        chk(rgc_voxelgrid(ctx_, submap_.data(), (int)(submap_.size() / 4), 16, 0.3f, target_.data(), &n_tgt, 0));   // :985-991
        pose_to_mat(q_last_curr_, t_last_curr_, guess);                                         // :993-996
        chk(rgc_set_target(ctx_, target_.data(), n_tgt, 16));                                   // :1007
      }
      if (!source_set) chk(chain ? rgc_set_source_device(ctx_, d_source_.p, n_src, 16) : rgc_set_source(ctx_, source_.data(), n_src, 16));   // :1008
      int it = 0, conv = 0, lmf = 0;
      chk(rgc_align(ctx_, guess, T, nullptr, &fitness, &it, &conv, &lmf));                       // :1009-1010
      double q_l[4], t_l[3];
      chk(rgc_extract_pose(T, q_l, t_l));                                                       // :1011-1016
      if (opt_.resident_map) {   // T is the scan's map-frame pose: back to the delta the fusion expects, T_w_curr^-1 * T
        double qi[4] = {-q_w_[0], -q_w_[1], -q_w_[2], q_w_[3]}, qd[4], d[3];
        qmul(qi, q_l, qd);
        for (int a = 0; a < 3; a++) d[a] = t_l[a] + origin_[a] - t_w_[a];
        qrot(qi, d, t_l);
        std::memcpy(q_l, qd, sizeof(qd));
      }
      // ground-change detector + pose fusion + composition + gravity blend, :1025-1214
      const bool have_ground = ground_valid && have_ground_last_;
      rgc_fuse_in fin;
      rgc_default_fuse_in(&fin);
      std::memcpy(fin.q_lidar_xyzw, q_l, sizeof(q_l)); std::memcpy(fin.t_lidar, t_l, sizeof(t_l));
      fin.fitness = fitness;
      const int gflag = rgc_ground_gate_step(&gate_, have_ground ? ground_last_ : nullptr, have_ground ? groundparam : nullptr, q_l, t_l,
                                             have_dq_imu_ ? dq_imu_ : nullptr, q_w_, fin.q_w_curr_f_xyzw);   // :1034-1087
      if (gflag < 0) chk(gflag);
      const bool use_ground = opt_.use_ground && have_ground && gflag == 0;                     // :1088
      fin.use_ground = use_ground ? 1 : 0;
      if (use_ground) {
        std::memcpy(fin.ground_last, ground_last_, sizeof(ground_last_));
        std::memcpy(fin.ground_cur, groundparam, sizeof(ground_last_));
      }
      if (opt_.use_imu && have_dq_imu_) { fin.use_imu = 1; std::memcpy(fin.q_imu_xyzw, dq_imu_, sizeof(dq_imu_)); }   // :1104-1119
      double q_f[4], t_f[3], q_new[4], t_new[3], t_lc[3], R_imu[9];
      chk(rgc_fuse_pose(&fin, q_f, t_f, nullptr));
      if (opt_.use_imu) matmul3(imu_.Rwi, R_il_, R_imu);                                         // IMU.Rwi * R_il, :1209
      chk(rgc_compose_pose(q_w_, t_w_, q_f, t_f, t_l, opt_.use_imu ? 1 : 0, opt_.use_imu ? R_imu : nullptr, q_new, t_new, t_lc));   // :1194-1214
      std::memcpy(q_w_, q_new, sizeof(q_new)); std::memcpy(t_w_, t_new, sizeof(t_new));
      std::memcpy(q_last_curr_, q_f, sizeof(q_f)); std::memcpy(t_last_curr_, t_lc, sizeof(t_lc));
      maintain_map(n_src);                                                                      // :1218-1256
    }
    // :1319-1322 -- the previous sweep is only ever read to make keyframe 0
    if (chain) {
      if (submapflag_ == 0 && n_full > 0) {
        const double I[4] = {0, 0, 0, 1}, Z[3] = {0, 0, 0};
        reserve(d_last_, (size_t)16 * n_full);
        chk(rgc_transform_cloud(ctx_, d_full, n_full, 16, I, Z, d_last_.p, 1));   // identity = a device-to-device copy
        n_last_ = n_full;
      }
    } else {
      full_last_.assign(full_.begin(), full_.begin() + (size_t)4 * n_full);
      n_last_ = n_full;
    }
    have_last_ = n_full > 0;
    if (ground_valid) { std::memcpy(ground_last_, groundparam, sizeof(ground_last_)); have_ground_last_ = true; }
    frames_++;
    publish(stamp, odom);
  }

  void publish(double stamp, OdometryMsg* odom) const {
    if (!odom) return;
    odom->stamp = stamp;
    std::memcpy(odom->position, t_w_, sizeof(t_w_));
    std::memcpy(odom->orientation_xyzw, q_w_, sizeof(q_w_));
  }

  // :857-931, 955-956: the first sweeps only initialise the pose (from IMU.Rwi * R_il with the IMU); afterwards the gyro samples
  // between the previous sweep and this one give delta_q_imu = q_last_curr, the registration's rotation guess.  false = the sweep is
  // dropped (initialisation, or no IMU coverage: getIMUInterval, :1376-1416).
  bool begin_frame(double stamp) {
    if (frames_seen_ < opt_.first_frames) {
      frames_seen_++;
      prev_time_ = stamp;
      t_w_[0] = t_w_[1] = t_w_[2] = 0.0;
      if (opt_.use_imu) {
        double R[9], ypr[3];
        matmul3(imu_.Rwi, R_il_, R);
        rgc_R2ypr(R, ypr);
        ypr[0] += opt_.init_yaw;
        rgc_ypr2R(ypr, R);
        R2q(R, q_w_);
      }
      return false;
    }
    if (opt_.use_imu) {
      if (imu_buf_.empty() || (prev_time_ <= imu_buf_.front().t && stamp <= imu_buf_.front().t) || !(stamp <= imu_buf_.back().t)) return false;
      while (imu_buf_.front().t <= prev_time_) imu_buf_.pop_front();
      st_.clear(); ga_.clear(); aa_.clear();
      auto take = [&](const ImuSample& m) { st_.push_back(m.t); for (int a = 0; a < 3; a++) { ga_.push_back(m.gyr[a]); aa_.push_back(m.acc[a]); } };
      while (imu_buf_.front().t < stamp) { take(imu_buf_.front()); imu_buf_.pop_front(); }
      take(imu_buf_.front());
      chk(rgc_imu_preintegrate(st_.data(), ga_.data(), aa_.data(), (int)st_.size(), prev_time_, stamp, dq_imu_, nullptr, nullptr, nullptr));
      have_dq_imu_ = true;
      std::memcpy(q_last_curr_, dq_imu_, sizeof(dq_imu_));                                       // :929-930
    }
    prev_time_ = stamp;
    frames_seen_++;
    return true;
  }

  static void matmul3(const double A[9], const double B[9], double C[9]) {
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) C[r * 3 + c] = A[r * 3] * B[c] + A[r * 3 + 1] * B[3 + c] + A[r * 3 + 2] * B[6 + c];
  }
  static void R2q(const double R[9], double q[4]) {   // positive-trace branch (attitudes near the identity)
    const double w = std::sqrt(std::fmax(0.0, 1.0 + R[0] + R[4] + R[8])) / 2;
    q[0] = (R[7] - R[5]) / (4 * w); q[1] = (R[2] - R[6]) / (4 * w); q[2] = (R[3] - R[1]) / (4 * w); q[3] = w;
    qnormalize(q);
  }

  void chk(int rc) { if (rc != RGC_OK) throw std::runtime_error(std::string(rgc_status_string(rc)) + ": " + rgc_last_error(ctx_)); }

  static void qmul(const double a[4], const double b[4], double o[4]) {
    const double x = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1], y = a[3] * b[1] - a[0] * b[2] + a[1] * b[3] + a[2] * b[0];
    const double z = a[3] * b[2] + a[0] * b[1] - a[1] * b[0] + a[2] * b[3], w = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
    o[0] = x; o[1] = y; o[2] = z; o[3] = w;
  }
  static void qnormalize(double q[4]) {
    const double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    for (int a = 0; a < 4; a++) q[a] /= n;
  }
  static void qrot(const double q[4], const double v[3], double o[3]) {
    const double ux = 2 * (q[1] * v[2] - q[2] * v[1]), uy = 2 * (q[2] * v[0] - q[0] * v[2]), uz = 2 * (q[0] * v[1] - q[1] * v[0]);
    const double x = v[0] + q[3] * ux + (q[1] * uz - q[2] * uy), y = v[1] + q[3] * uy + (q[2] * ux - q[0] * uz);
    const double z = v[2] + q[3] * uz + (q[0] * uy - q[1] * ux);
    o[0] = x; o[1] = y; o[2] = z;
  }
  static void q2R(const double q[4], double R[9]) {
    const double x = q[0], y = q[1], z = q[2], w = q[3];
    R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - z * w); R[2] = 2 * (x * z + y * w);
    R[3] = 2 * (x * y + z * w); R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - x * w);
    R[6] = 2 * (x * z - y * w); R[7] = 2 * (y * z + x * w); R[8] = 1 - 2 * (x * x + y * y);
  }
  static void pose_to_mat(const double q[4], const double t[3], float M[16]) {   // Matrix4f from Quaterniond / Vector3d
    double R[9];
    q2R(q, R);
    for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) M[r * 4 + c] = (float)R[r * 3 + c]; M[r * 4 + 3] = (float)t[r]; }
    M[12] = M[13] = M[14] = 0.f; M[15] = 1.f;
  }

  void first_keyframe() {   // the previous sweep is keyframe 0 at the identity pose, :963-972
    const double I[4] = {0, 0, 0, 1}, Z[3] = {0, 0, 0};
    if (opt_.resident_map) {
      std::memcpy(origin_, t_w_, sizeof(origin_));
      chk(rgc_map_reset(ctx_, origin_));
      if (opt_.device_chain) chk(rgc_map_insert(ctx_, d_last_.p, n_last_, 16, I, Z, 1, nullptr));
      else chk(rgc_map_insert(ctx_, full_last_.data(), n_last_, 16, I, Z, 0, nullptr));
    } else if (opt_.device_chain) {   // *laserCloudsubmap += *laserCloudFullLast, on the device (an identity transform is a copy)
      DevKf kf = new_keyframe(n_last_);
      chk(rgc_transform_cloud(ctx_, d_last_.p, n_last_, 16, I, Z, kf.p, 1));
      d_kf_.push_back(kf);
      reserve(d_submap_, (size_t)16 * (size_t)n_last_);
      chk(rgc_transform_cloud(ctx_, d_last_.p, n_last_, 16, I, Z, d_submap_.p, 1));
      n_submap_ = n_last_;
    } else {
      kf_cloud_.push_back(full_last_);
      submap_.insert(submap_.end(), full_last_.begin(), full_last_.end());
    }
    std::memcpy(kf_q_, I, sizeof(I)); std::memcpy(kf_t_, Z, sizeof(Z));
    have_kf_ = true;
    rgc_ground_gate_remember(&gate_);   // histoary_pose.push_back(q_w_curr_delta), :969
  }

  void maintain_map(int n_src) {
    if (have_kf_) {
      double Rb[9], Rc[9], yb[3], yc[3];
      q2R(kf_q_, Rb); q2R(q_w_, Rc);
      rgc_R2ypr(Rb, yb); rgc_R2ypr(Rc, yc);
      const float dx = (float)(kf_t_[0] - t_w_[0]), dy = (float)(kf_t_[1] - t_w_[1]), dz = (float)(kf_t_[2] - t_w_[2]);
      float dyaw = (float)(yb[0] - yc[0]);
      const float dpitch = (float)(yb[1] - yc[1]), droll = (float)(yb[2] - yc[2]);
      if (dyaw > M_PI) dyaw = (float)(dyaw - M_PI * 2);
      if (dyaw < -M_PI) dyaw = (float)(dyaw + M_PI * 2);
      const float kAngle = 0.2f, kDist = 0.3f;   // keyframeAddingAngle / keyframeAddingDistance, :280-281 (the angle meets DEGREES)
      if (std::fabs(droll) > kAngle || std::fabs(dpitch) > kAngle || std::fabs(dyaw) > kAngle || std::sqrt(dx * dx + dy * dy + dz * dz) > kDist ||
          submapflag_ < kSlipwide - 1) {
        if (opt_.resident_map) {
          chk(opt_.device_chain ? rgc_map_insert(ctx_, d_source_.p, n_src, 16, q_w_, t_w_, 1, nullptr)
                                : rgc_map_insert(ctx_, source_.data(), n_src, 16, q_w_, t_w_, 0, nullptr));   // :1237, once, never re-framed
          chk(rgc_map_evict(ctx_, opt_.max_keyframes, opt_.evict_radius > 0 ? t_w_ : nullptr, opt_.evict_radius, nullptr));   // :1242-1247
        } else if (opt_.device_chain) {
          DevKf kf = new_keyframe(n_src);
          chk(rgc_transform_cloud(ctx_, d_source_.p, n_src, 16, q_w_, t_w_, kf.p, 1));           // :1237
          d_kf_.push_back(kf);
        } else {
          // (a recycled buffer of an evicted keyframe: a fresh page-locked vector is a hipHostMalloc inside the frame, and freeing one when
          // the window slides synchronises the device -- the kind of stall the page-locked staging was introduced to remove)
          HostVec w;
          if (!kf_free_.empty()) { w = std::move(kf_free_.back()); kf_free_.pop_back(); }
          fit(w, (size_t)4 * n_src);
          chk(rgc_transform_cloud(ctx_, source_.data(), n_src, 16, q_w_, t_w_, w.data(), 0));
          kf_cloud_.push_back(std::move(w));
        }
        std::memcpy(kf_q_, q_w_, sizeof(kf_q_)); std::memcpy(kf_t_, t_w_, sizeof(kf_t_));
        kf_inserted_++;
      }
    }
    if (opt_.resident_map) {
      const double ddx = t_w_[0] - origin_[0], ddy = t_w_[1] - origin_[1], ddz = t_w_[2] - origin_[2];
      if (std::sqrt(ddx * ddx + ddy * ddy + ddz * ddz) > opt_.rebase_distance) {
        std::memcpy(origin_, t_w_, sizeof(origin_));
        chk(rgc_map_rebase(ctx_, origin_));
      }
      return;
    }
    if (opt_.device_chain) {   // :1241-1256 on the device: the window slides, every keyframe is re-expressed in the new body frame
      n_submap_ = 0;
      if ((int)d_kf_.size() > opt_.max_keyframes) { d_kf_free_.push_back(d_kf_.front()); d_kf_.pop_front(); }
      if (d_kf_.size() > 1) {
        const double qi[4] = {-q_w_[0], -q_w_[1], -q_w_[2], q_w_[3]};
        double ti[3];
        qrot(qi, t_w_, ti);
        for (int a = 0; a < 3; a++) ti[a] = -ti[a];
        size_t total = 0, largest = 0;
        for (const auto& kf : d_kf_) { total += (size_t)kf.n; largest = std::max(largest, (size_t)kf.n); }
        // (sized for a full window of keyframes at once: growing it when the third keyframe arrives is a hipFree + hipMalloc, several ms
        // in whichever frame that is)
        const size_t window = std::max(total, (size_t)opt_.max_keyframes * largest);
        reserve(d_submap_, (size_t)16 * (window > 0 ? window : 1));
        reserve(d_target_, (size_t)16 * (window > 0 ? window : 1));
        for (const auto& kf : d_kf_) {
          chk(rgc_transform_cloud(ctx_, kf.p, kf.n, 16, qi, ti, d_submap_.p + (size_t)4 * (size_t)n_submap_, 1));
          n_submap_ += kf.n;
        }
      }
      // The next frame's target is this sub-map through the 0.3 m leaf filter (:985-991), which depends on nothing of the next sweep:
      // it is started now -- the filter's kernels run while the host returns and the next message goes up -- and its count is fetched
      // behind the next sweep's own filter (rgc_voxelgrid_begin / _end): 55 us of kernels and a read-back off the frame's critical path.
      if (n_submap_ > 0) {
        reserve(d_target_, (size_t)16 * (size_t)n_submap_);
        chk(rgc_voxelgrid_begin(ctx_, d_submap_.p, n_submap_, 16, 0.3f, d_target_.p));
        target_begun_ = true;
      }
      return;
    }
    submap_.clear();
    if ((int)kf_cloud_.size() > opt_.max_keyframes) { kf_free_.push_back(std::move(kf_cloud_.front())); kf_cloud_.pop_front(); }   // :1242-1247
    if (kf_cloud_.size() > 1) {                                                                  // :1248-1256: every keyframe into the new body frame
      const double qi[4] = {-q_w_[0], -q_w_[1], -q_w_[2], q_w_[3]};
      double ti[3];
      qrot(qi, t_w_, ti);
      for (int a = 0; a < 3; a++) ti[a] = -ti[a];
      for (const auto& kf : kf_cloud_) {
        const size_t at = submap_.size();
        fit(submap_, at + kf.size());
        chk(rgc_transform_cloud(ctx_, kf.data(), (int)(kf.size() / 4), 16, qi, ti, submap_.data() + at, 0));
      }
    }
  }

  static constexpr int kSlipwide = 3;   // slipwide, RGC_odometer.cpp:299 (the first keyframes are forced, :1233)
  Options opt_;
  rgc_ctx* ctx_ = nullptr;
  rgc_fe_params fe_{};
  HostVec raw_, full_, full_last_, sharp_, flat_, inten_, source_, target_, submap_;
  std::deque<HostVec> kf_cloud_;
  std::vector<HostVec> kf_free_;   // buffers of evicted keyframes, taken again by the next one
  DevBuf d_raw_, d_source_, d_last_;     // device_chain: the unpacked message, the 0.2 m-filtered sweep, the previous sweep (keyframe 0)
  // device_chain without the resident map: the reference's keyframe window (world-frame clouds) and its re-framed concatenation on the device
  struct DevKf { float* p = nullptr; size_t cap = 0; int n = 0; };
  std::deque<DevKf> d_kf_;
  std::vector<DevKf> d_kf_free_;         // buffers of keyframes that left the window, re-used
  DevBuf d_submap_, d_target_;
  int n_submap_ = 0;
  bool target_begun_ = false;   // rgc_voxelgrid_begin(d_submap_ -> d_target_) is open
  DevKf new_keyframe(int n) {
    const size_t need = (size_t)16 * (size_t)(n > 0 ? n : 1);
    for (size_t i = 0; i < d_kf_free_.size(); i++)
      if (d_kf_free_[i].cap >= need) { DevKf k = d_kf_free_[i]; d_kf_free_.erase(d_kf_free_.begin() + (long)i); k.n = n; return k; }
    DevKf k;
    void* np = nullptr;
    chk(rgc_device_alloc(ctx_, need + need / 4, &np));
    k.p = (float*)np; k.cap = need + need / 4; k.n = n;
    return k;
  }
  int n_last_ = 0;
  bool have_last_ = false;
  int n_sharp_ = 0, n_flat_ = 0;
  double q_w_[4] = {0, 0, 0, 1}, t_w_[3] = {0, 0, 0};                 // q_w_curr, t_w_curr
  double q_last_curr_[4] = {0, 0, 0, 1}, t_last_curr_[3] = {0, 0, 0};  // para_q, para_t
  rgc_ground_gate gate_;                                               // gflag, changegroundflag, q_w_curr_delta, histoary_pose (:1034-1087)
  rgc_imu_filter imu_;                                                 // IMU (imu_s) + ComplementaryFilter state
  struct ImuSample { double t, acc[3], gyr[3]; };
  std::deque<ImuSample> imu_buf_;                                      // accBuf / gyrBuf
  std::vector<double> st_, ga_, aa_;                                   // the interval handed to rgc_imu_preintegrate
  double R_il_[9];
  double dq_imu_[4] = {0, 0, 0, 1};                                    // delta_q_imu
  bool have_dq_imu_ = false;
  double prev_time_ = 0.0;
  int frames_seen_ = 0;
  double ground_last_[11] = {0};
  bool have_ground_last_ = false, have_kf_ = false;
  double kf_q_[4] = {0, 0, 0, 1}, kf_t_[3] = {0, 0, 0}, origin_[3] = {0, 0, 0};
  int submapflag_ = 0, frames_ = 0, kf_inserted_ = 0;
};

// Bag replay at full rate: the reference runs scanRegistration and the odometry as two ROS nodes, i.e. concurrently.  Here the
// front-end of sweep k+1 (own context, own stream, own host thread) overlaps the frame body of sweep k; the sweep is handed over
// on the device through two slots.  Poses are those of the unpipelined node (the stages see the same data in the same order).
class ReplayPipeline {
public:
  // front_workers: how many sweeps are in the front-end at once (each worker = one host thread + one context; workers take alternate
  // sweeps, the frame body consumes them in order).  The front-end of a sweep (0.56 ms, two host round trips) is longer than the frame
  // body and sweeps are independent there -- yet MEASURED on MI355X / ROCm 7.2 a second worker makes the replay slower (0.65 ms per
  // sweep with one, 0.98 with two, 0.85 with three: three host threads that synchronise with the device several times per sweep
  // contend inside the runtime), so the default is one.
  explicit ReplayPipeline(OdometryNode::Options o, int front_workers = 1)
      : body_((o.resident_map = true, o.device_chain = true, o)), opt_(o), front_(front_workers < 1 ? 1 : front_workers), slot_(2 * front_.size()) {
    for (Front& f : front_) {
      int rc = rgc_create(o.hip_device, nullptr, &f.ctx);
      if (rc != RGC_OK) throw std::runtime_error(std::string("rgc_create: ") + rgc_status_string(rc));
    }
    rgc_default_fe_params(&fe_);
    fe_.n_scans = o.scan_line; fe_.min_range = o.minimum_range; fe_.max_range = o.maxmum_range;
  }
  // IMU messages go to the frame body's node (deliver them before run(): the replay has every message up front)
  void imuCallback(double stamp, const double acc[3], const double gyr[3]) { body_.imuCallback(stamp, acc, gyr); }
  ~ReplayPipeline() {
    for (Slot& s : slot_) if (s.d) rgc_device_free(front_[0].ctx, s.d);
    for (Front& f : front_) {
      if (f.d_raw) rgc_device_free(f.ctx, f.d_raw);
      if (f.ctx) rgc_destroy(f.ctx);
    }
  }
  ReplayPipeline(const ReplayPipeline&) = delete;
  ReplayPipeline& operator=(const ReplayPipeline&) = delete;

  // messages[k]: n_points[k] records of `layout`; stamps[k] their time stamps.  Fills one pose and one ground message per sweep.
  // done_ms (nullable): wall time, from the call, at which each pose was ready.
  void run(const std::vector<const void*>& messages, const std::vector<int>& n_points, const rgc_pc2_layout& layout, const std::vector<double>& stamps,
           std::vector<OdometryMsg>* odom, std::vector<GroundMsg>* ground, std::vector<double>* done_ms = nullptr) {
    const size_t N = messages.size(), F = front_.size(), NS = slot_.size();
    const auto t_call = std::chrono::steady_clock::now();
    if (done_ms) done_ms->assign(N, 0.0);
    odom->assign(N, OdometryMsg());
    ground->assign(N, GroundMsg());
    stop_ = false;                                   // a run that failed must not poison the next one
    for (Slot& s : slot_) s.full = false;
    std::exception_ptr front_error;
    std::vector<std::thread> workers;
    for (size_t w = 0; w < F; w++)
      workers.emplace_back([&, w]() {
        try {
          for (size_t k = w; k < N; k += F) {        // sweep k always lands in slot k % NS: the body takes the slots in sweep order
            Slot& s = slot_[k % NS];
            { std::unique_lock<std::mutex> lk(m_); cv_.wait(lk, [&] { return !s.full || stop_; }); if (stop_) return; }
            front_stage(front_[w], messages[k], n_points[k], layout, s);
            { std::lock_guard<std::mutex> lk(m_); s.full = true; }
            cv_.notify_all();
          }
        } catch (...) {
          { std::lock_guard<std::mutex> lk(m_); if (!front_error) front_error = std::current_exception(); stop_ = true; }
          cv_.notify_all();
        }
      });
    auto join_all = [&]() { for (std::thread& t : workers) t.join(); };
    try {
      for (size_t k = 0; k < N; k++) {
        Slot& s = slot_[k % NS];
        { std::unique_lock<std::mutex> lk(m_); cv_.wait(lk, [&] { return s.full || stop_; }); if (stop_ && !s.full) break; }
        body_.handleFrontEndOutput(s.d, s.n, s.ground.param, s.ground.valid, stamps[k], &(*odom)[k]);
        (*ground)[k] = s.ground;
        if (done_ms) (*done_ms)[k] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_call).count();
        { std::lock_guard<std::mutex> lk(m_); s.full = false; }
        cv_.notify_all();
      }
    } catch (...) {
      { std::lock_guard<std::mutex> lk(m_); stop_ = true; }
      cv_.notify_all();
      join_all();
      throw;
    }
    join_all();
    if (front_error) std::rethrow_exception(front_error);
  }
  OdometryNode& node() { return body_; }

private:
  struct Slot { float* d = nullptr; size_t cap = 0; int n = 0; GroundMsg ground; bool full = false; };
  struct Front {                                     // one front-end worker: its context, its raw-sweep buffer, its feature staging
    rgc_ctx* ctx = nullptr;
    float* d_raw = nullptr; size_t raw_cap = 0;
    HostVec sharp, flat, inten;
  };
  static void chk(rgc_ctx* c, int rc) { if (rc != RGC_OK) throw std::runtime_error(std::string(rgc_status_string(rc)) + ": " + rgc_last_error(c)); }
  static void grow(rgc_ctx* c, float*& p, size_t& cap, size_t bytes) {
    if (bytes <= cap) return;
    if (p) chk(c, rgc_device_free(c, p));
    p = nullptr; cap = 0;
    void* np = nullptr;
    chk(c, rgc_device_alloc(c, bytes + bytes / 4, &np));
    p = (float*)np; cap = bytes + bytes / 4;
  }
  void front_stage(Front& f, const void* data, int n, const rgc_pc2_layout& layout, Slot& s) {   // ScanRegistration::laserCloudHandler
    grow(f.ctx, f.d_raw, f.raw_cap, (size_t)16 * (size_t)(n > 0 ? n : 1));
    chk(f.ctx, rgc_pc2_unpack(f.ctx, data, n, &layout, f.d_raw, nullptr, nullptr, 1));
    const int fcap = fe_.n_scans * 6 * 41;
    f.sharp.resize((size_t)5 * fcap); f.flat.resize((size_t)5 * fcap); f.inten.resize((size_t)5 * fcap);
    rgc_fe_out fo;
    std::memset(&fo, 0, sizeof(fo));
    fo.cloud = nullptr; fo.cloud_cap = n > 0 ? n : 1;
    fo.sharp = f.sharp.data(); fo.flat = f.flat.data(); fo.inten = f.inten.data(); fo.feat_cap = fcap;
    chk(f.ctx, rgc_frontend_device(f.ctx, f.d_raw, n, 16, &fe_, &fo));
    float* d_full = nullptr; int nn = 0;
    chk(f.ctx, rgc_frontend_cloud_device(f.ctx, &d_full, &nn));
    s.n = fo.n_cloud;
    if (s.n > 0) {
      const double I[4] = {0, 0, 0, 1}, Z[3] = {0, 0, 0};
      grow(f.ctx, s.d, s.cap, (size_t)16 * (size_t)s.n);   // (device memory is not tied to the context that allocated it)
      chk(f.ctx, rgc_transform_cloud(f.ctx, d_full, s.n, 16, I, Z, s.d, 1));   // identity = device-to-device copy into the hand-over slot
    }
    std::memcpy(s.ground.param, fo.groundparam, sizeof(s.ground.param));
    s.ground.valid = fo.ground_valid != 0;
  }

  OdometryNode body_;
  OdometryNode::Options opt_;
  std::vector<Front> front_;
  rgc_fe_params fe_{};
  std::vector<Slot> slot_;
  std::mutex m_;
  std::condition_variable cv_;
  bool stop_ = false;
};

}  // namespace rgc
