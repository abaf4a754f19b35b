// rgc_host.cpp -- scalar host stages of the odometer's per-frame body that stay on the CPU (a handful of flops each;
// SURVEY.md §7.1 step 7): IMU rotation pre-integration (B1), the pose-fusion solve the reference hands to Ceres (B7)
// and the pose composition + gravity blend (B8).  Part of librgc_hip.so (C-ABI in include/rgc_hip.h); needs no GPU.
// Reference citations are relative to /root/reference/rgc_slam/.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>

#include "../../include/rgc_hip.h"

namespace {

struct Q { double x, y, z, w; };

inline Q qmul(const Q& a, const Q& b) {  // Hamilton product a (x) b
  return Q{a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y, a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x,
           a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w, a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z};
}
inline Q qconj(const Q& a) { return Q{-a.x, -a.y, -a.z, a.w}; }
inline Q qnormalized(const Q& a) {
  const double n = std::sqrt(a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w);
  return n > 0 ? Q{a.x / n, a.y / n, a.z / n, a.w / n} : a;
}
inline void qrot(const Q& q, const double v[3], double o[3]) {  // Eigen QuaternionBase::_transformVector
  double ux = q.y * v[2] - q.z * v[1], uy = q.z * v[0] - q.x * v[2], uz = q.x * v[1] - q.y * v[0];
  ux += ux; uy += uy; uz += uz;
  o[0] = v[0] + q.w * ux + (q.y * uz - q.z * uy);
  o[1] = v[1] + q.w * uy + (q.z * ux - q.x * uz);
  o[2] = v[2] + q.w * uz + (q.x * uy - q.y * ux);
}
inline void q2R(const Q& q, double R[9]) {  // Eigen toRotationMatrix
  const double tx = 2 * q.x, ty = 2 * q.y, tz = 2 * q.z;
  const double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w, txx = tx * q.x, txy = ty * q.x, txz = tz * q.x, tyy = ty * q.y,
               tyz = tz * q.y, tzz = tz * q.z;
  R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
  R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
}
inline Q R2q(const double R[9]) {  // Eigen quaternion-from-matrix (trace branches) [3P-memory]
  Q q;
  double t = R[0] + R[4] + R[8];
  if (t > 0) {
    t = std::sqrt(t + 1.0);
    q.w = 0.5 * t;
    t = 0.5 / t;
    q.x = (R[7] - R[5]) * t; q.y = (R[2] - R[6]) * t; q.z = (R[3] - R[1]) * t;
  } else {
    int i = 0;
    if (R[4] > R[0]) i = 1;
    if (R[8] > R[i * 4]) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    t = std::sqrt(R[i * 4] - R[j * 4] - R[k * 4] + 1.0);
    double v[3];
    v[i] = 0.5 * t;
    t = 0.5 / t;
    q.w = (R[k * 3 + j] - R[j * 3 + k]) * t;
    v[j] = (R[j * 3 + i] + R[i * 3 + j]) * t;
    v[k] = (R[k * 3 + i] + R[i * 3 + k]) * t;
    q.x = v[0]; q.y = v[1]; q.z = v[2];
  }
  return q;
}

// Utility::R2ypr / ypr2R, include/rgc_slam/utility.h:105-147 -- DEGREES, order Rz * Ry * Rx
void R2ypr(const double R[9], double ypr[3]) {
  const double n0 = R[0], n1 = R[3], n2 = R[6], o0 = R[1], o1 = R[4], a0 = R[2], a1 = R[5];
  const double y = std::atan2(n1, n0);
  const double p = std::atan2(-n2, n0 * std::cos(y) + n1 * std::sin(y));
  const double r = std::atan2(a0 * std::sin(y) - a1 * std::cos(y), -o0 * std::sin(y) + o1 * std::cos(y));
  ypr[0] = y / M_PI * 180.0; ypr[1] = p / M_PI * 180.0; ypr[2] = r / M_PI * 180.0;
}
void ypr2R(const double ypr[3], double R[9]) {
  const double y = ypr[0] / 180.0 * M_PI, p = ypr[1] / 180.0 * M_PI, r = ypr[2] / 180.0 * M_PI;
  const double cy = std::cos(y), sy = std::sin(y), cp = std::cos(p), sp = std::sin(p), cr = std::cos(r), sr = std::sin(r);
  // Rz * Ry * Rx
  R[0] = cy * cp; R[1] = cy * sp * sr - sy * cr; R[2] = cy * sp * cr + sy * sr;
  R[3] = sy * cp; R[4] = sy * sp * sr + cy * cr; R[5] = sy * sp * cr - cy * sr;
  R[6] = -sp;     R[7] = cp * sr;                R[8] = cp * cr;
}

// ---- B7 residuals (src/lidarFactor.hpp:132-172 DeltaRFactor, :228-265 DeltaPFactor, :311-350 Ground_DeltaFactor) ----
struct Fuse {
  const rgc_fuse_in* in;
  double c_imu;
  int nres;
};

void residuals(const Fuse& f, const Q& q, const double t[3], double* r) {
  const rgc_fuse_in& in = *f.in;
  int m = 0;
  {  // lidar rotation prior, variance = fitness (RGC_odometer.cpp:1031-1032)
    const Q ql{in.q_lidar_xyzw[0], in.q_lidar_xyzw[1], in.q_lidar_xyzw[2], in.q_lidar_xyzw[3]};
    const Q e = qmul(qconj(ql), q);  // ceres::QuaternionProduct(relative_q_inv, q)
    r[m++] = 2 * e.x / in.fitness; r[m++] = 2 * e.y / in.fitness; r[m++] = 2 * e.z / in.fitness;
  }
  if (in.use_ground) {
    const double pv = in.fitness / 10;  // :1090
    for (int a = 0; a < 3; a++) r[m++] = (t[a] - in.t_lidar[a]) / pv;
    // Ground_DeltaFactor: ground_s = {norm, vector_1, vector_2, distance, source} = groundparam.msg order
    const double* gl = in.ground_last;
    const double* gc = in.ground_cur;
    const Q qw{in.q_w_curr_f_xyzw[0], in.q_w_curr_f_xyzw[1], in.q_w_curr_f_xyzw[2], in.q_w_curr_f_xyzw[3]};
    double nc[3], dt[3];
    qrot(q, gc, nc);   // ground_norm_cur = q_last_curr * g_curr_norm
    qrot(qw, t, dt);   // delta_t = q_w_curr * t_last_curr
    const double dist_cur = gc[9] + dt[2];
    const double pvar = in.ground_cov;
    r[m++] = (gl[9] - dist_cur) / (pvar / 1000);
    r[m++] = std::fabs(gl[3] * nc[0] + gl[4] * nc[1] + gl[5] * nc[2]) / (pvar * 10);
    r[m++] = std::fabs(gl[6] * nc[0] + gl[7] * nc[1] + gl[8] * nc[2]) / (pvar * 10);
  }
  if (in.use_imu) {
    const Q qi{in.q_imu_xyzw[0], in.q_imu_xyzw[1], in.q_imu_xyzw[2], in.q_imu_xyzw[3]};
    const Q e = qmul(qconj(qi), q);
    r[m++] = 2 * e.x / f.c_imu; r[m++] = 2 * e.y / f.c_imu; r[m++] = 2 * e.z / f.c_imu;
  }
}

// ceres::EigenQuaternionParameterization::Plus [3P-memory]: x+ = q_delta (x) x, q_delta = (sin|d|/|d| d, cos|d|)
Q qplus(const Q& x, const double d[3]) {
  const double n = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
  if (n == 0) return x;
  const double s = std::sin(n) / n;
  return qmul(Q{s * d[0], s * d[1], s * d[2], std::cos(n)}, x);
}

bool solve_sym(double* A, double* b, int n) {  // Gaussian elimination with partial pivoting, in place, x -> b
  for (int c = 0; c < n; c++) {
    int piv = c;
    for (int r = c + 1; r < n; r++) if (std::fabs(A[r * n + c]) > std::fabs(A[piv * n + c])) piv = r;
    if (A[piv * n + c] == 0) return false;
    if (piv != c) { for (int j = 0; j < n; j++) std::swap(A[c * n + j], A[piv * n + j]); std::swap(b[c], b[piv]); }
    for (int r = c + 1; r < n; r++) {
      const double f = A[r * n + c] / A[c * n + c];
      for (int j = c; j < n; j++) A[r * n + j] -= f * A[c * n + j];
      b[r] -= f * b[c];
    }
  }
  for (int i = n - 1; i >= 0; i--) {
    double s = b[i];
    for (int j = i + 1; j < n; j++) s -= A[i * n + j] * b[j];
    b[i] = s / A[i * n + i];
  }
  return true;
}

}  // namespace

extern "C" {

void rgc_R2ypr(const double R[9], double ypr_deg[3]) { if (R && ypr_deg) R2ypr(R, ypr_deg); }
void rgc_ypr2R(const double ypr_deg[3], double R[9]) { if (ypr_deg && R) ypr2R(ypr_deg, R); }

// B1  vg_ICP::IMU_preintegration / IMU_preintegration2 over the samples of one sweep
// (src/RGC_odometer.cpp:883-931, 1418-1438).  stamps/gyr/acc: the vectors getIMUInterval returns (:1376-1416).
int rgc_imu_preintegrate(const double* stamps, const double* gyr3, const double* acc3, int n, double prev_time, double cur_time,
                         double dq_xyzw[4], double dq2_xyzw[4], double dp[3], double dv[3]) {
  if (!stamps || !gyr3 || n < 1 || !dq_xyzw) return RGC_ERR_INVALID;
  Q dq{0, 0, 0, 1}, dq2{0, 0, 0, 1};
  double p[3] = {0, 0, 0}, v[3] = {0, 0, 0};
  for (int i = 0; i < n; i++) {
    double dt;
    int i0, i1;
    if (i == 0) { dt = stamps[0] - prev_time; i0 = i1 = 0; }                      // :903-910
    else if (i == n - 1) { dt = cur_time - stamps[i - 1]; i0 = i - 1; i1 = i; }  // :911-918
    else { dt = stamps[i] - stamps[i - 1]; i0 = i - 1; i1 = i; }                 // :919-926
    const double* g = gyr3 + 3 * i;
    dq = qnormalized(qmul(dq, Q{g[0] * dt / 2, g[1] * dt / 2, g[2] * dt / 2, 1}));  // :1420-1421
    if (acc3) {  // mid-point integration, :1424-1438
      const double *a0 = acc3 + 3 * i0, *a1 = acc3 + 3 * i1, *g0 = gyr3 + 3 * i0, *g1 = gyr3 + 3 * i1;
      double ua0[3], ua1[3];
      qrot(dq2, a0, ua0);
      const double ug[3] = {0.5 * (g0[0] + g1[0]), 0.5 * (g0[1] + g1[1]), 0.5 * (g0[2] + g1[2])};
      const Q nq = qnormalized(qmul(dq2, Q{ug[0] * dt / 2, ug[1] * dt / 2, ug[2] * dt / 2, 1}));
      qrot(nq, a1, ua1);
      for (int a = 0; a < 3; a++) {
        const double ua = 0.5 * (ua0[a] + ua1[a]);
        p[a] = p[a] + v[a] * dt + 0.5 * ua * dt * dt;
        v[a] = v[a] + ua * dt;
      }
      dq2 = nq;
    }
  }
  dq = qnormalized(dq);  // q_last_curr.normalize(), :930
  dq_xyzw[0] = dq.x; dq_xyzw[1] = dq.y; dq_xyzw[2] = dq.z; dq_xyzw[3] = dq.w;
  if (dq2_xyzw) { dq2_xyzw[0] = dq2.x; dq2_xyzw[1] = dq2.y; dq2_xyzw[2] = dq2.z; dq2_xyzw[3] = dq2.w; }
  if (dp) std::memcpy(dp, p, sizeof(p));
  if (dv) std::memcpy(dv, v, sizeof(v));
  return RGC_OK;
}

// vg_ICP::imu_callback + ComplementaryFilter (src/RGC_odometer.cpp:444-486, 545-625), Mid_Filter (include/rgc_slam/utility.h)
void rgc_imu_filter_init(rgc_imu_filter* f) {
  if (!f) return;
  std::memset(f, 0, sizeof(*f));
  const double ba[3] = {0.23054, -0.22046, -0.14313}, bg[3] = {0.00127, -0.00061, -0.00267};  // utility.h:253-254
  std::memcpy(f->ba, ba, sizeof(ba));
  std::memcpy(f->bg, bg, sizeof(bg));
  f->Rwi[0] = f->Rwi[4] = f->Rwi[8] = 1.0;
}

namespace {
const int kMfSize[3] = {201, 41, 41};  // accx_MF(201), accy_MF(41), accz_MF(41), RGC_odometer.cpp:39
double mid_filter(rgc_imu_filter* f, int axis, double data) {  // Mid_Filter::MFilter: median of the ring buffer (zeros until it has filled)
  const int n = kMfSize[axis];
  f->mf_buf[axis][f->mf_count[axis]] = data;
  if (++f->mf_count[axis] >= n) f->mf_count[axis] = 0;
  double tmp[201];
  std::memcpy(tmp, f->mf_buf[axis], sizeof(double) * n);
  std::nth_element(tmp, tmp + (n - 1) / 2, tmp + n);  // the bubble sort's element (n - 1) / 2
  return tmp[(n - 1) / 2];
}
double norm_angle(double a) { return a > M_PI ? a - 2 * M_PI : (a < -M_PI ? a + 2 * M_PI : a); }               // utility.h:82-90
double norm_rp(double a) { return a > M_PI / 2 ? a - M_PI : (a < -M_PI / 2 ? a + M_PI : a); }                   // utility.h:92-100
}  // namespace

int rgc_imu_filter_push(rgc_imu_filter* f, double t, const double acc[3], const double gyr[3], double acc_out[3], double gyr_out[3]) {
  if (!f || !acc || !gyr) return RGC_ERR_INVALID;
  if (f->dropped < 100) {  // :446-451
    f->dropped++;
    return 0;
  }
  const double rad2deg = 180.0 / M_PI;
  double a[3], g[3];
  for (int i = 0; i < 3; i++) { a[i] = acc[i] - f->ba[i]; g[i] = gyr[i] - f->bg[i]; }  // :470-471
  if (acc_out) std::memcpy(acc_out, a, sizeof(a));
  if (gyr_out) std::memcpy(gyr_out, g, sizeof(g));
  f->count++;  // :484
  // ---- ComplementaryFilter, :545-625 ----
  double d_t = t - f->t_last;
  if (f->count == 1) d_t = 0.005;  // first_flag, :554-559 (imu_last = t_imu: the last roll / pitch are this sample's, i.e. still 0)
  double ax = mid_filter(f, 0, a[0]), ay = mid_filter(f, 1, a[1]), az = mid_filter(f, 2, a[2]);  // :561-563
  const double k = f->count < 300 ? 0.9 : 0.002;                                                 // :565-572
  double gx = g[0], gy = g[1], gz = g[2];
  if (std::fabs(gz * rad2deg) < 0.2) gz = 0;                                                     // :574-577
  if (f->count > 300) {                                                                           // :579-597
    double R[9];
    const double ypr[3] = {0.0, f->pitch * rad2deg, f->roll * rad2deg};
    ypr2R(ypr, R);
    const double mx = R[2] * 9.81, my = R[5] * 9.81;  // Rimu * (0, 0, 9.81)
    const double rx = std::fabs(mx) / std::fabs(ax);
    if (std::fabs(ax) > 0.3 && rx < 0.8) ax = rx * ax + (1 - rx) * mx;
    const double ry = std::fabs(my) / std::fabs(ay);
    if (std::fabs(ay) > 0.3 && ry < 0.8) ay = ry * ay + (1 - ry) * my;
  }
  const double roll_acc = std::atan2(ay, az), pitch_acc = -std::atan2(ax, az);                    // :598-599
  {  // body rates -> Euler rates: inverse of eulerRates2bodyRates(roll, pitch), :206-220, 601-605
    const double cr = std::cos(f->roll), sr = std::sin(f->roll), cp = std::cos(f->pitch), sp = std::sin(f->pitch);
    double M[9] = {1, 0, -sp, 0, cr, sr * cp, 0, -sr, cr * cp};
    double b[3] = {gx, gy, gz};
    if (solve_sym(M, b, 3)) { gx = b[0]; gy = b[1]; gz = b[2]; }
  }
  double roll = k * roll_acc + (1.0 - k) * (f->roll + gx * d_t);                                  // :607-609
  double pitch = k * pitch_acc + (1.0 - k) * (f->pitch + gy * d_t);
  double yaw = f->yaw + gz / 0.9998 * d_t;
  if (std::fabs(gz * rad2deg) > 5.0) {                                                            // :611-616
    const double low = 0.005;
    roll = low * roll + (1 - low) * f->roll_last;
    pitch = low * pitch + (1 - low) * f->pitch_last;
  }
  f->roll = norm_rp(roll);                                                                        // :618-621
  f->pitch = norm_rp(pitch);
  f->yaw = norm_angle(yaw);
  const double ypr[3] = {f->yaw * rad2deg, f->pitch * rad2deg, f->roll * rad2deg};
  ypr2R(ypr, f->Rwi);
  f->t_last = t;                                                                                  // imu_last = t_imu, :623
  f->roll_last = f->roll;
  f->pitch_last = f->pitch;
  return 1;
}

// the ground-change detector, src/RGC_odometer.cpp:1034-1087
void rgc_ground_gate_init(rgc_ground_gate* g) {
  if (!g) return;
  std::memset(g, 0, sizeof(*g));
  g->changegroundflag = 25;  // :327
  g->q_w_curr_delta[3] = 1.0;  // :20
}
namespace {
void gate_remember(rgc_ground_gate* g, const double q[4]) {
  const int slot = g->n_history < 64 ? g->n_history++ : 0;
  std::memcpy(g->history[slot], q, sizeof(double) * 4);
}
}  // namespace
void rgc_ground_gate_remember(rgc_ground_gate* g) {
  if (g) gate_remember(g, g->q_w_curr_delta);
}
int rgc_ground_gate_step(rgc_ground_gate* g, const double gl[11], const double gc[11], const double q_lidar[4], const double t_lidar[3],
                         const double dq_imu[4], const double q_w_curr[4], double q_w_curr_f[4]) {
  if (!g || !q_w_curr || !q_w_curr_f) return RGC_ERR_INVALID;
  const Q qw{q_w_curr[0], q_w_curr[1], q_w_curr[2], q_w_curr[3]};
  if (gl && gc && q_lidar && t_lidar) {
    const Q ql{q_lidar[0], q_lidar[1], q_lidar[2], q_lidar[3]};
    double nc[3];
    qrot(ql, gc, nc);                                                                  // ground_norm_cur, :1034
    const double dcur = gc[9] + nc[0] * t_lidar[0] + nc[1] * t_lidar[1] + nc[2] * t_lidar[2];  // :1035
    double e1 = 0;
    for (int a = 0; a < 3; a++) { const double d = gl[9] * gl[a] - dcur * nc[a]; e1 += d * d; }
    e1 = std::sqrt(e1);                                                                // :1036
    const double e2 = std::fabs(gl[3] * nc[0] + gl[4] * nc[1] + gl[5] * nc[2]);        // :1037
    double pitch_deg = 0.0;
    if (dq_imu) {
      double R[9], ypr[3];
      q2R(Q{dq_imu[0], dq_imu[1], dq_imu[2], dq_imu[3]}, R);
      R2ypr(R, ypr);                                                                   // d_ypr, :1039
      pitch_deg = ypr[1];
    }
    if (e1 >= 0.02 && e2 >= 0.02 && std::fabs(pitch_deg) > 0.5) {                      // :1042-1048
      g->changegroundflag = 0;
      g->gflag = 1;
    }
  }
  if (g->gflag == 1 && g->changegroundflag < 25) {                                     // :1049-1085
    g->changegroundflag++;
    if (g->changegroundflag == 25) {
      double R[9], now[3], tmp[3], best = 1000.0;
      q2R(qw, R);
      R2ypr(R, now);
      int pick = -1;
      for (int i = 0; i < g->n_history; i++) {
        q2R(Q{g->history[i][0], g->history[i][1], g->history[i][2], g->history[i][3]}, R);
        R2ypr(R, tmp);
        const double pe = tmp[1] - now[1], re = tmp[2] - now[2];
        const double e = std::sqrt(pe * pe + re * re);
        if (e < best) { best = e; pick = i; }
      }
      if (best < 4 && pick >= 0) {
        std::memcpy(g->q_w_curr_delta, g->history[pick], sizeof(double) * 4);
      } else {
        std::memcpy(g->q_w_curr_delta, q_w_curr, sizeof(double) * 4);
        gate_remember(g, g->q_w_curr_delta);
      }
      g->gflag = 0;
    }
  }
  const Q qd{g->q_w_curr_delta[0], g->q_w_curr_delta[1], g->q_w_curr_delta[2], g->q_w_curr_delta[3]};
  const Q qf = qnormalized(qmul(qconj(qd), qw));                                       // :1086-1087
  q_w_curr_f[0] = qf.x; q_w_curr_f[1] = qf.y; q_w_curr_f[2] = qf.z; q_w_curr_f[3] = qf.w;
  return g->gflag;
}

void rgc_default_fuse_in(rgc_fuse_in* in) {
  if (!in) return;
  std::memset(in, 0, sizeof(*in));
  in->q_lidar_xyzw[3] = in->q_w_curr_f_xyzw[3] = in->q_imu_xyzw[3] = 1.0;
  in->fitness = 1.0;
  in->ground_cov = 0.2;       // RGC_odometer.cpp:1092
  in->max_iterations = 6;     // :1190
}

// B7  the Ceres problem of RGC_odometer.cpp:1025-1032,1088-1119,1188-1193 as a damped Gauss-Newton on the 6-dim
// tangent space (NULL loss, same residual blocks and weights).  The problem is 6-DoF, <= 12 residuals and nearly
// quadratic: any converged damped GN reproduces Ceres' minimiser far inside 1e-4.
int rgc_fuse_pose(const rgc_fuse_in* in, double q_out[4], double t_out[3], int* iterations) {
  if (!in || !q_out || !t_out) return RGC_ERR_INVALID;
  if (!(in->fitness > 0) || !std::isfinite(in->fitness)) return RGC_ERR_INVALID;
  Fuse f{in, 1.0, 3 + (in->use_ground ? 6 : 0) + (in->use_imu ? 3 : 0)};
  if (in->use_imu) {  // :1107-1116
    double R[9], ypr[3];
    q2R(Q{in->q_imu_xyzw[0], in->q_imu_xyzw[1], in->q_imu_xyzw[2], in->q_imu_xyzw[3]}, R);
    R2ypr(R, ypr);
    f.c_imu = std::sqrt(ypr[0] * ypr[0] + ypr[1] * ypr[1] + ypr[2] * ypr[2]) > 0.6 ? 0.0005 : 1 - in->fitness;
    if (f.c_imu == 0) return RGC_ERR_INVALID;
  }
  // para_q / para_t seeded with the lidar result, :1017-1023
  Q q{in->q_lidar_xyzw[0], in->q_lidar_xyzw[1], in->q_lidar_xyzw[2], in->q_lidar_xyzw[3]};
  double t[3] = {in->t_lidar[0], in->t_lidar[1], in->t_lidar[2]};
  const int nd = in->use_ground ? 6 : 3;  // without the ground blocks para_t has no residual and keeps the lidar value (:1098-1102)
  const int nr = f.nres;
  double r0[12], rp[12], rm[12], J[12 * 6];
  auto cost = [&](const double* r) { double s = 0; for (int i = 0; i < nr; i++) s += r[i] * r[i]; return 0.5 * s; };
  double lambda = 1e-4;  // Ceres: initial trust-region radius 1e4
  int it = 0;
  const int maxit = in->max_iterations > 0 ? in->max_iterations : 6;
  for (; it < maxit; it++) {
    residuals(f, q, t, r0);
    const double c0 = cost(r0);
    const double h = 1e-6;
    for (int d = 0; d < nd; d++) {  // central differences on the manifold
      double dp[6] = {0, 0, 0, 0, 0, 0}, tp[3], tm[3];
      dp[d] = h;
      Q qp = d < 3 ? qplus(q, dp) : q;
      for (int a = 0; a < 3; a++) tp[a] = t[a] + dp[3 + a];
      residuals(f, qp, tp, rp);
      dp[d] = -h;
      Q qm = d < 3 ? qplus(q, dp) : q;
      for (int a = 0; a < 3; a++) tm[a] = t[a] + dp[3 + a];
      residuals(f, qm, tm, rm);
      for (int i = 0; i < nr; i++) J[i * 6 + d] = (rp[i] - rm[i]) / (2 * h);
    }
    double H[36], g[6];
    for (int a = 0; a < nd; a++) {
      g[a] = 0;
      for (int i = 0; i < nr; i++) g[a] += J[i * 6 + a] * r0[i];
      for (int b = 0; b < nd; b++) {
        double s = 0;
        for (int i = 0; i < nr; i++) s += J[i * 6 + a] * J[i * 6 + b];
        H[a * nd + b] = s;
      }
    }
    bool accepted = false;
    double step_norm = 0;
    for (int tries = 0; tries < 12 && !accepted; tries++) {
      double A[36], x[6];
      for (int a = 0; a < nd * nd; a++) A[a] = H[a];
      for (int a = 0; a < nd; a++) {
        const double dg = H[a * nd + a];
        A[a * nd + a] += lambda * (dg < 1e-6 ? 1e-6 : (dg > 1e32 ? 1e32 : dg));  // Ceres' clamped Jacobi-scaled LM diagonal
        x[a] = -g[a];
      }
      if (!solve_sym(A, x, nd)) { lambda *= 10; continue; }
      double d6[6] = {0, 0, 0, 0, 0, 0};
      for (int a = 0; a < nd; a++) d6[a] = x[a];
      const Q qn = qplus(q, d6);
      const double tn[3] = {t[0] + d6[3], t[1] + d6[4], t[2] + d6[5]};
      residuals(f, qn, tn, rp);
      const double c1 = cost(rp);
      if (c1 <= c0) {
        q = qn; t[0] = tn[0]; t[1] = tn[1]; t[2] = tn[2];
        lambda = lambda / 3 > 1e-12 ? lambda / 3 : 1e-12;
        accepted = true;
        step_norm = 0;
        for (int a = 0; a < nd; a++) step_norm += x[a] * x[a];
      } else {
        lambda *= 4;
      }
    }
    if (!accepted || step_norm < 1e-26) { it++; break; }
  }
  q = qnormalized(q);
  q_out[0] = q.x; q_out[1] = q.y; q_out[2] = q.z; q_out[3] = q.w;
  t_out[0] = t[0]; t_out[1] = t[1]; t_out[2] = t[2];
  if (iterations) *iterations = it;
  return RGC_OK;
}

// B8  pose composition + gravity blend, RGC_odometer.cpp:1194-1214
int rgc_compose_pose(const double q_w_curr[4], const double t_w_curr[3], const double q_fused[4], const double t_fused[3],
                     const double t_lidar[3], int use_imu, const double R_imu_wl[9], double q_w_out[4], double t_w_out[3],
                     double t_last_curr_out[3]) {
  if (!q_w_curr || !t_w_curr || !q_fused || !t_fused || !t_lidar || !q_w_out || !t_w_out) return RGC_ERR_INVALID;
  if (use_imu && !R_imu_wl) return RGC_ERR_INVALID;
  const Q qw{q_w_curr[0], q_w_curr[1], q_w_curr[2], q_w_curr[3]};
  double t1[3], t2[3], tl[3], d[3];
  qrot(qw, t_fused, t1);                          // :1195
  qrot(qw, t_lidar, t2);                          // :1196
  const double tt[3] = {t2[0], t2[1], t1[2]};     // :1197-1199  z from the fused solve, x/y from the lidar
  qrot(qconj(qw), tt, tl);                        // :1200
  qrot(qw, tl, d);
  for (int a = 0; a < 3; a++) t_w_out[a] = t_w_curr[a] + d[a];  // :1201
  Q qn = qnormalized(qmul(qw, Q{q_fused[0], q_fused[1], q_fused[2], q_fused[3]}));  // :1202-1203
  if (use_imu) {                                  // :1206-1214
    double Rw[9], yw[3], yi[3], Rn[9];
    q2R(qn, Rw);
    R2ypr(Rw, yw);
    R2ypr(R_imu_wl, yi);
    yw[1] = 0.95 * yw[1] + 0.05 * yi[1];
    yw[2] = 0.95 * yw[2] + 0.05 * yi[2];
    ypr2R(yw, Rn);
    qn = qnormalized(R2q(Rn));
  }
  q_w_out[0] = qn.x; q_w_out[1] = qn.y; q_w_out[2] = qn.z; q_w_out[3] = qn.w;
  if (t_last_curr_out) std::memcpy(t_last_curr_out, tl, sizeof(tl));
  return RGC_OK;
}

// C9  result extraction, RGC_odometer.cpp:1011-1016: Affine3f::translation(), Affine3f::rotation() (polar part via
// SVD; for the rigid output of align it is the 3x3 block re-orthonormalised) -> Quaternionf -> double
int rgc_extract_pose(const float T[16], double q_xyzw[4], double t[3]) {
  if (!T || !q_xyzw || !t) return RGC_ERR_INVALID;
  // polar decomposition by Newton iteration on the fp32 values held in double: R <- (R + R^-T) / 2
  double R[9] = {T[0], T[1], T[2], T[4], T[5], T[6], T[8], T[9], T[10]};
  for (int it = 0; it < 8; it++) {
    const double c00 = R[4] * R[8] - R[5] * R[7], c01 = R[5] * R[6] - R[3] * R[8], c02 = R[3] * R[7] - R[4] * R[6];
    const double det = R[0] * c00 + R[1] * c01 + R[2] * c02;
    if (!(std::fabs(det) > 1e-12)) return RGC_ERR_INVALID;
    const double inv[9] = {c00 / det, (R[2] * R[7] - R[1] * R[8]) / det, (R[1] * R[5] - R[2] * R[4]) / det,
                           c01 / det, (R[0] * R[8] - R[2] * R[6]) / det, (R[2] * R[3] - R[0] * R[5]) / det,
                           c02 / det, (R[1] * R[6] - R[0] * R[7]) / det, (R[0] * R[4] - R[1] * R[3]) / det};
    for (int a = 0; a < 3; a++)
      for (int b = 0; b < 3; b++) R[a * 3 + b] = 0.5 * (R[a * 3 + b] + inv[b * 3 + a]);
  }
  float Rf[9];
  for (int a = 0; a < 9; a++) Rf[a] = (float)R[a];  // q_drift is a Quaternionf of an Affine3f rotation
  double Rd[9];
  for (int a = 0; a < 9; a++) Rd[a] = (double)Rf[a];
  const Q q = R2q(Rd);
  const float qf[4] = {(float)q.x, (float)q.y, (float)q.z, (float)q.w};
  for (int a = 0; a < 4; a++) q_xyzw[a] = (double)qf[a];
  t[0] = (double)T[3]; t[1] = (double)T[7]; t[2] = (double)T[11];
  return RGC_OK;
}

// ---- f3: the message field tables and the two files the odometer writes ------------------------------------------------
// pcl::toROSMsg -> pcl::toPCLPointCloud2 emits one PointField per registered field of the point type [3P-memory: PCL
// point_types.hpp]: PointXYZI = x y z intensity (offsets 0 4 8 16, 32-byte points), PointXYZINormal = x y z intensity
// normal_x normal_y normal_z curvature (offsets 0 4 8 32 16 20 24 36, 48-byte points); all FLOAT32, count 1.
int rgc_pc2_point_fields(int kind, rgc_pc2_field* out, int cap, int* point_step) {
  static const struct { const char* name; int off; } xyzi[4] = {{"x", 0}, {"y", 4}, {"z", 8}, {"intensity", 16}};
  static const struct { const char* name; int off; } xyzin[8] = {{"x", 0}, {"y", 4}, {"z", 8}, {"intensity", 32}, {"normal_x", 16},
                                                                {"normal_y", 20}, {"normal_z", 24}, {"curvature", 36}};
  if (kind != 0 && kind != 1) return RGC_ERR_INVALID;
  const int nf = kind == 0 ? 4 : 8;
  if (point_step) *point_step = kind == 0 ? 32 : 48;
  if (!out) return nf;
  if (cap < nf) return RGC_ERR_INVALID;
  for (int i = 0; i < nf; i++) {
    const char* nm = kind == 0 ? xyzi[i].name : xyzin[i].name;
    memset(out[i].name, 0, sizeof(out[i].name));
    strncpy(out[i].name, nm, sizeof(out[i].name) - 1);
    out[i].offset = kind == 0 ? xyzi[i].off : xyzin[i].off;
    out[i].datatype = 7;
    out[i].count = 1;
  }
  return nf;
}

// RGC_odometer.cpp:1315-1316: std::fixed << setprecision(6) << stamp << " " << setprecision(9) << t ... q (x y z w) << endl
int rgc_tum_line(double stamp, const double t[3], const double q[4], char* buf, int cap) {
  if (!t || !q || !buf || cap <= 0) return RGC_ERR_INVALID;
  const int n = snprintf(buf, (size_t)cap, "%.6f %.9f %.9f %.9f %.9f %.9f %.9f %.9f\n", stamp, t[0], t[1], t[2], q[0], q[1], q[2], q[3]);
  return (n < 0 || n >= cap) ? RGC_ERR_INVALID : n;
}

// pcl::io::savePCDFileASCII(file, cloud) = PCDWriter::writeASCII(..., precision 8) [3P-memory: PCL pcd_io]: header lines
// exactly as below, then one "x y z intensity" line per point written through an ostream with precision(8) and the default
// (%g-like) float format; NaN is written as "nan".  binary = 1: the same header with DATA binary and 16 raw bytes per point.
int rgc_pcd_write(const char* path, const float* xyzi, int n, int binary) {
  if (!path || (!xyzi && n > 0) || n < 0) return RGC_ERR_INVALID;
  FILE* f = fopen(path, "wb");
  if (!f) return RGC_ERR_INVALID;
  fprintf(f, "# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z intensity\nSIZE 4 4 4 4\nTYPE F F F F\nCOUNT 1 1 1 1\n"
             "WIDTH %d\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS %d\nDATA %s\n", n, n, binary ? "binary" : "ascii");
  int ok = 1;
  if (binary) {
    if (n > 0) ok = fwrite(xyzi, 16, (size_t)n, f) == (size_t)n;
  } else {
    for (int i = 0; i < n && ok; i++) {
      for (int a = 0; a < 4; a++) {
        const float v = xyzi[4 * (size_t)i + a];
        if (std::isnan(v)) ok = ok && fputs("nan", f) >= 0;
        else ok = ok && fprintf(f, "%.8g", (double)v) > 0;
        ok = ok && fputc(a == 3 ? '\n' : ' ', f) != EOF;
      }
    }
  }
  ok = (fclose(f) == 0) && ok;
  return ok ? RGC_OK : RGC_ERR_INVALID;
}

}  // extern "C"
