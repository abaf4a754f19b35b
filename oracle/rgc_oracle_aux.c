/*
 * rgc_oracle_aux.c -- CPU restatement of the stages around the registration operator (de-skew, sub-map re-framing,
 * pose fusion, front-end).  TEST INFRASTRUCTURE ONLY -- see rgc_oracle.h.  PARITY UNPINNED.
 * Paths cited are relative to /root/reference/rgc_slam/.
 */
#include "rgc_oracle.h"

#include <math.h>
#include <stddef.h>

/* Eigen::Quaterniond * Vector3d (QuaternionBase::_transformVector) [3P-memory] */
static void quat_rotate(const double q[4] /* x y z w */, const double v[3], double out[3]) {
  double uv[3] = {q[1] * v[2] - q[2] * v[1], q[2] * v[0] - q[0] * v[2], q[0] * v[1] - q[1] * v[0]};
  uv[0] += uv[0]; uv[1] += uv[1]; uv[2] += uv[2];
  out[0] = v[0] + q[3] * uv[0] + (q[1] * uv[2] - q[2] * uv[1]);
  out[1] = v[1] + q[3] * uv[1] + (q[2] * uv[0] - q[0] * uv[2]);
  out[2] = v[2] + q[3] * uv[2] + (q[0] * uv[1] - q[1] * uv[0]);
}

/* Eigen::Quaterniond::slerp(t, other) from identity [3P-memory] */
static void slerp_from_identity(double t, const double o[4], double out[4]) {
  const double one = 1.0 - 2.220446049250313e-16;
  const double d = o[3];
  const double absD = fabs(d);
  double s0, s1;
  if (absD >= one) { s0 = 1.0 - t; s1 = t; }
  else {
    const double theta = acos(absD), st = sin(theta);
    s0 = sin((1.0 - t) * theta) / st;
    s1 = sin(t * theta) / st;
  }
  if (d < 0) s1 = -s1;
  out[0] = s1 * o[0]; out[1] = s1 * o[1]; out[2] = s1 * o[2]; out[3] = s0 + s1 * o[3];
}

/* vg_ICP::adjustDistortion, src/RGC_odometer.cpp:1441-1481 (the same loop body for sharp, flat and full clouds) */
void orc_deskew(float* xyzi, int n, int stride, const double q_last_curr[4], const double t_last_curr[3]) {
  const float SCAN_PERIOD = 0.1f;                          /* :323 */
  const double n2 = q_last_curr[0] * q_last_curr[0] + q_last_curr[1] * q_last_curr[1] + q_last_curr[2] * q_last_curr[2] + q_last_curr[3] * q_last_curr[3];
  const double qi[4] = {-q_last_curr[0] / n2, -q_last_curr[1] / n2, -q_last_curr[2] / n2, q_last_curr[3] / n2}; /* :1444 */
  for (int i = 0; i < n; i++) {
    float* p = xyzi + (size_t)i * stride;
    double s = 1 - (p[3] - (int)(p[3])) / SCAN_PERIOD;     /* :1469 (float expression widened to double) */
    double qs[4];
    slerp_from_identity(s, qi, qs);                        /* :1470 */
    double v[3] = {(double)p[0] - s * t_last_curr[0], (double)p[1] - s * t_last_curr[1], (double)p[2] - s * t_last_curr[2]}, e[3];
    quat_rotate(qs, v, e);                                 /* :1473 */
    p[0] = (float)e[0]; p[1] = (float)e[1]; p[2] = (float)e[2];
  }
}

/* vg_ICP::transformPointCloud, src/RGC_odometer.cpp:1495-1514 */
void orc_transform_cloud(const float* xyzi, int n, int stride, const double q[4], const double t[3], float* out4) {
  for (int i = 0; i < n; i++) {
    const float* p = xyzi + (size_t)i * stride;
    double v[3] = {(double)p[0], (double)p[1], (double)p[2]}, e[3];
    quat_rotate(q, v, e);
    out4[i * 4 + 0] = (float)(e[0] + t[0]);
    out4[i * 4 + 1] = (float)(e[1] + t[1]);
    out4[i * 4 + 2] = (float)(e[2] + t[2]);
    out4[i * 4 + 3] = stride > 3 ? p[3] : 0.f;
  }
}
