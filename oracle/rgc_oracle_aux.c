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

/* ==========================================================================================================
 * A1-A8  ScanRegistration::laserCloudHandler, src/scanRegistration.cpp:89-730 (+ removeClosedPointCloud :732-763)
 * Sequential restatement.  Deliberate, documented clean-ups of reference quirks:
 *  - the member arrays (scan_angle, groundcloudMarked, cloudNeighborPicked, labels ...) are zero-initialised per
 *    frame (the reference leaks values of earlier frames into indices its reset loop :270-306 does not touch);
 *  - std::sort ties (:482-483) are broken by ascending point index (std::sort leaves them unspecified);
 *  - no 30000-point cap (static arrays :4-6,42-52);
 *  - rings with fewer than 11 points take no part in ground marking (the reference's size_t bound underflows).
 * ========================================================================================================== */
#include <stdlib.h>
#include <string.h>

typedef struct { float v; int i; } fe_key;
static int fe_cmp(const void* a, const void* b) {
  const fe_key* x = (const fe_key*)a; const fe_key* y = (const fe_key*)b;
  if (x->v < y->v) return -1;
  if (x->v > y->v) return 1;
  return x->i < y->i ? -1 : (x->i > y->i);
}

/* 3x3 symmetric eigen (Jacobi) from rgc_oracle.c */
void orc_eig3(const double A[9], double evals[3], double evecs[9]);

void orc_fe_default_params(orc_fe_params* p) {
  p->n_scans = 16; p->min_range = 0.5; p->max_range = 80.0; p->use_intensity = 1;   /* launch/run.launch:6,12-13,18 */
}

int orc_frontend(const float* in, int n, int stride, const orc_fe_params* prm, orc_fe_out* o) {
  const int NS = prm->n_scans;
  const double scanPeriod = 0.1;                                   /* :35 */
  const int groundScanInd = 7;                                     /* :34 */
  const double laderH = 0.56;                                      /* :39 */
  const float Ground_scan_range[16] = {2.66f, 3.04f, 3.56f, 4.30f, 5.44f, 7.41f, 11.63f, 27.12f, 0, 0, 0, 0, 0, 0, 0, 0}; /* :40 */
  if (NS != 16 && NS != 32 && NS != 64) return -1;
  memset(o->ring_count, 0, sizeof(o->ring_count));
  o->n_cloud = o->n_sharp = o->n_flat = o->n_inten = o->n_ground = 0; o->ground_valid = 0;
  /* A1: removeNaNFromPointCloud (:112) + removeClosedPointCloud (:113, :732-763) */
  float* P = (float*)malloc(sizeof(float) * 4 * (size_t)(n > 0 ? n : 1));
  int m = 0;
  const float th1 = (float)prm->min_range, th2 = (float)prm->max_range;
  for (int i = 0; i < n; i++) {
    const float* p = in + (size_t)i * stride;
    if (!isfinite(p[0]) || !isfinite(p[1]) || !isfinite(p[2])) continue;
    float dis = p[0] * p[0] + p[1] * p[1] + p[2] * p[2];
    if (dis < th1 * th1) continue;
    if (dis > th2 * th2) continue;
    if (p[0] < 0 && fabsf(p[1]) < 0.5) continue;
    P[m * 4] = p[0]; P[m * 4 + 1] = p[1]; P[m * 4 + 2] = p[2]; P[m * 4 + 3] = p[3];
    m++;
  }
  if (m < 1) { free(P); return 0; }
  /* A2: ring + rel-time (:117-213) */
  float startOri = -atan2f(P[1], P[0]);
  float endOri = (float)(-atan2f(P[(m - 1) * 4 + 1], P[(m - 1) * 4]) + 2 * M_PI);
  if (endOri - startOri > 3 * M_PI) endOri -= 2 * M_PI;
  else if (endOri - startOri < M_PI) endOri += 2 * M_PI;
  int* ring = (int*)malloc(sizeof(int) * (size_t)m);
  float* enc = (float*)malloc(sizeof(float) * (size_t)m);
  int halfPassed = 0;
  for (int i = 0; i < m; i++) {
    const float x = P[i * 4], y = P[i * 4 + 1], z = P[i * 4 + 2];
    float verticalAngle = (float)(atanf(z / sqrtf(x * x + y * y)) * 180 / M_PI);
    int scanID = 0;
    ring[i] = -1;
    if (NS == 16) {
      scanID = (int)((verticalAngle + 15) / 2 + 0.5);
      if (scanID > (NS - 1) || scanID < 0) continue;
    } else if (NS == 32) {
      scanID = (int)((verticalAngle + 92.0 / 3.0) * 3.0 / 4.0);
      if (scanID > (NS - 1) || scanID < 0) continue;
    } else {
      if (verticalAngle >= -8.83) scanID = (int)((2 - verticalAngle) * 3.0 + 0.5);
      else scanID = NS / 2 + (int)((-8.83 - verticalAngle) * 2.0 + 0.5);
      if (verticalAngle > 2 || verticalAngle < -24.33 || scanID > 50 || scanID < 0) continue;
    }
    float ori = -atan2f(y, x);
    if (!halfPassed) {
      if (ori < startOri - M_PI / 2) ori += 2 * M_PI;
      else if (ori > startOri + M_PI * 3 / 2) ori -= 2 * M_PI;
      if (ori - startOri > M_PI) halfPassed = 1;
    } else {
      ori += 2 * M_PI;
      if (ori < endOri - M_PI * 3 / 2) ori += 2 * M_PI;
      else if (ori > endOri + M_PI / 2) ori -= 2 * M_PI;
    }
    float relTime = (ori - startOri) / (endOri - startOri);
    ring[i] = scanID;
    enc[i] = (float)(scanID + scanPeriod * relTime);
    o->ring_count[scanID]++;
  }
  /* stable bucket by ring (:217-230) */
  int rstart[65];
  rstart[0] = 0;
  for (int r = 0; r < NS; r++) rstart[r + 1] = rstart[r] + o->ring_count[r];
  const int cs = rstart[NS];
  o->n_cloud = cs;
  float* C = o->cloud;                                  /* cs x 4: x,y,z, ring + 0.1*relTime */
  int* inum2 = (int*)malloc(sizeof(int) * (size_t)(cs > 0 ? cs : 1));
  {
    int fill[64];
    memcpy(fill, rstart, sizeof(int) * NS);
    for (int i = 0; i < m; i++) {
      if (ring[i] < 0) continue;
      const int d = fill[ring[i]]++;
      C[d * 4] = P[i * 4]; C[d * 4 + 1] = P[i * 4 + 1]; C[d * 4 + 2] = P[i * 4 + 2]; C[d * 4 + 3] = enc[i];
      inum2[d] = (int)P[i * 4 + 3];                     /* point_intensity is an int (:132,140) */
    }
  }
  for (int r = 0; r < NS; r++) { o->scan_start[r] = rstart[r] + 5; o->scan_end[r] = rstart[r + 1] - 5; }
  free(P); free(ring); free(enc);
  /* per-point arrays, zero-initialised per frame */
  float* range_vec = (float*)calloc((size_t)cs + 16, sizeof(float));
  float* scan_angle = (float*)calloc((size_t)cs + 16, sizeof(float));
  int* inum = (int*)malloc(sizeof(int) * ((size_t)cs + 16));
  float* curv = o->curvature; float* curv2 = o->curvature2; float* icurv = o->inten_curvature;
  float* dsrc = (float*)calloc((size_t)cs + 16, sizeof(float));
  float* osrc = (float*)calloc((size_t)cs + 16, sizeof(float));
  int* picked = o->picked; int* ipicked = (int*)calloc((size_t)cs + 16, sizeof(int));
  int* label = o->label; int* ilabel = o->inten_label; int* gmark = o->ground_marked;
  memset(curv, 0, sizeof(float) * (size_t)cs); memset(curv2, 0, sizeof(float) * (size_t)cs); memset(icurv, 0, sizeof(float) * (size_t)cs);
  memset(picked, 0, sizeof(int) * (size_t)cs); memset(label, 0, sizeof(int) * (size_t)cs); memset(ilabel, 0, sizeof(int) * (size_t)cs);
  memset(gmark, 0, sizeof(int) * (size_t)cs);
  memcpy(inum, inum2, sizeof(int) * (size_t)cs);
  /* A3 (:234-268) */
  for (int i = 0; i < cs; i++) range_vec[i] = sqrtf(C[i * 4] * C[i * 4] + C[i * 4 + 1] * C[i * 4 + 1] + C[i * 4 + 2] * C[i * 4 + 2]);
  for (int i = 5; i < cs - 5; i++) {
    if (range_vec[i] < 2) {
      const double a[3] = {C[(i + 5) * 4], C[(i + 5) * 4 + 1], C[(i + 5) * 4 + 2]};
      const double b[3] = {C[(i - 5) * 4], C[(i - 5) * 4 + 1], C[(i - 5) * 4 + 2]};
      const double c[3] = {(a[0] + b[0]) / 2, (a[1] + b[1]) / 2, (a[2] + b[2]) / 2};
      const double p[3] = {C[i * 4], C[i * 4 + 1], C[i * 4 + 2]};
      const double u[3] = {a[0] - b[0], a[1] - b[1], a[2] - b[2]}, v[3] = {p[0] - c[0], p[1] - c[1], p[2] - c[2]};
      const double nrm[3] = {u[1] * v[2] - u[2] * v[1], u[2] * v[0] - u[0] * v[2], u[0] * v[1] - u[1] * v[0]};
      const double nn = sqrt(nrm[0] * nrm[0] + nrm[1] * nrm[1] + nrm[2] * nrm[2]), pn = sqrt(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]);
      scan_angle[i] = (float)((nrm[0] * p[0] + nrm[1] * p[1] + nrm[2] * p[2]) / (nn * pn));
      if (scan_angle[i] < 0) scan_angle[i] = -scan_angle[i];
    }
  }
  for (int i = 5; i < cs - 5; i++) {
    if (scan_angle[i] < 0.07 && range_vec[i] < 2) {
      inum[i] = (int)(0.9 * inum2[i]);                                   /* deque<int>: truncation on every store */
      for (int j = -5; j < 6; j++) if (j != 0) inum[i] = (int)(inum[i] + 0.005 * inum2[i + j]);
    }
  }
  /* A4 (:270-306) */
  for (int i = 5; i < cs - 5; i++) {
    float dX = C[(i - 5) * 4], dY = C[(i - 5) * 4 + 1], dZ = C[(i - 5) * 4 + 2];
    for (int k = -4; k <= -1; k++) { dX = dX + C[(i + k) * 4]; dY = dY + C[(i + k) * 4 + 1]; dZ = dZ + C[(i + k) * 4 + 2]; }
    dX = dX - 10 * C[i * 4]; dY = dY - 10 * C[i * 4 + 1]; dZ = dZ - 10 * C[i * 4 + 2];
    for (int k = 1; k <= 5; k++) { dX = dX + C[(i + k) * 4]; dY = dY + C[(i + k) * 4 + 1]; dZ = dZ + C[(i + k) * 4 + 2]; }
    int dIi = inum[i - 5] + inum[i - 4] + inum[i - 3] + inum[i - 2] + inum[i - 1] - 10 * inum[i] + inum[i + 1] + inum[i + 2] + inum[i + 3] + inum[i + 4] + inum[i + 5];
    float diffI = (float)dIi;
    float dis_factor = (float)(2.0 / (1.0 + range_vec[i] / 20.0));
    if (dis_factor < 0.2) dis_factor = 0.2f;
    curv[i] = (dX * dX + dY * dY + dZ * dZ) * dis_factor;
    dsrc[i] = (float)(0.5 + dis_factor);
    float inten_factor;
    if (scan_angle[i] < 0.07 && range_vec[i] < 2) {
      inten_factor = (float)(scan_angle[i] * 10 + 0.6);
      icurv[i] = (float)((scan_angle[i] + 0.3) * diffI);
    } else {
      inten_factor = 3;
      icurv[i] = diffI;
    }
    osrc[i] = inten_factor;
    float dr = (float)(range_vec[i - 5] + range_vec[i - 4] + range_vec[i - 3] + range_vec[i - 2] + range_vec[i - 1] - 10.0 * range_vec[i] +
                       range_vec[i + 1] + range_vec[i + 2] + range_vec[i + 3] + range_vec[i + 4] + range_vec[i + 5]);
    curv2[i] = fabsf(dr * dis_factor);
  }
  /* A5 ground marking + plane (:308-431) */
  {
    double center[3] = {0, 0, 0}, gw = 0;
    int gsize = 0, gcap = 1024;
    double* near = (double*)malloc(sizeof(double) * 4 * (size_t)gcap);   /* x y z weight */
    int scanStart_ind = 0;
    for (int i = 0; i < groundScanInd && i < NS; i++) {
      const int sz = o->ring_count[i];
      if (sz >= 11) {
        for (int col = 5; col < sz - 5; col++) {
          const int ci = scanStart_ind + col;
          const float th = (float)(0.8 * (1.0 + i / (groundScanInd - 1)));          /* integer division, :323 */
          const float dr = fabsf(range_vec[ci] - Ground_scan_range[i]);
          const double w = 1.5 - i / (groundScanInd - 1);                             /* :325 */
          if (dr < th && C[ci * 4 + 2] < 0.3) {
            gmark[ci] = 1;
            for (int nn = -5; nn < 5; nn++) {                                         /* asymmetric, :333 */
              if (fabsf(range_vec[ci + nn] - range_vec[ci]) < th / 2) {
                gmark[ci + nn] = 1;
                if (o->n_ground < o->ground_cap) memcpy(o->ground_pts + (size_t)o->n_ground * 4, C + (size_t)(ci + nn) * 4, sizeof(float) * 4);
                o->n_ground++;
                if (gsize == gcap) { gcap *= 2; near = (double*)realloc(near, sizeof(double) * 4 * (size_t)gcap); }
                near[gsize * 4] = C[(ci + nn) * 4]; near[gsize * 4 + 1] = C[(ci + nn) * 4 + 1]; near[gsize * 4 + 2] = C[(ci + nn) * 4 + 2];
                near[gsize * 4 + 3] = w;
                for (int a = 0; a < 3; a++) center[a] += w * near[gsize * 4 + a];
                gw += w;
                gsize++;
              }
            }
          }
        }
      }
      scanStart_ind += sz;
    }
    if (gsize != 0) {
      for (int a = 0; a < 3; a++) center[a] /= gw;
      double cov[9] = {0};
      for (int j = 0; j < gsize; j++) {
        const double d[3] = {near[j * 4] - center[0], near[j * 4 + 1] - center[1], near[j * 4 + 2] - center[2]};
        for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) cov[a * 3 + b] += near[j * 4 + 3] * d[a] * d[b];
      }
      for (int a = 0; a < 9; a++) cov[a] /= gw;
      double ev[3], V[9];
      orc_eig3(cov, ev, V);                   /* descending, equal eigenvalues in their original order */
      /* SelfAdjointEigenSolver sorts ASCENDING by selection (the first minimum of what is left moves to the front).  Where eigenvalues are
       * distinct that is the reverse of the descending order; where they tie exactly -- a single ground point: the zero matrix, whose
       * eigenvectors come out as the identity -- the tied columns keep their original order: col(0) = (1, 0, 0), not (0, 0, 1). */
      int asc[3] = {2, 1, 0};
      if (ev[0] == ev[1] && ev[1] == ev[2]) { asc[0] = 0; asc[1] = 1; asc[2] = 2; }
      else if (ev[1] == ev[2]) { asc[0] = 1; asc[1] = 2; asc[2] = 0; }
      else if (ev[0] == ev[1]) { asc[0] = 2; asc[1] = 0; asc[2] = 1; }
      double nrm[3] = {V[asc[0]], V[3 + asc[0]], V[6 + asc[0]]}, v1[3] = {V[asc[1]], V[3 + asc[1]], V[6 + asc[1]]}, v2[3] = {V[asc[2]], V[3 + asc[2]], V[6 + asc[2]]};
      double nl = sqrt(nrm[0] * nrm[0] + nrm[1] * nrm[1] + nrm[2] * nrm[2]);
      for (int a = 0; a < 3; a++) nrm[a] /= nl;
      if (center[0] * nrm[0] + center[1] * nrm[1] + center[2] * nrm[2] < 0) for (int a = 0; a < 3; a++) nrm[a] = -nrm[a];
      double distance = 0, src1 = 0;
      for (int j = 0; j < gsize; j++) {
        double d[3] = {near[j * 4] - center[0], near[j * 4 + 1] - center[1], near[j * 4 + 2] - center[2]};
        const double dl = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
        double dw = 1 - 100 * fabs((nrm[0] * d[0] + nrm[1] * d[1] + nrm[2] * d[2]) / dl);
        if (dl == 0) dw = 1;                  /* Eigen normalized() of a zero vector stays zero */
        if (dw < 0) dw = 0.1;
        src1 += dw;
        distance += dw * (nrm[0] * near[j * 4] + nrm[1] * near[j * 4 + 1] + nrm[2] * near[j * 4 + 2]);
      }
      distance = distance / src1;
      src1 = src1 / gsize;
      if ((distance / laderH) > 1.1 || (distance / laderH) < 0.9) distance = laderH;
      if (src1 < 0.9) distance = 0.9 * laderH + 0.1 * distance;
      double* g = o->groundparam;
      g[0] = nrm[0]; g[1] = nrm[1]; g[2] = nrm[2]; g[3] = v1[0]; g[4] = v1[1]; g[5] = v1[2]; g[6] = v2[0]; g[7] = v2[1]; g[8] = v2[2];
      g[9] = distance; g[10] = 1 - src1;
      o->ground_valid = 1;
    }
    free(near);
  }
  /* A6 occlusion mask (:433-456) */
  for (int i = 5; i < cs - 5; i++) {
    const float d1 = range_vec[i], d2 = range_vec[i + 1];
    if (d1 - d2 > 0.04 * d2) { for (int k = -5; k <= 0; k++) picked[i + k] = 1; }
    else if (d2 - d1 > 0.04 * d1) { for (int k = 1; k <= 6; k++) if (i + k < cs) picked[i + k] = 1; }
  }
  /* A7 sector sort + greedy selection (:469-644) */
  fe_key* ks = (fe_key*)malloc(sizeof(fe_key) * ((size_t)cs + 1));
  fe_key* ki = (fe_key*)malloc(sizeof(fe_key) * ((size_t)cs + 1));
#define SQ(l_, m_) ((C[(l_) * 4] - C[(m_) * 4]) * (C[(l_) * 4] - C[(m_) * 4]) + (C[(l_) * 4 + 1] - C[(m_) * 4 + 1]) * (C[(l_) * 4 + 1] - C[(m_) * 4 + 1]) + \
                   (C[(l_) * 4 + 2] - C[(m_) * 4 + 2]) * (C[(l_) * 4 + 2] - C[(m_) * 4 + 2]))
  for (int i = 0; i < NS; i++) {
    const int S = o->scan_start[i], E = o->scan_end[i];
    if (E - S < 10) continue;
    for (int j = 0; j < 6; j++) {
      const int sp = S + (E - S) * j / 6, ep = S + (E - S) * (j + 1) / 6 - 1;
      int cnt = 0;
      for (int k = sp; k <= ep; k++) { ks[cnt].v = curv[k]; ks[cnt].i = k; ki[cnt].v = icurv[k]; ki[cnt].i = k; cnt++; }
      qsort(ks, (size_t)cnt, sizeof(fe_key), fe_cmp);
      qsort(ki, (size_t)cnt, sizeof(fe_key), fe_cmp);
      int largest = 0;
      for (int k = cnt - 1; k >= 0; k--) {
        const int ind = ks[k].i;
        if (picked[ind] == 0 && gmark[ind] != 1 && curv[ind] > 0.1 && curv2[ind] > 0.3) {
          largest++;
          if (largest <= 20) {
            label[ind] = 2;
            if (o->n_sharp < o->feat_cap) { float* f = o->sharp + (size_t)o->n_sharp * 5; memcpy(f, C + (size_t)ind * 4, 16); f[4] = dsrc[ind] + 1; }
            o->n_sharp++;
          } else if (largest <= 21) {
            label[ind] = 1;
          } else break;
          picked[ind] = 1;
          for (int l = 1; l <= 5; l++) { if (SQ(ind + l, ind + l - 1) > 0.05) break; picked[ind + l] = 1; }
          for (int l = -1; l >= -5; l--) { if (SQ(ind + l, ind + l + 1) > 0.05) break; picked[ind + l] = 1; }
        }
      }
      int smallest = 0;
      for (int k = 0; k < cnt; k++) {
        const int ind = ks[k].i;
        if (picked[ind] == 0 && curv[ind] < 0.3 && curv2[ind] < 0.4) {
          smallest++;
          if (smallest <= 40) {
            label[ind] = -1;
            if (o->n_flat < o->feat_cap) { float* f = o->flat + (size_t)o->n_flat * 5; memcpy(f, C + (size_t)ind * 4, 16); f[4] = dsrc[ind]; }
            o->n_flat++;
          } else break;
          picked[ind] = 1;
          for (int l = 1; l <= 5; l++) { if (SQ(ind + l, ind + l - 1) > 0.05) break; picked[ind + l] = 1; }
          for (int l = -1; l >= -5; l--) { if (SQ(ind + l, ind + l + 1) > 0.05) break; picked[ind + l] = 1; }
        }
      }
      int largest2 = 0;
      for (int k = cnt - 1; k >= 0; k--) {
        const int ind = ki[k].i;
        if (ipicked[ind] == 0 && gmark[ind] != 1 && icurv[ind] > 65 && label[ind] != 2 && label[ind] != 1) {
          largest2++;
          if (largest2 <= 20) {
            ilabel[ind] = 2;
            if (o->n_inten < o->feat_cap) { float* f = o->inten + (size_t)o->n_inten * 5; memcpy(f, C + (size_t)ind * 4, 16); f[4] = osrc[ind]; }
            o->n_inten++;
          } else if (largest2 <= 21) {
            ilabel[ind] = 1;
          } else break;
          ipicked[ind] = 1;
          for (int l = 1; l <= 5; l++) { if (fabsf((float)(inum[ind + l] - inum[ind + l - 1])) > 35) break; ipicked[ind + l] = 1; }
          for (int l = -1; l >= -5; l--) { if (fabsf((float)(inum[ind + l] - inum[ind + l + 1])) > 35) break; ipicked[ind + l] = 1; }
        }
      }
    }
  }
#undef SQ
  o->n_sharp_own = o->n_sharp;
  if (prm->use_intensity) {                         /* :645-663 */
    const double sharp = o->n_sharp, plane = o->n_flat;
    if (sharp / plane < 0.3) {
      for (int k = 0; k < o->n_inten; k++) {
        if (o->n_sharp < o->feat_cap && k < o->feat_cap) memcpy(o->sharp + (size_t)o->n_sharp * 5, o->inten + (size_t)k * 5, 20);
        o->n_sharp++;
      }
    }
  }
  free(ks); free(ki); free(inum); free(inum2); free(range_vec); free(scan_angle); free(dsrc); free(osrc); free(ipicked);
  return 0;
}
