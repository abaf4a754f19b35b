/*
 * rgc_oracle_map.c -- CPU restatement of the mapping node's scan-to-map FEATURE registration:
 * RGC_mapping.cpp:1069-1358 with lidarFactor.hpp:9-51 (LidarEdgeFactor) and :91-121 (LidarPlaneNormFactor).
 *
 * TEST INFRASTRUCTURE ONLY (see rgc_oracle.h).  PARITY UNPINNED: the reference has no tests for this path and cannot be
 * built here; the arithmetic that lives in third-party code is restated from its published algorithm --
 *   pcl::KdTreeFLANN::nearestKSearch  -> exact fp32 kNN (orc_knn_query);
 *   Eigen::SelfAdjointEigenSolver<3x3> -> cyclic Jacobi (orc_eig3);
 *   Eigen colPivHouseholderQr().solve  -> Householder QR with column pivoting, least-squares solve (below);
 *   ceres::Solve (trust region, LEVENBERG_MARQUARDT, DENSE_QR, HuberLoss(0.1), EigenQuaternionParameterization, <= 6
 *   iterations, Ceres 1.14 defaults) -> the LM loop of orc_mapreg_solve below (robustification by the Triggs corrector,
 *   which for Huber reduces to scaling residual and Jacobian by sqrt(rho'); damping diag(J^T J)/radius; step acceptance
 *   and radius update of LevenbergMarquardtStrategy).  Pinned by oracle/py_mapreg.py (numpy/scipy).
 * The IMU block of :1285-1312 (RelativeRFactor, PitchRollFactor) and the ground block of :1314-1340
 * (Ground_DeltaFactor_goable), both NULL loss, are optional inputs.
 */
#include "rgc_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* q (x,y,z,w) applied to p: Eigen's quaternion * vector */
static void quat_rot(const double q[4], const double p[3], double out[3]) {
  const double x = q[0], y = q[1], z = q[2], w = q[3];
  /* t = 2 * cross(q.xyz, p); out = p + w t + cross(q.xyz, t) */
  const double tx = 2 * (y * p[2] - z * p[1]), ty = 2 * (z * p[0] - x * p[2]), tz = 2 * (x * p[1] - y * p[0]);
  out[0] = p[0] + w * tx + (y * tz - z * ty);
  out[1] = p[1] + w * ty + (z * tx - x * tz);
  out[2] = p[2] + w * tz + (x * ty - y * tx);
}

/* pointAssociateToMap (RGC_mapping.cpp:1811-1820): double math, stored back into a float point */
static void associate_points(const float* feat, int nf, const double q[4], const double t[3], float* sel /* nf*3 */) {
  for (int i = 0; i < nf; i++) {
    const double p[3] = {(double)feat[4 * i], (double)feat[4 * i + 1], (double)feat[4 * i + 2]};
    double w[3];
    quat_rot(q, p, w);
    for (int a = 0; a < 3; a++) sel[3 * i + a] = (float)(w[a] + t[a]);
  }
}

/* least-squares solution of A x = b for a 5x3 A: Householder QR with column pivoting (Eigen::ColPivHouseholderQR) */
static void lstsq_5x3_colpiv(const double Ain[5][3], const double bin[5], double x[3]) {
  double A[5][3], b[5];
  int perm[3] = {0, 1, 2};
  memcpy(A, Ain, sizeof(A));
  memcpy(b, bin, sizeof(b));
  int rank = 3;
  for (int k = 0; k < 3; k++) {
    /* pivot: remaining column with the largest norm */
    int piv = k;
    double best = -1.0;
    for (int j = k; j < 3; j++) {
      double s = 0;
      for (int i = k; i < 5; i++) s += A[i][j] * A[i][j];
      if (s > best) { best = s; piv = j; }
    }
    if (best <= 0.0) { rank = k; break; }
    if (piv != k) {
      for (int i = 0; i < 5; i++) { double tmp = A[i][k]; A[i][k] = A[i][piv]; A[i][piv] = tmp; }
      int tp = perm[k]; perm[k] = perm[piv]; perm[piv] = tp;
    }
    /* Householder vector for column k */
    double norm = 0;
    for (int i = k; i < 5; i++) norm += A[i][k] * A[i][k];
    norm = sqrt(norm);
    const double alpha = A[k][k] > 0 ? -norm : norm;
    double v[5] = {0, 0, 0, 0, 0};
    for (int i = k; i < 5; i++) v[i] = A[i][k];
    v[k] -= alpha;
    double vv = 0;
    for (int i = k; i < 5; i++) vv += v[i] * v[i];
    if (vv > 0) {
      for (int j = k; j < 3; j++) {
        double d = 0;
        for (int i = k; i < 5; i++) d += v[i] * A[i][j];
        d = 2 * d / vv;
        for (int i = k; i < 5; i++) A[i][j] -= d * v[i];
      }
      double d = 0;
      for (int i = k; i < 5; i++) d += v[i] * b[i];
      d = 2 * d / vv;
      for (int i = k; i < 5; i++) b[i] -= d * v[i];
    }
  }
  double y[3] = {0, 0, 0};
  for (int k = rank - 1; k >= 0; k--) {
    double s = b[k];
    for (int j = k + 1; j < rank; j++) s -= A[k][j] * y[j];
    y[k] = s / A[k][k];
  }
  for (int k = 0; k < 3; k++) x[perm[k]] = y[k];
}

/* RGC_mapping.cpp:1092-1138 (and :1144-1188 with the last pose): edge association + line test */
int orc_mapreg_associate_edges(const float* feat, int nf, const double q_xyzw[4], const double t[3], const float* map_xyz, int nmap,
                               int mstride, orc_edge_factor* out, int num_threads) {
  if (nf <= 0) return 0;
  float* sel = (float*)malloc(sizeof(float) * 3 * (size_t)nf);
  int* idx = (int*)malloc(sizeof(int) * 5 * (size_t)nf);
  float* d2 = (float*)malloc(sizeof(float) * 5 * (size_t)nf);
  associate_points(feat, nf, q_xyzw, t, sel);
  int rc = orc_knn_query(map_xyz, nmap, mstride, sel, nf, 3, 5, idx, d2, num_threads);
  int count = 0;
  if (rc == 0) {
    for (int i = 0; i < nf; i++) {
      orc_edge_factor* f = &out[i];
      memset(f, 0, sizeof(*f));
      if (!(d2[5 * i + 4] < 1.0f)) continue; /* :1098 */
      double P[5][3], center[3] = {0, 0, 0};
      for (int j = 0; j < 5; j++)
        for (int a = 0; a < 3; a++) { P[j][a] = (double)map_xyz[(size_t)idx[5 * i + j] * mstride + a]; center[a] += P[j][a]; }
      for (int a = 0; a < 3; a++) center[a] /= 5.0;
      double cov[9] = {0};
      for (int j = 0; j < 5; j++) {
        const double z[3] = {P[j][0] - center[0], P[j][1] - center[1], P[j][2] - center[2]};
        for (int a = 0; a < 3; a++)
          for (int b = 0; b < 3; b++) cov[a * 3 + b] += z[a] * z[b];
      }
      double ev[3], U[9];
      orc_eig3(cov, ev, U); /* descending: ev[0] = the reference's eigenvalues()[2] */
      if (!(ev[0] > 3 * ev[1])) continue; /* :1122 */
      for (int a = 0; a < 3; a++) {
        const double dir = U[a * 3 + 0];
        f->a[a] = 0.1 * dir + center[a];
        f->b[a] = -0.1 * dir + center[a];
      }
      f->var = (double)feat[4 * i + 3]; /* edge_var = pointOri.normal_x (float) */
      f->valid = 1;
      count++;
    }
  }
  free(sel); free(idx); free(d2);
  return rc < 0 ? rc : count;
}

/* RGC_mapping.cpp:1191-1236 (and :1238-1282): plane association + fit */
int orc_mapreg_associate_planes(const float* feat, int nf, const double q_xyzw[4], const double t[3], const float* map_xyz, int nmap,
                                int mstride, orc_plane_factor* out, int num_threads) {
  if (nf <= 0) return 0;
  float* sel = (float*)malloc(sizeof(float) * 3 * (size_t)nf);
  int* idx = (int*)malloc(sizeof(int) * 5 * (size_t)nf);
  float* d2 = (float*)malloc(sizeof(float) * 5 * (size_t)nf);
  associate_points(feat, nf, q_xyzw, t, sel);
  int rc = orc_knn_query(map_xyz, nmap, mstride, sel, nf, 3, 5, idx, d2, num_threads);
  int count = 0;
  if (rc == 0) {
    for (int i = 0; i < nf; i++) {
      orc_plane_factor* f = &out[i];
      memset(f, 0, sizeof(*f));
      if (!(d2[5 * i + 4] < 2.0f)) continue; /* :1200 */
      double A[5][3];
      const double b[5] = {-1, -1, -1, -1, -1};
      for (int j = 0; j < 5; j++)
        for (int a = 0; a < 3; a++) A[j][a] = (double)map_xyz[(size_t)idx[5 * i + j] * mstride + a];
      double n[3];
      lstsq_5x3_colpiv(A, b, n);
      const double nn = sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
      if (!(nn > 0)) continue;
      const double d = 1.0 / nn; /* negative_OA_dot_norm */
      for (int a = 0; a < 3; a++) n[a] /= nn;
      int ok = 1;
      for (int j = 0; j < 5; j++)
        if (fabs(n[0] * A[j][0] + n[1] * A[j][1] + n[2] * A[j][2] + d) > 0.2) { ok = 0; break; } /* :1218-1226 */
      if (!ok) continue;
      for (int a = 0; a < 3; a++) f->n[a] = n[a];
      f->d = d;
      f->var = (double)feat[4 * i + 3];
      f->valid = 1;
      count++;
    }
  }
  free(sel); free(idx); free(d2);
  return rc < 0 ? rc : count;
}

/* ---- residuals and Jacobians on the 6-dim local parameterisation of one pose ---------------------------------------
 * EigenQuaternionParameterization::Plus: q' = dq (x) q with dq = (sin|d|/|d| d, cos|d|): a rotation by 2|d| about d, applied
 * on the left, so d(R p)/dd = -2 [R p]x at d = 0.  Ceres multiplies the ambient (4-dim) autodiff Jacobian by the plus
 * Jacobian, which is the same derivative. */
static void skew_times(const double v[3], double M[9]) { /* M = [v]x */
  M[0] = 0; M[1] = -v[2]; M[2] = v[1];
  M[3] = v[2]; M[4] = 0; M[5] = -v[0];
  M[6] = -v[1]; M[7] = v[0]; M[8] = 0;
}

/* accumulate one robustified residual block (dim rows) with Jacobian J (dim x 6) into H (21 upper), g (6), cost */
static void accumulate_block(const double* r, const double* J, int dim, double huber_a, double* H21, double* g6, double* cost) {
  double s = 0;
  for (int a = 0; a < dim; a++) s += r[a] * r[a];
  /* HuberLoss(a): rho(s) = s (s <= a^2), 2 a sqrt(s) - a^2 otherwise; rho' = 1 or a / sqrt(s); rho'' <= 0 -> Corrector scales
   * residual and Jacobian by sqrt(rho') */
  const double b = huber_a * huber_a;
  double rho, rho1;
  if (s > b) { const double sq = sqrt(s); rho = 2 * huber_a * sq - b; rho1 = huber_a / sq; }
  else { rho = s; rho1 = 1.0; }
  *cost += 0.5 * rho;
  if (!H21) return;
  const double w = rho1; /* (sqrt(rho1))^2 */
  int u = 0;
  for (int a = 0; a < 6; a++)
    for (int c = a; c < 6; c++) {
      double v = 0;
      for (int k = 0; k < dim; k++) v += J[k * 6 + a] * J[k * 6 + c];
      H21[u++] += w * v;
    }
  for (int a = 0; a < 6; a++) {
    double v = 0;
    for (int k = 0; k < dim; k++) v += J[k * 6 + a] * r[k];
    g6[a] += w * v;
  }
}

/* cost (and, if H21 != NULL, the normal equations) of one pose over its edge and plane factors */
static void pose_terms(const float* cfeat, const orc_edge_factor* ef, int ne, const float* sfeat, const orc_plane_factor* pf, int np,
                       const double q[4], const double t[3], double huber_a, double* H21, double* g6, double* cost) {
  for (int i = 0; i < ne; i++) {
    if (!ef[i].valid) continue;
    const double p[3] = {(double)cfeat[4 * i], (double)cfeat[4 * i + 1], (double)cfeat[4 * i + 2]};
    double Rp[3], lp[3];
    quat_rot(q, p, Rp);
    for (int a = 0; a < 3; a++) lp[a] = Rp[a] + t[a];
    const double* A = ef[i].a; const double* B = ef[i].b;
    const double u[3] = {lp[0] - A[0], lp[1] - A[1], lp[2] - A[2]}, v[3] = {lp[0] - B[0], lp[1] - B[1], lp[2] - B[2]};
    const double nu[3] = {u[1] * v[2] - u[2] * v[1], u[2] * v[0] - u[0] * v[2], u[0] * v[1] - u[1] * v[0]};
    const double de[3] = {A[0] - B[0], A[1] - B[1], A[2] - B[2]};
    const double den = sqrt(de[0] * de[0] + de[1] * de[1] + de[2] * de[2]);
    const double sc = ef[i].var / den;
    const double r[3] = {nu[0] * sc, nu[1] * sc, nu[2] * sc};
    double J[18];
    if (H21) {
      /* d nu / d lp = -[de]x ; d lp / d delta = -2 [Rp]x ; d lp / d t = I */
      double Sde[9], SRp[9];
      skew_times(de, Sde);
      skew_times(Rp, SRp);
      for (int a = 0; a < 3; a++)
        for (int c = 0; c < 3; c++) {
          double m = 0;
          for (int k = 0; k < 3; k++) m += Sde[a * 3 + k] * SRp[k * 3 + c];
          J[a * 6 + c] = sc * 2.0 * m;          /* (-[de]x)(-2 [Rp]x) */
          J[a * 6 + 3 + c] = -sc * Sde[a * 3 + c];
        }
    }
    accumulate_block(r, J, 3, huber_a, H21, g6, cost);
  }
  for (int i = 0; i < np; i++) {
    if (!pf[i].valid) continue;
    const double p[3] = {(double)sfeat[4 * i], (double)sfeat[4 * i + 1], (double)sfeat[4 * i + 2]};
    double Rp[3];
    quat_rot(q, p, Rp);
    const double* n = pf[i].n;
    const double r[1] = {(n[0] * (Rp[0] + t[0]) + n[1] * (Rp[1] + t[1]) + n[2] * (Rp[2] + t[2]) + pf[i].d) * pf[i].var};
    double J[6];
    if (H21) {
      /* n^T (-2 [Rp]x) = -2 (Rp x n)^T ... written out: (n^T [Rp]x)_c = (n x Rp)_c with a sign: n^T [v]x = (n x v)^T?  [v]x w = v x w,
       * n^T [v]x = -(v x n)^T ... = (n x v)^T */
      const double nxRp[3] = {n[1] * Rp[2] - n[2] * Rp[1], n[2] * Rp[0] - n[0] * Rp[2], n[0] * Rp[1] - n[1] * Rp[0]};
      for (int c = 0; c < 3; c++) { J[c] = -2.0 * nxRp[c] * pf[i].var; J[3 + c] = n[c] * pf[i].var; }
    }
    accumulate_block(r, J, 1, huber_a, H21, g6, cost);
  }
}

static void quat_plus(const double q[4], const double d[3], double out[4]) { /* EigenQuaternionParameterization::Plus */
  const double nd = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
  double dq[4];
  if (nd > 0.0) { const double s = sin(nd) / nd; dq[0] = s * d[0]; dq[1] = s * d[1]; dq[2] = s * d[2]; dq[3] = cos(nd); }
  else { dq[0] = d[0]; dq[1] = d[1]; dq[2] = d[2]; dq[3] = 1.0; }
  /* out = dq (x) q  (x,y,z,w) */
  const double ax = dq[0], ay = dq[1], az = dq[2], aw = dq[3], bx = q[0], by = q[1], bz = q[2], bw = q[3];
  out[0] = aw * bx + ax * bw + ay * bz - az * by;
  out[1] = aw * by - ax * bz + ay * bw + az * bx;
  out[2] = aw * bz + ax * by - ay * bx + az * bw;
  out[3] = aw * bw - ax * bx - ay * by - az * bz;
}

static void quat_mul(const double a[4], const double b[4], double o[4]) { /* x,y,z,w */
  o[0] = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
  o[1] = a[3] * b[1] - a[0] * b[2] + a[1] * b[3] + a[2] * b[0];
  o[2] = a[3] * b[2] + a[0] * b[1] - a[1] * b[0] + a[2] * b[3];
  o[3] = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
}

/* Ground_DeltaFactor_goable::operator() (lidarFactor.hpp:357-391) */
static void ground_residual(const orc_mapreg_ground* G, const double q[4], const double t[3], double r[3]) {
  const double lqc[4] = {-G->last_q[0], -G->last_q[1], -G->last_q[2], G->last_q[3]}; /* conjugate */
  double q_lc[4], dt[3] = {t[0] - G->last_t[0], t[1] - G->last_t[1], t[2] - G->last_t[2]}, t_lc[3], gn[3], delta_t[3];
  quat_mul(lqc, q, q_lc);
  quat_rot(lqc, dt, t_lc);
  quat_rot(q_lc, G->cur_norm, gn);
  quat_rot(G->q_history, t_lc, delta_t);
  const double dist_cur = G->cur_distance + delta_t[2];
  r[0] = (G->last_distance - dist_cur) / (G->p_var / 1000);
  r[1] = fabs(G->last_v1[0] * gn[0] + G->last_v1[1] * gn[1] + G->last_v1[2] * gn[2]) / (G->p_var * 10);
  r[2] = fabs(G->last_v2[0] * gn[0] + G->last_v2[1] * gn[1] + G->last_v2[2] * gn[2]) / (G->p_var * 10);
}

/* adds the (non-robustified, NULL loss) ground block of one pose: cost, and H21 / g6 if given.  The Jacobian on the 6-dim local
 * parameterisation is taken by central differences (step 1e-6): the block is three scalars of a smooth function (abs() aside,
 * where Ceres' autodiff uses sign()), and both sides of the parity comparison share this exact routine. */
static void ground_terms(const orc_mapreg_ground* G, const double q[4], const double t[3], double* H21, double* g6, double* cost) {
  if (!G) return;
  double r[3];
  ground_residual(G, q, t, r);
  *cost += 0.5 * (r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
  if (!H21) return;
  double J[18];
  const double h = 1e-6;
  for (int a = 0; a < 6; a++) {
    double rp[3], rm[3], qq[4], tt[3], d[3] = {0, 0, 0};
    for (int sgn = 0; sgn < 2; sgn++) {
      const double e = sgn ? -h : h;
      memcpy(qq, q, sizeof(qq)); memcpy(tt, t, sizeof(tt));
      if (a < 3) { d[0] = d[1] = d[2] = 0; d[a] = e; quat_plus(q, d, qq); } else tt[a - 3] += e;
      ground_residual(G, qq, tt, sgn ? rm : rp);
    }
    for (int k = 0; k < 3; k++) J[k * 6 + a] = (rp[k] - rm[k]) / (2 * h);
  }
  int u = 0;
  for (int a = 0; a < 6; a++)
    for (int c = a; c < 6; c++) {
      double v = 0;
      for (int k = 0; k < 3; k++) v += J[k * 6 + a] * J[k * 6 + c];
      H21[u++] += v;
    }
  for (int a = 0; a < 6; a++) {
    double v = 0;
    for (int k = 0; k < 3; k++) v += J[k * 6 + a] * r[k];
    g6[a] += v;
  }
}

/* ---- the IMU block of RGC_mapping.cpp:1285-1312 (USE_IMU = 1, the launch file's default): RelativeRFactor on (q_last, q_cur)
 * (lidarFactor.hpp:174-226) and a PitchRollFactor on each pose (:434-468), all with NULL loss: seven residuals that couple
 * the two rotations. */
static void quat_to_pitch_roll(const double q[4], double* pitch, double* roll) { /* Quaternion2EulerAngle, :405-433; q = x,y,z,w */
  const double w = q[3], x = q[0], y = q[1], z = q[2];
  *roll = atan2(2 * (w * x + y * z), 1 - 2 * (x * x + y * y));
  const double sinp = 2 * (w * y - x * z);
  *pitch = sinp >= 1 ? M_PI / 2 : (sinp <= -1 ? -M_PI / 2 : asin(sinp));
}

static void imu_residual(const orc_mapreg_imu* I, const double qc[4], const double ql[4], double r[7]) {
  const double qli[4] = {-ql[0], -ql[1], -ql[2], ql[3]};       /* QuaternionInverse(w_q_i) */
  const double rqi[4] = {-I->delta_q[0], -I->delta_q[1], -I->delta_q[2], I->delta_q[3]};
  double q_ij[4], e[4];
  quat_mul(qli, qc, q_ij);                                       /* q_i^-1 (x) q_j  with i = last, j = cur */
  quat_mul(rqi, q_ij, e);                                        /* relative_q^-1 (x) q_i_j */
  for (int a = 0; a < 3; a++) r[a] = 2 * e[a] / I->imu_cov;
  double p, ro;
  quat_to_pitch_roll(qc, &p, &ro);
  r[3] = 2 * (p - I->pitch_cur) / I->pr_var; r[4] = 2 * (ro - I->roll_cur) / I->pr_var;
  quat_to_pitch_roll(ql, &p, &ro);
  r[5] = 2 * (p - I->pitch_last) / I->pr_var; r[6] = 2 * (ro - I->roll_last) / I->pr_var;
}

/* adds the IMU block: cost, and the full 12x12 H / 12-vector g if given (central differences on the tangent, step 1e-6) */
static void imu_terms(const orc_mapreg_imu* I, const double x[14], double* H144, double* g12, double* cost) {
  if (!I) return;
  double r[7];
  imu_residual(I, x, x + 7, r);
  for (int k = 0; k < 7; k++) *cost += 0.5 * r[k] * r[k];
  if (!H144) return;
  double J[7 * 12];
  const double h = 1e-6;
  for (int a = 0; a < 12; a++) {
    double rp[7], rm[7];
    if (a % 6 >= 3) { for (int k = 0; k < 7; k++) J[k * 12 + a] = 0.0; continue; } /* translations do not enter */
    for (int sgn = 0; sgn < 2; sgn++) {
      double qc[4], ql[4], d[3] = {0, 0, 0};
      memcpy(qc, x, sizeof(qc)); memcpy(ql, x + 7, sizeof(ql));
      d[a % 6] = sgn ? -h : h;
      if (a < 6) quat_plus(x, d, qc); else quat_plus(x + 7, d, ql);
      imu_residual(I, qc, ql, sgn ? rm : rp);
    }
    for (int k = 0; k < 7; k++) J[k * 12 + a] = (rp[k] - rm[k]) / (2 * h);
  }
  for (int a = 0; a < 12; a++) {
    for (int c = 0; c < 12; c++) {
      double v = 0;
      for (int k = 0; k < 7; k++) v += J[k * 12 + a] * J[k * 12 + c];
      H144[a * 12 + c] += v;
    }
    double v = 0;
    for (int k = 0; k < 7; k++) v += J[k * 12 + a] * r[k];
    g12[a] += v;
  }
}

/* Cholesky solve of a symmetric positive definite n x n system (n <= 12) */
static int chol_solve_n(const double* Ain, const double* rhs, double* x, int n) {
  double L[144] = {0}, y[12];
  for (int i = 0; i < n; i++)
    for (int j = 0; j <= i; j++) {
      double s = Ain[i * n + j];
      for (int k = 0; k < j; k++) s -= L[i * n + k] * L[j * n + k];
      if (i == j) { if (!(s > 0)) return -1; L[i * n + i] = sqrt(s); }
      else L[i * n + j] = s / L[j * n + j];
    }
  for (int i = 0; i < n; i++) { double s = rhs[i]; for (int k = 0; k < i; k++) s -= L[i * n + k] * y[k]; y[i] = s / L[i * n + i]; }
  for (int i = n - 1; i >= 0; i--) { double s = y[i]; for (int k = i + 1; k < n; k++) s -= L[k * n + i] * x[k]; x[i] = s / L[i * n + i]; }
  return 0;
}

typedef struct {
  const float *corner_cur, *surf_cur, *corner_last, *surf_last;
  const orc_edge_factor *e_cur, *e_last;
  const orc_plane_factor *p_cur, *p_last;
  int n_ccur, n_scur, n_clast, n_slast;
  const orc_mapreg_ground *ground_cur, *ground_last;
  const orc_mapreg_imu* imu;
} mapreg_problem;

/* robust cost at x and, if H144 != NULL, the 12x12 normal equations (two 6x6 pose blocks + the IMU coupling) */
static double mapreg_evaluate(const mapreg_problem* P, const double x[14], double* H144, double* g12) {
  const double huber_a = 0.1;
  double H[2][21], g[2][6], cost = 0;
  memset(H, 0, sizeof(H)); memset(g, 0, sizeof(g));
  pose_terms(P->corner_cur, P->e_cur, P->n_ccur, P->surf_cur, P->p_cur, P->n_scur, x, x + 4, huber_a, H144 ? H[0] : NULL, H144 ? g[0] : NULL, &cost);
  pose_terms(P->corner_last, P->e_last, P->n_clast, P->surf_last, P->p_last, P->n_slast, x + 7, x + 11, huber_a, H144 ? H[1] : NULL,
             H144 ? g[1] : NULL, &cost);
  ground_terms(P->ground_cur, x, x + 4, H144 ? H[0] : NULL, H144 ? g[0] : NULL, &cost);
  ground_terms(P->ground_last, x + 7, x + 11, H144 ? H[1] : NULL, H144 ? g[1] : NULL, &cost);
  if (H144) {
    memset(H144, 0, sizeof(double) * 144);
    for (int b = 0; b < 2; b++) {
      int u = 0;
      for (int a = 0; a < 6; a++)
        for (int c = a; c < 6; c++) { H144[(6 * b + a) * 12 + 6 * b + c] = H[b][u]; H144[(6 * b + c) * 12 + 6 * b + a] = H[b][u]; u++; }
      for (int a = 0; a < 6; a++) g12[6 * b + a] = g[b][a];
    }
  }
  imu_terms(P->imu, x, H144, g12, &cost);
  return cost;
}

/* ceres::Solve restated (see the header of this file).  poses: q_cur[4] t_cur[3] q_last[4] t_last[3] in/out. */
int orc_mapreg_solve(const float* corner_cur, const orc_edge_factor* e_cur, int n_ccur, const float* surf_cur, const orc_plane_factor* p_cur,
                     int n_scur, const float* corner_last, const orc_edge_factor* e_last, int n_clast, const float* surf_last,
                     const orc_plane_factor* p_last, int n_slast, const orc_mapreg_ground* ground_cur, const orc_mapreg_ground* ground_last,
                     const orc_mapreg_imu* imu, double poses[14], int max_iterations, orc_mapreg_trace* trace) {
  const mapreg_problem P = {corner_cur, surf_cur, corner_last, surf_last, e_cur, e_last, p_cur, p_last, n_ccur, n_scur, n_clast, n_slast,
                            ground_cur, ground_last, imu};
  double radius = 1e4, decrease_factor = 2.0; /* initial_trust_region_radius, LevenbergMarquardtStrategy */
  double H[144], g[12];
  int it = 0, n_success = 0;
  double cost = mapreg_evaluate(&P, poses, H, g);
  if (trace) { trace->initial_cost = cost; trace->iterations = 0; trace->successful = 0; }
  for (it = 0; it < max_iterations; it++) {
    double gmax = 0; /* gradient tolerance (max-norm) */
    for (int a = 0; a < 12; a++) if (fabs(g[a]) > gmax) gmax = fabs(g[a]);
    if (gmax <= 1e-10) break;
    /* LM step: (H + diag(clamp(diag H)) / radius) d = -g */
    double A[144], rhs[12], d[12], model = 0;
    memcpy(A, H, sizeof(A));
    for (int a = 0; a < 12; a++) {
      double dg = H[a * 13];
      if (dg < 1e-6) dg = 1e-6;   /* min_lm_diagonal */
      if (dg > 1e32) dg = 1e32;   /* max_lm_diagonal */
      A[a * 13] += dg / radius;
      rhs[a] = -g[a];
    }
    const int ok = chol_solve_n(A, rhs, d, 12) == 0;
    if (ok)
      for (int a = 0; a < 12; a++) { /* model cost change = -d^T (g + 0.5 H d) */
        double Hd = 0;
        for (int c = 0; c < 12; c++) Hd += H[a * 12 + c] * d[c];
        model -= d[a] * (g[a] + 0.5 * Hd);
      }
    double rho = -1.0, xn[14], Hn[144], gn[12], new_cost = cost;
    memcpy(xn, poses, sizeof(xn));
    if (ok && model > 0) {
      for (int b = 0; b < 2; b++) {
        quat_plus(poses + 7 * b, d + 6 * b, xn + 7 * b);
        for (int a = 0; a < 3; a++) xn[7 * b + 4 + a] = poses[7 * b + 4 + a] + d[6 * b + 3 + a];
      }
      new_cost = mapreg_evaluate(&P, xn, Hn, gn);
      rho = (cost - new_cost) / model;
    }
    if (rho > 1e-3) { /* min_relative_decrease: successful step */
      const double old_cost = cost;
      memcpy(poses, xn, sizeof(xn));
      double f = 1.0 - pow(2.0 * rho - 1.0, 3);
      if (f < 1.0 / 3.0) f = 1.0 / 3.0;
      radius = radius / f;
      if (radius > 1e16) radius = 1e16; /* max_trust_region_radius */
      decrease_factor = 2.0;
      n_success++;
      memcpy(H, Hn, sizeof(H)); memcpy(g, gn, sizeof(g));
      cost = new_cost;
      double step2 = 0, x2 = 0;
      for (int a = 0; a < 12; a++) step2 += d[a] * d[a];
      for (int a = 0; a < 14; a++) x2 += poses[a] * poses[a];
      if (fabs(old_cost - cost) <= 1e-6 * old_cost) { it++; break; }                 /* function_tolerance */
      if (sqrt(step2) <= 1e-8 * (sqrt(x2) + 1e-8)) { it++; break; }                  /* parameter_tolerance */
    } else {
      radius = radius / decrease_factor;
      decrease_factor *= 2.0;
      if (radius < 1e-32) { it++; break; } /* min_trust_region_radius */
    }
  }
  if (trace) { trace->final_cost = cost; trace->iterations = it; trace->successful = n_success; trace->radius = radius; }
  return 0;
}

/* the whole optimisation block of one mapping frame: 2 x (associate, solve) -- RGC_mapping.cpp:1076-1358 -- then the
 * normalisation of :1375-1376 */
int orc_mapreg_optimize(const float* corner_cur, int n_ccur, const float* surf_cur, int n_scur, const float* corner_last, int n_clast,
                        const float* surf_last, int n_slast, const float* corner_map, int n_cmap, const float* surf_map, int n_smap,
                        int mstride, const orc_mapreg_ground* ground_cur, const orc_mapreg_ground* ground_last, const orc_mapreg_imu* imu,
                        double poses[14], orc_mapreg_trace trace[2], int num_threads) {
  /* the gate of :1069 */
  if (!(n_ccur > 10 && n_scur > 50 && n_cmap > 10 && n_smap > 50)) return 1;
  orc_edge_factor* ec = (orc_edge_factor*)malloc(sizeof(orc_edge_factor) * (size_t)(n_ccur > 0 ? n_ccur : 1));
  orc_edge_factor* el = (orc_edge_factor*)malloc(sizeof(orc_edge_factor) * (size_t)(n_clast > 0 ? n_clast : 1));
  orc_plane_factor* pc = (orc_plane_factor*)malloc(sizeof(orc_plane_factor) * (size_t)(n_scur > 0 ? n_scur : 1));
  orc_plane_factor* pl = (orc_plane_factor*)malloc(sizeof(orc_plane_factor) * (size_t)(n_slast > 0 ? n_slast : 1));
  int rc = 0;
  for (int iter = 0; iter < 2 && rc == 0; iter++) { /* :1076 */
    int a = orc_mapreg_associate_edges(corner_cur, n_ccur, poses, poses + 4, corner_map, n_cmap, mstride, ec, num_threads);
    int b = orc_mapreg_associate_edges(corner_last, n_clast, poses + 7, poses + 11, corner_map, n_cmap, mstride, el, num_threads);
    int c = orc_mapreg_associate_planes(surf_cur, n_scur, poses, poses + 4, surf_map, n_smap, mstride, pc, num_threads);
    int d = orc_mapreg_associate_planes(surf_last, n_slast, poses + 7, poses + 11, surf_map, n_smap, mstride, pl, num_threads);
    if (a < 0 || b < 0 || c < 0 || d < 0) { rc = -1; break; }
    if (trace) { trace[iter].n_edge_cur = a; trace[iter].n_edge_last = b; trace[iter].n_plane_cur = c; trace[iter].n_plane_last = d; }
    orc_mapreg_solve(corner_cur, ec, n_ccur, surf_cur, pc, n_scur, corner_last, el, n_clast, surf_last, pl, n_slast, ground_cur, ground_last, imu, poses, 6,
                     trace ? &trace[iter] : NULL);
  }
  /* q_w_last.normalize(); q_w_curr.normalize(); (:1375-1376) */
  for (int b = 0; b < 2; b++) {
    double* q = poses + (b ? 7 : 0);
    const double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    if (n > 0) for (int a = 0; a < 4; a++) q[a] /= n;
  }
  free(ec); free(el); free(pc); free(pl);
  return rc;
}
