/*
 * rgc_oracle.c -- CPU restatement (C99 + OpenMP) of the RGC-SLAM registration path.
 * TEST INFRASTRUCTURE ONLY -- see rgc_oracle.h.  PARITY UNPINNED (no reference tests exist).
 *
 * Paths cited are relative to /root/reference/rgc_slam/.
 * Build with -ffp-contract=off: the reference is built for baseline x86-64 (-O3, no -march,
 * CMakeLists.txt:6) so no FMA contraction happens in FLANN's float L2 distance.
 */
#include "rgc_oracle.h"

#include <float.h>
#include <limits.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_MAXK 64

void orc_default_params(orc_params* p) {
  p->voxel_res = 1.0;
  p->max_iterations = 25;
  p->lm_max_iterations = 10;
  p->rotation_eps = 2e-3;
  p->translation_eps = 1e-6;
  p->lm_init_lambda_factor = 1e-9;
  p->k_correspondences = 20;
  p->neighbor_method = ORC_DIRECT1;
  p->num_threads = 14;
  p->regularization = ORC_REG_PLANE;
  p->voxel_mode = ORC_VOXEL_ADDITIVE;
}

static int clip_threads(int n) {
#ifdef _OPENMP
  int m = omp_get_max_threads();
  /* 0 = "the default": the host's threads, but no more than 16 -- the parallel loops of this restatement are short, and on a 256-thread
   * host they run SEVEN TIMES SLOWER on all threads than on the reference's 14 (setNumThreads(14), src/RGC_odometer.cpp:1006).  An explicit
   * count is honoured up to what the host has. */
  if (n <= 0) n = m < 16 ? m : 16;
  if (n > m) n = m;
  return n;
#else
  (void)n;
  return 1;
#endif
}

/* ------------------------------------------------------------------------------------------
 * Uniform grid used for EXACT nearest-neighbour queries.  Stands in for
 * pcl::search::KdTree -> KdTreeFLANN (epsilon = 0, sorted results, flann::L2_Simple<float>):
 * any exact search returns the same set; distances are accumulated in float as
 * ((dx*dx + dy*dy) + dz*dz) exactly like L2_Simple.  fast_gicp_impl.hpp:254.
 * ---------------------------------------------------------------------------------------- */
typedef struct {
  int n, stride;
  const float* p;
  double cell;
  int minc[3], dim[3];
  int* start; /* ncell + 1 */
  int* idx;   /* n, grouped by cell, ascending original index inside a cell */
} grid_t;

static inline int cell_of(double x, double cell) { return (int)floor(x / cell); }

static int grid_build_once(grid_t* g, double cell, double* mean_occ) {
  const int n = g->n;
  int minc[3] = {INT_MAX, INT_MAX, INT_MAX}, maxc[3] = {INT_MIN, INT_MIN, INT_MIN};
  for (int i = 0; i < n; i++) {
    const float* q = g->p + (size_t)i * g->stride;
    for (int a = 0; a < 3; a++) {
      int c = cell_of(q[a], cell);
      if (c < minc[a]) minc[a] = c;
      if (c > maxc[a]) maxc[a] = c;
    }
  }
  double nc = 1.0;
  for (int a = 0; a < 3; a++) {
    g->minc[a] = minc[a];
    g->dim[a] = maxc[a] - minc[a] + 1;
    nc *= (double)g->dim[a];
  }
  if (nc > 1.5e8) return -1;
  size_t ncell = (size_t)nc;
  g->cell = cell;
  free(g->start);
  g->start = (int*)calloc(ncell + 1, sizeof(int));
  if (!g->idx) g->idx = (int*)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
  int* cid = (int*)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
  for (int i = 0; i < n; i++) {
    const float* q = g->p + (size_t)i * g->stride;
    int cx = cell_of(q[0], cell) - g->minc[0];
    int cy = cell_of(q[1], cell) - g->minc[1];
    int cz = cell_of(q[2], cell) - g->minc[2];
    int c = (cz * g->dim[1] + cy) * g->dim[0] + cx;
    cid[i] = c;
    g->start[c + 1]++;
  }
  size_t occ = 0;
  for (size_t c = 0; c < ncell; c++) {
    if (g->start[c + 1]) occ++;
    g->start[c + 1] += g->start[c];
  }
  int* fill = (int*)malloc(sizeof(int) * ncell);
  memcpy(fill, g->start, sizeof(int) * ncell);
  for (int i = 0; i < n; i++) g->idx[fill[cid[i]]++] = i; /* stable: ascending index per cell */
  free(fill);
  free(cid);
  *mean_occ = occ ? (double)n / (double)occ : 0.0;
  return 0;
}

static int grid_build(grid_t* g, const float* p, int n, int stride, int k) {
  memset(g, 0, sizeof(*g));
  g->p = p;
  g->n = n;
  g->stride = stride;
  if (n <= 0) return -1;
  /* initial guess from the bounding box assuming a surface-like cloud */
  float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  for (int i = 0; i < n; i++) {
    const float* q = p + (size_t)i * stride;
    for (int a = 0; a < 3; a++) {
      if (q[a] < lo[a]) lo[a] = q[a];
      if (q[a] > hi[a]) hi[a] = q[a];
    }
  }
  double ext[3];
  for (int a = 0; a < 3; a++) ext[a] = (double)hi[a] - (double)lo[a] + 1e-3;
  /* two largest extents ~ surface area */
  double e0 = ext[0], e1 = ext[1], e2 = ext[2], t;
  if (e0 < e1) { t = e0; e0 = e1; e1 = t; }
  if (e1 < e2) { t = e1; e1 = e2; e2 = t; }
  if (e0 < e1) { t = e0; e0 = e1; e1 = t; }
  double cell = sqrt(e0 * e1 * 3.0 / (double)n);
  if (cell < 1e-3) cell = 1e-3;
  (void)k;
  double occ = 0;
  for (int pass = 0; pass < 6; pass++) {
    int rc = grid_build_once(g, cell, &occ);
    if (rc < 0) { cell *= 2.0; continue; }
    if (occ > 6.0 && pass < 4) { cell *= 1.0 / sqrt(occ / 3.0); continue; }
    return 0;
  }
  double dummy;
  while (grid_build_once(g, cell, &dummy) < 0) cell *= 2.0;
  return 0;
}

static void grid_free(grid_t* g) {
  free(g->start);
  free(g->idx);
  g->start = NULL;
  g->idx = NULL;
}

/* flann::L2_Simple<float>: result += diff*diff, accumulated in float, dims in order */
static inline float l2_simple_f32(const float* a, const float* b) {
  float r = 0.0f, d;
  d = a[0] - b[0]; r += d * d;
  d = a[1] - b[1]; r += d * d;
  d = a[2] - b[2]; r += d * d;
  return r;
}

/* max-heap on (d2, idx): the root is the WORST kept neighbour */
static inline int worse(float d2a, int ia, float d2b, int ib) { return d2a > d2b || (d2a == d2b && ia > ib); }

static void heap_push(float* hd, int* hi, int* hn, int k, float d2, int id) {
  if (*hn < k) {
    int i = (*hn)++;
    hd[i] = d2; hi[i] = id;
    while (i > 0) {
      int par = (i - 1) / 2;
      if (worse(hd[i], hi[i], hd[par], hi[par])) {
        float td = hd[i]; hd[i] = hd[par]; hd[par] = td;
        int ti = hi[i]; hi[i] = hi[par]; hi[par] = ti;
        i = par;
      } else break;
    }
  } else if (worse(hd[0], hi[0], d2, id)) {
    hd[0] = d2; hi[0] = id;
    int i = 0;
    for (;;) {
      int l = 2 * i + 1, r = l + 1, m = i;
      if (l < k && worse(hd[l], hi[l], hd[m], hi[m])) m = l;
      if (r < k && worse(hd[r], hi[r], hd[m], hi[m])) m = r;
      if (m == i) break;
      float td = hd[i]; hd[i] = hd[m]; hd[m] = td;
      int ti = hi[i]; hi[i] = hi[m]; hi[m] = ti;
      i = m;
    }
  }
}

/* exact k nearest neighbours of query q (any point) in the grid; returns count (<= k) sorted ascending */
static int grid_knn(const grid_t* g, const float* q, int k, int* out_idx, float* out_d2) {
  float hd[ORC_MAXK];
  int hi[ORC_MAXK];
  int hn = 0;
  int c[3];
  for (int a = 0; a < 3; a++) c[a] = cell_of(q[a], g->cell) - g->minc[a];
  int rmax = 0;
  for (int a = 0; a < 3; a++) {
    int lo = c[a], hi_ = g->dim[a] - 1 - c[a];
    if (lo > rmax) rmax = lo;
    if (hi_ > rmax) rmax = hi_;
  }
  for (int r = 0;; r++) {
    /* scan shell r (cells with Chebyshev distance exactly r), clipped to the grid */
    int z0 = c[2] - r, z1 = c[2] + r, y0 = c[1] - r, y1 = c[1] + r, x0 = c[0] - r, x1 = c[0] + r;
    for (int z = (z0 < 0 ? 0 : z0); z <= (z1 >= g->dim[2] ? g->dim[2] - 1 : z1); z++) {
      int az = abs(z - c[2]);
      for (int y = (y0 < 0 ? 0 : y0); y <= (y1 >= g->dim[1] ? g->dim[1] - 1 : y1); y++) {
        int ay = abs(y - c[1]);
        int on_face = (az == r || ay == r);
        int xstep = on_face ? 1 : (2 * r > 0 ? 2 * r : 1);
        for (int x = x0; x <= x1; x += xstep) {
          if (x < 0 || x >= g->dim[0]) continue;
          size_t cid = ((size_t)z * g->dim[1] + y) * g->dim[0] + x;
          for (int s = g->start[cid]; s < g->start[cid + 1]; s++) {
            int id = g->idx[s];
            float d2 = l2_simple_f32(q, g->p + (size_t)id * g->stride);
            heap_push(hd, hi, &hn, k, d2, id);
          }
        }
      }
    }
    if (r >= rmax) break; /* whole grid scanned */
    if (hn == k) {
      /* every unscanned point lies outside the cube of cells [c-r, c+r]; its true distance is at
       * least the distance from q to the cube faces that are not grid borders */
      double bound = DBL_MAX;
      for (int a = 0; a < 3; a++) {
        double qa = q[a];
        if (c[a] - r > 0) { double d = qa - (double)(c[a] - r + g->minc[a]) * g->cell; if (d < bound) bound = d; }
        if (c[a] + r < g->dim[a] - 1) { double d = (double)(c[a] + r + 1 + g->minc[a]) * g->cell - qa; if (d < bound) bound = d; }
      }
      if (bound == DBL_MAX) break;
      if (bound > 0 && (double)hd[0] < bound * bound * (1.0 - 1e-5)) break;
    }
  }
  /* sort ascending by (d2, idx): simple insertion sort (k small) */
  for (int i = 1; i < hn; i++) {
    float d = hd[i]; int id = hi[i]; int j = i - 1;
    while (j >= 0 && worse(hd[j], hi[j], d, id)) { hd[j + 1] = hd[j]; hi[j + 1] = hi[j]; j--; }
    hd[j + 1] = d; hi[j + 1] = id;
  }
  for (int i = 0; i < hn; i++) { out_idx[i] = hi[i]; if (out_d2) out_d2[i] = hd[i]; }
  return hn;
}

int orc_knn(const float* pts, int n, int stride, int k, int* idx_out, float* d2_out, int num_threads) {
  if (k > ORC_MAXK || k <= 0 || n < k) return -1;
  grid_t g;
  if (grid_build(&g, pts, n, stride, k) < 0) return -2;
  int nt = clip_threads(num_threads);
  (void)nt;
#pragma omp parallel for num_threads(nt) schedule(dynamic, 256)
  for (int i = 0; i < n; i++) {
    grid_knn(&g, pts + (size_t)i * stride, k, idx_out + (size_t)i * k, d2_out ? d2_out + (size_t)i * k : NULL);
  }
  grid_free(&g);
  return 0;
}

/* exact kNN of ARBITRARY query points against a cloud (pcl::KdTreeFLANN::nearestKSearch restated: fp32 L2_Simple over
 * x,y,z, ascending (distance, index)).  queries: nq points at qstride floats.  Fewer than k points in the cloud: -1. */
int orc_knn_query(const float* pts, int n, int stride, const float* queries, int nq, int qstride, int k, int* idx_out, float* d2_out,
                  int num_threads) {
  if (k > ORC_MAXK || k <= 0 || n < k) return -1;
  grid_t g;
  if (grid_build(&g, pts, n, stride, k) < 0) return -2;
  int nt = clip_threads(num_threads);
  (void)nt;
#pragma omp parallel for num_threads(nt) schedule(dynamic, 256)
  for (int i = 0; i < nq; i++) grid_knn(&g, queries + (size_t)i * qstride, k, idx_out + (size_t)i * k, d2_out ? d2_out + (size_t)i * k : NULL);
  grid_free(&g);
  return 0;
}

/* ------------------------------------------------------------------------------------------
 * symmetric 3x3 eigen decomposition, cyclic Jacobi.  Stands in for Eigen::JacobiSVD of the
 * symmetric PSD covariance (fast_gicp_impl.hpp:273): for such a matrix U = V = eigenvectors
 * and singular values = eigenvalues (descending).
 * ---------------------------------------------------------------------------------------- */
void orc_eig3(const double Ain[9], double evals[3], double evecs[9]) {
  double A[3][3], V[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) A[i][j] = 0.5 * (Ain[i * 3 + j] + Ain[j * 3 + i]);
  for (int sweep = 0; sweep < 60; sweep++) {
    double off = A[0][1] * A[0][1] + A[0][2] * A[0][2] + A[1][2] * A[1][2];
    double diag = A[0][0] * A[0][0] + A[1][1] * A[1][1] + A[2][2] * A[2][2];
    if (off <= 1e-40 * diag || off == 0.0) break;
    for (int p = 0; p < 2; p++)
      for (int q = p + 1; q < 3; q++) {
        if (A[p][q] == 0.0) continue;
        double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
        double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < 3; k++) { /* A <- A * G */
          double akp = A[k][p], akq = A[k][q];
          A[k][p] = c * akp - s * akq;
          A[k][q] = s * akp + c * akq;
        }
        for (int k = 0; k < 3; k++) { /* A <- G^T * A */
          double apk = A[p][k], aqk = A[q][k];
          A[p][k] = c * apk - s * aqk;
          A[q][k] = s * apk + c * aqk;
        }
        for (int k = 0; k < 3; k++) {
          double vkp = V[k][p], vkq = V[k][q];
          V[k][p] = c * vkp - s * vkq;
          V[k][q] = s * vkp + c * vkq;
        }
      }
  }
  int ord[3] = {0, 1, 2};
  double ev[3] = {A[0][0], A[1][1], A[2][2]};
  for (int i = 0; i < 2; i++)
    for (int j = i + 1; j < 3; j++)
      if (ev[ord[j]] > ev[ord[i]]) { int t = ord[i]; ord[i] = ord[j]; ord[j] = t; }
  for (int j = 0; j < 3; j++) {
    evals[j] = ev[ord[j]];
    for (int i = 0; i < 3; i++) evecs[i * 3 + j] = V[i][ord[j]];
  }
}

/* fast_gicp_impl.hpp:256-293: neighbours -> mean-centred -> cov = N N^T / k -> SVD ->
 * U diag(1,1,1e-3) V^T (RegularizationMethod::PLANE, :280-282,293). */
void orc_cov_from_neighbors(const float* pts, int stride, const int* idx, int k, double cov9[9], double normal[3]) {
  double mean[3] = {0, 0, 0};
  for (int j = 0; j < k; j++) {
    const float* q = pts + (size_t)idx[j] * stride;
    mean[0] += (double)q[0]; mean[1] += (double)q[1]; mean[2] += (double)q[2];
  }
  mean[0] /= k; mean[1] /= k; mean[2] /= k;
  double S[9] = {0};
  for (int j = 0; j < k; j++) {
    const float* q = pts + (size_t)idx[j] * stride;
    double d[3] = {(double)q[0] - mean[0], (double)q[1] - mean[1], (double)q[2] - mean[2]};
    for (int a = 0; a < 3; a++)
      for (int b = 0; b < 3; b++) S[a * 3 + b] += d[a] * d[b];
  }
  for (int a = 0; a < 9; a++) S[a] /= k;
  double ev[3], U[9];
  orc_eig3(S, ev, U);
  const double vals[3] = {1.0, 1.0, 1e-3};
  for (int a = 0; a < 3; a++)
    for (int b = 0; b < 3; b++) {
      double s = 0;
      for (int j = 0; j < 3; j++) s += U[a * 3 + j] * vals[j] * U[b * 3 + j];
      cov9[a * 3 + b] = s;
    }
  if (normal) { normal[0] = U[2]; normal[1] = U[5]; normal[2] = U[8]; }
}

static int inv3(const double A[9], double B[9]);
/* fast_gicp_impl.hpp:262-293: what calculate_covariances makes of one neighbourhood's sample covariance under each RegularizationMethod.
 * JacobiSVD of a symmetric positive semi-definite matrix is its eigen-decomposition (U = V, singular values = eigenvalues, descending). */
void orc_regularize(const double S[9], int method, double cov9[9]) {
  if (method == ORC_REG_NONE) { memcpy(cov9, S, sizeof(double) * 9); return; }                      /* :264-265 */
  if (method == ORC_REG_FROBENIUS) {                                                                  /* :266-271 */
    const double lambda = 1e-3;
    double C[9], Ci[9], N[9];
    memcpy(C, S, sizeof(C));
    C[0] += lambda; C[4] += lambda; C[8] += lambda;
    if (inv3(C, Ci)) memset(Ci, 0, sizeof(Ci));
    double nrm = 0;
    for (int a = 0; a < 9; a++) nrm += Ci[a] * Ci[a];
    nrm = sqrt(nrm);                                                                                  /* Eigen's norm(): Frobenius */
    for (int a = 0; a < 9; a++) N[a] = Ci[a] / nrm;
    if (inv3(N, cov9)) memset(cov9, 0, sizeof(double) * 9);
    return;
  }
  double ev[3], U[9], vals[3];
  orc_eig3(S, ev, U);                                                                                 /* :273 */
  if (method == ORC_REG_PLANE) { vals[0] = 1.0; vals[1] = 1.0; vals[2] = 1e-3; }                      /* :280-282 */
  else if (method == ORC_REG_MIN_EIG) { for (int j = 0; j < 3; j++) vals[j] = ev[j] > 1e-3 ? ev[j] : 1e-3; }  /* :283-285 */
  else { for (int j = 0; j < 3; j++) { const double t = ev[j] / ev[0]; vals[j] = t > 1e-3 ? t : 1e-3; } }     /* :286-289 */
  for (int a = 0; a < 3; a++)
    for (int b = 0; b < 3; b++) {
      double s = 0;
      for (int j = 0; j < 3; j++) s += U[a * 3 + j] * vals[j] * U[b * 3 + j];                         /* :293 */
      cov9[a * 3 + b] = s;
    }
}

int orc_covariances_m(const float* pts, int n, int stride, int k, int method, double* cov9_out, int num_threads) {
  if (k > ORC_MAXK || k <= 0 || n < k) return -1;
  grid_t g;
  if (grid_build(&g, pts, n, stride, k) < 0) return -2;
  int nt = clip_threads(num_threads);
  (void)nt;
#pragma omp parallel for num_threads(nt) schedule(guided, 8)
  for (int i = 0; i < n; i++) {
    int idx[ORC_MAXK];
    grid_knn(&g, pts + (size_t)i * stride, k, idx, NULL);
    double mean[3] = {0, 0, 0}, S[9] = {0};
    for (int j = 0; j < k; j++) {
      const float* q = pts + (size_t)idx[j] * stride;
      mean[0] += (double)q[0]; mean[1] += (double)q[1]; mean[2] += (double)q[2];
    }
    mean[0] /= k; mean[1] /= k; mean[2] /= k;
    for (int j = 0; j < k; j++) {
      const float* q = pts + (size_t)idx[j] * stride;
      double d[3] = {(double)q[0] - mean[0], (double)q[1] - mean[1], (double)q[2] - mean[2]};
      for (int a = 0; a < 3; a++)
        for (int b = 0; b < 3; b++) S[a * 3 + b] += d[a] * d[b];
    }
    for (int a = 0; a < 9; a++) S[a] /= k;
    orc_regularize(S, method, cov9_out + (size_t)i * 9);
  }
  grid_free(&g);
  return 0;
}

int orc_covariances(const float* pts, int n, int stride, int k, double* cov9_out, double* normal_out, int num_threads) {
  if (k > ORC_MAXK || k <= 0 || n < k) return -1;
  grid_t g;
  if (grid_build(&g, pts, n, stride, k) < 0) return -2;
  int nt = clip_threads(num_threads);
  (void)nt;
  /* fast_gicp_impl.hpp:250: omp parallel for schedule(guided, 8) */
#pragma omp parallel for num_threads(nt) schedule(guided, 8)
  for (int i = 0; i < n; i++) {
    int idx[ORC_MAXK];
    grid_knn(&g, pts + (size_t)i * stride, k, idx, NULL);
    orc_cov_from_neighbors(pts, stride, idx, k, cov9_out + (size_t)i * 9, normal_out ? normal_out + (size_t)i * 3 : NULL);
  }
  grid_free(&g);
  return 0;
}

/* ------------------------------------------------------------------------------------------
 * Gaussian voxel map, ADDITIVE mode.  fast_vgicp_voxel.hpp:105-122 (AdditiveGaussianVoxel),
 * :129-156 (create_voxelmap, points visited in cloud order, fp64 sums), :158-160 (voxel_coord).
 * The std::unordered_map<Vector3i,...,Vector3iHash> (:46-55,179-180) is replaced by an
 * open-addressing table; iteration order never influences a result on this path.
 * ---------------------------------------------------------------------------------------- */
typedef struct {
  int c[3];
  int num;
  double mean[3];
  double cov[9];
} voxel_t;

struct orc_voxelmap {
  double res;
  size_t cap; /* power of two */
  int* slot;  /* cap entries: index into vox or -1 */
  voxel_t* vox;
  int nvox, vcap;
};

void orc_voxel_coord(const double x[3], double res, int c[3]) {
  for (int a = 0; a < 3; a++) c[a] = (int)floor(x[a] / res - 0.5);
}

static inline uint64_t hash3(int x, int y, int z) {
  uint64_t h = (uint64_t)(uint32_t)x * 0x9E3779B97F4A7C15ull;
  h ^= (uint64_t)(uint32_t)y * 0xC2B2AE3D27D4EB4Full + (h << 6) + (h >> 2);
  h ^= (uint64_t)(uint32_t)z * 0x165667B19E3779F9ull + (h << 6) + (h >> 2);
  h ^= h >> 29;
  h *= 0xBF58476D1CE4E5B9ull;
  h ^= h >> 32;
  return h;
}

static voxel_t* vm_find(const orc_voxelmap* vm, const int c[3], int insert) {
  size_t mask = vm->cap - 1;
  size_t s = (size_t)hash3(c[0], c[1], c[2]) & mask;
  for (;;) {
    int v = vm->slot[s];
    if (v < 0) {
      if (!insert) return NULL;
      orc_voxelmap* w = (orc_voxelmap*)vm;
      if (w->nvox == w->vcap) {
        w->vcap = w->vcap * 2;
        w->vox = (voxel_t*)realloc(w->vox, sizeof(voxel_t) * (size_t)w->vcap);
      }
      voxel_t* nv = &w->vox[w->nvox];
      memset(nv, 0, sizeof(*nv));
      nv->c[0] = c[0]; nv->c[1] = c[1]; nv->c[2] = c[2];
      w->slot[s] = w->nvox++;
      return nv;
    }
    voxel_t* x = &vm->vox[v];
    if (x->c[0] == c[0] && x->c[1] == c[1] && x->c[2] == c[2]) return x;
    s = (s + 1) & mask;
  }
}

orc_voxelmap* orc_voxelmap_create(const float* pts, int n, int stride, const double* cov9, double res) {
  return orc_voxelmap_create_m(pts, n, stride, cov9, res, 0);
}
orc_voxelmap* orc_voxelmap_create_m(const float* pts, int n, int stride, const double* cov9, double res, int multiplicative) {
  orc_voxelmap* vm = (orc_voxelmap*)calloc(1, sizeof(*vm));
  vm->res = res;
  size_t cap = 64;
  while (cap < (size_t)n * 2 + 16) cap <<= 1;
  vm->cap = cap;
  vm->slot = (int*)malloc(sizeof(int) * cap);
  for (size_t i = 0; i < cap; i++) vm->slot[i] = -1;
  vm->vcap = 1024;
  vm->vox = (voxel_t*)malloc(sizeof(voxel_t) * (size_t)vm->vcap);
  for (int i = 0; i < n; i++) {
    const float* q = pts + (size_t)i * stride;
    double x[3] = {(double)q[0], (double)q[1], (double)q[2]};
    int c[3];
    orc_voxel_coord(x, res, c);
    voxel_t* v = vm_find(vm, c, 1);
    v->num++;                                        /* :112-116 append */
    if (multiplicative) {                            /* MultiplicativeGaussianVoxel::append, :82-91: cov += C^-1, mean += C^-1 p (the 4x4's (3,3) = 1 trick is the 3x3 inverse) */
      double Ci[9];
      if (inv3(cov9 + (size_t)i * 9, Ci)) memset(Ci, 0, sizeof(Ci));
      for (int a = 0; a < 9; a++) v->cov[a] += Ci[a];
      for (int a = 0; a < 3; a++) v->mean[a] += Ci[a * 3] * x[0] + Ci[a * 3 + 1] * x[1] + Ci[a * 3 + 2] * x[2];
      continue;
    }
    for (int a = 0; a < 3; a++) v->mean[a] += x[a];
    for (int a = 0; a < 9; a++) v->cov[a] += cov9[(size_t)i * 9 + a];
  }
  for (int j = 0; j < vm->nvox; j++) {               /* :118-121 finalize */
    voxel_t* v = &vm->vox[j];
    if (multiplicative) {                            /* :93-99: cov = (sum C^-1)^-1, mean = cov * sum C^-1 p */
      double C[9], m[3] = {v->mean[0], v->mean[1], v->mean[2]};
      if (inv3(v->cov, C)) memset(C, 0, sizeof(C));
      memcpy(v->cov, C, sizeof(C));
      for (int a = 0; a < 3; a++) v->mean[a] = C[a * 3] * m[0] + C[a * 3 + 1] * m[1] + C[a * 3 + 2] * m[2];
      continue;
    }
    for (int a = 0; a < 3; a++) v->mean[a] /= v->num;
    for (int a = 0; a < 9; a++) v->cov[a] /= v->num;
  }
  return vm;
}

void orc_voxelmap_free(orc_voxelmap* vm) {
  if (!vm) return;
  free(vm->slot);
  free(vm->vox);
  free(vm);
}

int orc_voxelmap_size(const orc_voxelmap* vm) { return vm->nvox; }

static int cmp_vox(const void* a, const void* b) {
  const voxel_t* x = *(const voxel_t* const*)a;
  const voxel_t* y = *(const voxel_t* const*)b;
  for (int k = 0; k < 3; k++) {
    if (x->c[k] < y->c[k]) return -1;
    if (x->c[k] > y->c[k]) return 1;
  }
  return 0;
}

void orc_voxelmap_dump(const orc_voxelmap* vm, int* coords, int* num, double* mean, double* cov9) {
  const voxel_t** order = (const voxel_t**)malloc(sizeof(voxel_t*) * (size_t)(vm->nvox > 0 ? vm->nvox : 1));
  for (int j = 0; j < vm->nvox; j++) order[j] = &vm->vox[j];
  qsort(order, (size_t)vm->nvox, sizeof(voxel_t*), cmp_vox);
  for (int j = 0; j < vm->nvox; j++) {
    const voxel_t* v = order[j];
    for (int a = 0; a < 3; a++) { coords[j * 3 + a] = v->c[a]; mean[j * 3 + a] = v->mean[a]; }
    num[j] = v->num;
    for (int a = 0; a < 9; a++) cov9[(size_t)j * 9 + a] = v->cov[a];
  }
  free(order);
}

/* ------------------------------------------------------------------------------------------
 * Registration object (FastVGICP as configured at RGC_odometer.cpp:998-1008)
 * ---------------------------------------------------------------------------------------- */
typedef struct {
  int src;          /* source point index            */
  const voxel_t* v; /* target voxel                  */
  double M[9];      /* (C_B + R C_A R^T)^-1, frozen  */
} corr_t;

struct orc_reg {
  orc_params prm;
  float* src; int n_src;
  float* tgt; int n_tgt;
  double* src_cov; double* tgt_cov;
  orc_voxelmap* vm;
  corr_t* corr; int n_corr, corr_cap;
  grid_t tgt_grid; int tgt_grid_ok;
  int n_lin, n_err;
};

orc_reg* orc_reg_create(const orc_params* p) {
  orc_reg* r = (orc_reg*)calloc(1, sizeof(*r));
  if (p) r->prm = *p; else orc_default_params(&r->prm);
  return r;
}

static void reg_clear_target(orc_reg* r) {
  free(r->tgt); r->tgt = NULL; r->n_tgt = 0;
  free(r->tgt_cov); r->tgt_cov = NULL;
  orc_voxelmap_free(r->vm); r->vm = NULL;
  if (r->tgt_grid_ok) { grid_free(&r->tgt_grid); r->tgt_grid_ok = 0; }
}
static void reg_clear_source(orc_reg* r) {
  free(r->src); r->src = NULL; r->n_src = 0;
  free(r->src_cov); r->src_cov = NULL;
}

void orc_reg_free(orc_reg* r) {
  if (!r) return;
  reg_clear_target(r);
  reg_clear_source(r);
  free(r->corr);
  free(r);
}

static float* copy_xyz(const float* pts, int n, int stride) {
  float* o = (float*)malloc(sizeof(float) * 3 * (size_t)(n > 0 ? n : 1));
  for (int i = 0; i < n; i++) {
    o[i * 3 + 0] = pts[(size_t)i * stride + 0];
    o[i * 3 + 1] = pts[(size_t)i * stride + 1];
    o[i * 3 + 2] = pts[(size_t)i * stride + 2];
  }
  return o;
}

/* fast_vgicp_impl.hpp:56-63 + fast_gicp_impl.hpp:83-90: new target => covariances and voxel map dropped */
int orc_reg_set_target(orc_reg* r, const float* pts, int n, int stride) {
  reg_clear_target(r);
  r->tgt = copy_xyz(pts, n, stride);
  r->n_tgt = n;
  return 0;
}
/* fast_gicp_impl.hpp:72-80 */
int orc_reg_set_source(orc_reg* r, const float* pts, int n, int stride) {
  reg_clear_source(r);
  r->src = copy_xyz(pts, n, stride);
  r->n_src = n;
  return 0;
}

/* fast_gicp_impl.hpp:103-110 (covariances) + fast_vgicp_impl.hpp:120-123 (voxel map on first linearize) */
int orc_reg_prepare(orc_reg* r) {
  const int k = r->prm.k_correspondences;
  if (!r->src || !r->tgt) return -1;
  if (r->n_src < k || r->n_tgt < k) return -2;
  if (!r->src_cov) {
    r->src_cov = (double*)malloc(sizeof(double) * 9 * (size_t)r->n_src);
    int rc = r->prm.regularization == ORC_REG_PLANE ? orc_covariances(r->src, r->n_src, 3, k, r->src_cov, NULL, r->prm.num_threads)
                                                    : orc_covariances_m(r->src, r->n_src, 3, k, r->prm.regularization, r->src_cov, r->prm.num_threads);
    if (rc) return rc;
  }
  if (!r->tgt_cov) {
    r->tgt_cov = (double*)malloc(sizeof(double) * 9 * (size_t)r->n_tgt);
    int rc = r->prm.regularization == ORC_REG_PLANE ? orc_covariances(r->tgt, r->n_tgt, 3, k, r->tgt_cov, NULL, r->prm.num_threads)
                                                    : orc_covariances_m(r->tgt, r->n_tgt, 3, k, r->prm.regularization, r->tgt_cov, r->prm.num_threads);
    if (rc) return rc;
  }
  if (!r->vm) r->vm = orc_voxelmap_create_m(r->tgt, r->n_tgt, 3, r->tgt_cov, r->prm.voxel_res, r->prm.voxel_mode == ORC_VOXEL_MULTIPLICATIVE);
  return 0;
}

const double* orc_reg_source_cov(orc_reg* r) { return orc_reg_prepare(r) ? NULL : r->src_cov; }
const double* orc_reg_target_cov(orc_reg* r) { return orc_reg_prepare(r) ? NULL : r->tgt_cov; }
const orc_voxelmap* orc_reg_voxelmap(orc_reg* r) { return orc_reg_prepare(r) ? NULL : r->vm; }
int orc_reg_num_correspondences(const orc_reg* r) { return r->n_corr; }
int orc_reg_num_linearize(const orc_reg* r) { return r->n_lin; }
int orc_reg_num_error(const orc_reg* r) { return r->n_err; }

/* fast_vgicp_voxel.hpp:10-44 */
static int neighbor_offsets(int method, int off[27][3]) {
  if (method == ORC_DIRECT1) { off[0][0] = off[0][1] = off[0][2] = 0; return 1; }
  if (method == ORC_DIRECT7) {
    static const int o7[7][3] = {{0, 0, 0}, {1, 0, 0}, {-1, 0, 0}, {0, 1, 0}, {0, -1, 0}, {0, 0, 1}, {0, 0, -1}};
    memcpy(off, o7, sizeof(o7));
    return 7;
  }
  int m = 0;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++)
      for (int k = 0; k < 3; k++) { off[m][0] = i - 1; off[m][1] = j - 1; off[m][2] = k - 1; m++; }
  return 27;
}

static int inv3(const double A[9], double B[9]) {  /* Eigen's Matrix3d::inverse(): cofactors over the determinant */
  double c00 = A[4] * A[8] - A[5] * A[7];
  double c01 = A[5] * A[6] - A[3] * A[8];
  double c02 = A[3] * A[7] - A[4] * A[6];
  double det = A[0] * c00 + A[1] * c01 + A[2] * c02;
  if (det == 0.0) return -1;
  double id = 1.0 / det;
  B[0] = c00 * id;
  B[1] = (A[2] * A[7] - A[1] * A[8]) * id;
  B[2] = (A[1] * A[5] - A[2] * A[4]) * id;
  B[3] = c01 * id;
  B[4] = (A[0] * A[8] - A[2] * A[6]) * id;
  B[5] = (A[2] * A[3] - A[0] * A[5]) * id;
  B[6] = c02 * id;
  B[7] = (A[1] * A[6] - A[0] * A[7]) * id;
  B[8] = (A[0] * A[4] - A[1] * A[3]) * id;
  return 0;
}

static inline void xform(const double T[16], const double p[3], double q[3]) {
  q[0] = T[0] * p[0] + T[1] * p[1] + T[2] * p[2] + T[3];
  q[1] = T[4] * p[0] + T[5] * p[1] + T[6] * p[2] + T[7];
  q[2] = T[8] * p[0] + T[9] * p[1] + T[10] * p[2] + T[11];
}

/* fast_vgicp_impl.hpp:73-116.  The reference concatenates per-thread lists (order depends on the
 * OpenMP schedule); here the list is in source order, which only changes fp64 summation order. */
static void update_correspondences(orc_reg* r, const double T[16]) {
  int off[27][3];
  const int noff = neighbor_offsets(r->prm.neighbor_method, off);
  size_t need = (size_t)r->n_src * (size_t)noff;
  if ((size_t)r->corr_cap < need) {
    free(r->corr);
    r->corr = (corr_t*)malloc(sizeof(corr_t) * (need ? need : 1));
    r->corr_cap = (int)need;
  }
  int nt = clip_threads(r->prm.num_threads);
  (void)nt;
  /* pass 1: lookups into a dense slot array (n_src * noff), then compact in order */
  const voxel_t** hit = (const voxel_t**)malloc(sizeof(voxel_t*) * (need ? need : 1));
#pragma omp parallel for num_threads(nt) schedule(guided, 8)
  for (int i = 0; i < r->n_src; i++) {
    double p[3] = {(double)r->src[i * 3], (double)r->src[i * 3 + 1], (double)r->src[i * 3 + 2]}, q[3];
    xform(T, p, q);                               /* :84-85 */
    int c[3];
    orc_voxel_coord(q, r->prm.voxel_res, c);      /* :86 */
    for (int o = 0; o < noff; o++) {
      int cc[3] = {c[0] + off[o][0], c[1] + off[o][1], c[2] + off[o][2]};
      hit[(size_t)i * noff + o] = vm_find(r->vm, cc, 0); /* :88-92 */
    }
  }
  int m = 0;
  for (size_t s = 0; s < need; s++)
    if (hit[s]) { r->corr[m].src = (int)(s / (size_t)noff); r->corr[m].v = hit[s]; m++; }
  r->n_corr = m;
  free(hit);
  /* :104-115  M = (C_B + T C_A T^T)^-1 restricted to 3x3 */
#pragma omp parallel for num_threads(nt) schedule(guided, 8)
  for (int j = 0; j < m; j++) {
    corr_t* co = &r->corr[j];
    const double* CA = r->src_cov + (size_t)co->src * 9;
    double RC[9], RCR[9];
    for (int a = 0; a < 3; a++)
      for (int b = 0; b < 3; b++) {
        double s = 0;
        for (int k = 0; k < 3; k++) s += T[a * 4 + k] * CA[k * 3 + b];
        RC[a * 3 + b] = s;
      }
    for (int a = 0; a < 3; a++)
      for (int b = 0; b < 3; b++) {
        double s = 0;
        for (int k = 0; k < 3; k++) s += RC[a * 3 + k] * T[b * 4 + k];
        RCR[a * 3 + b] = co->v->cov[a * 3 + b] + s;
      }
    if (inv3(RCR, co->M)) memset(co->M, 0, sizeof(co->M));
  }
}

/* fast_vgicp_impl.hpp:119-180 */
double orc_reg_linearize(orc_reg* r, const double T[16], double* H, double* b) {
  if (orc_reg_prepare(r)) return NAN;
  update_correspondences(r, T);
  r->n_lin++;
  int nt = clip_threads(r->prm.num_threads);
  double sum = 0.0;
  double* Hs = (double*)calloc((size_t)nt * 36, sizeof(double));
  double* bs = (double*)calloc((size_t)nt * 6, sizeof(double));
  const int want = (H != NULL && b != NULL);
#pragma omp parallel for num_threads(nt) reduction(+ : sum) schedule(guided, 8)
  for (int j = 0; j < r->n_corr; j++) {
    const corr_t* co = &r->corr[j];
    double p[3] = {(double)r->src[co->src * 3], (double)r->src[co->src * 3 + 1], (double)r->src[co->src * 3 + 2]}, q[3];
    xform(T, p, q);
    double e[3] = {co->v->mean[0] - q[0], co->v->mean[1] - q[1], co->v->mean[2] - q[2]}; /* :146-147 */
    double w = sqrt((double)co->v->num);                                                  /* :149 */
    const double* M = co->M;
    double Me[3] = {M[0] * e[0] + M[1] * e[1] + M[2] * e[2], M[3] * e[0] + M[4] * e[1] + M[5] * e[2], M[6] * e[0] + M[7] * e[1] + M[8] * e[2]};
    sum += w * (e[0] * Me[0] + e[1] * Me[1] + e[2] * Me[2]);                              /* :150 */
    if (!want) continue;
    /* J = [skew(q) | -I] (3x6)  :156-160 ; skewd so3/so3.hpp:21-31 */
    double J[3][6] = {{0, -q[2], q[1], -1, 0, 0}, {q[2], 0, -q[0], 0, -1, 0}, {-q[1], q[0], 0, 0, 0, -1}};
    double MJ[3][6];
    for (int a = 0; a < 3; a++)
      for (int c = 0; c < 6; c++) MJ[a][c] = M[a * 3] * J[0][c] + M[a * 3 + 1] * J[1][c] + M[a * 3 + 2] * J[2][c];
    int tid = 0;
#ifdef _OPENMP
    tid = omp_get_thread_num();
#endif
    double* Ht = Hs + (size_t)tid * 36;
    double* bt = bs + (size_t)tid * 6;
    for (int a = 0; a < 6; a++) {
      for (int c = 0; c < 6; c++) Ht[a * 6 + c] += w * (J[0][a] * MJ[0][c] + J[1][a] * MJ[1][c] + J[2][a] * MJ[2][c]); /* :162 */
      bt[a] += w * (J[0][a] * Me[0] + J[1][a] * Me[1] + J[2][a] * Me[2]);                                               /* :163 */
    }
  }
  if (want) {
    memset(H, 0, sizeof(double) * 36);
    memset(b, 0, sizeof(double) * 6);
    for (int t = 0; t < nt; t++) { /* :170-177 */
      for (int a = 0; a < 36; a++) H[a] += Hs[(size_t)t * 36 + a];
      for (int a = 0; a < 6; a++) b[a] += bs[(size_t)t * 6 + a];
    }
  }
  free(Hs);
  free(bs);
  return sum;
}

/* fast_vgicp_impl.hpp:183-204 */
double orc_reg_compute_error(orc_reg* r, const double T[16]) {
  int nt = clip_threads(r->prm.num_threads);
  (void)nt;
  double sum = 0.0;
  r->n_err++;
#pragma omp parallel for num_threads(nt) reduction(+ : sum)
  for (int j = 0; j < r->n_corr; j++) {
    const corr_t* co = &r->corr[j];
    double p[3] = {(double)r->src[co->src * 3], (double)r->src[co->src * 3 + 1], (double)r->src[co->src * 3 + 2]}, q[3];
    xform(T, p, q);
    double e[3] = {co->v->mean[0] - q[0], co->v->mean[1] - q[1], co->v->mean[2] - q[2]};
    double w = sqrt((double)co->v->num);
    const double* M = co->M;
    sum += w * (e[0] * (M[0] * e[0] + M[1] * e[1] + M[2] * e[2]) + e[1] * (M[3] * e[0] + M[4] * e[1] + M[5] * e[2]) +
                e[2] * (M[6] * e[0] + M[7] * e[1] + M[8] * e[2]));
  }
  return sum;
}

/* so3/so3.hpp:58-77 (Sophus expmap, quaternion w,x,y,z) */
void orc_so3_exp(const double omega[3], double q[4]) {
  double theta_sq = omega[0] * omega[0] + omega[1] * omega[1] + omega[2] * omega[2];
  double imag, real;
  if (theta_sq < 1e-10) {
    double theta_quad = theta_sq * theta_sq;
    imag = 0.5 - 1.0 / 48.0 * theta_sq + 1.0 / 3840.0 * theta_quad;
    real = 1.0 - 1.0 / 8.0 * theta_sq + 1.0 / 384.0 * theta_quad;
  } else {
    double theta = sqrt(theta_sq), half = 0.5 * theta;
    imag = sin(half) / theta;
    real = cos(half);
  }
  q[0] = real; q[1] = imag * omega[0]; q[2] = imag * omega[1]; q[3] = imag * omega[2];
}

/* Eigen::Quaterniond::toRotationMatrix() (no normalisation inside) */
static void quat_to_R(const double q[4], double R[9]) {
  const double w = q[0], x = q[1], y = q[2], z = q[3];
  const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
  const double twx = tx * w, twy = ty * w, twz = tz * w;
  const double txx = tx * x, txy = ty * x, txz = tz * x;
  const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
  R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
  R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
}

/* lsq_registration_impl.hpp:82-91 */
int orc_is_converged(const double d[16], double rot_eps, double trans_eps) {
  double m = 0;
  for (int a = 0; a < 3; a++)
    for (int b = 0; b < 3; b++) {
      double v = fabs(d[a * 4 + b] - (a == b ? 1.0 : 0.0)) / rot_eps;
      if (v > m) m = v;
    }
  for (int a = 0; a < 3; a++) {
    double v = fabs(d[a * 4 + 3]) / trans_eps;
    if (v > m) m = v;
  }
  return m < 1;
}

/* solve (A) x = rhs for symmetric 6x6 A.  Stands in for Eigen::LDLT (lsq_registration_impl.hpp:136-137);
 * Gaussian elimination with partial pivoting. */
static int solve6(const double A[36], const double rhs[6], double x[6]) {
  double M[6][7];
  for (int i = 0; i < 6; i++) {
    for (int j = 0; j < 6; j++) M[i][j] = A[i * 6 + j];
    M[i][6] = rhs[i];
  }
  for (int c = 0; c < 6; c++) {
    int piv = c;
    for (int r = c + 1; r < 6; r++)
      if (fabs(M[r][c]) > fabs(M[piv][c])) piv = r;
    if (M[piv][c] == 0.0) return -1;
    if (piv != c)
      for (int j = 0; j < 7; j++) { double t = M[c][j]; M[c][j] = M[piv][j]; M[piv][j] = t; }
    for (int r = c + 1; r < 6; r++) {
      double f = M[r][c] / M[c][c];
      for (int j = c; j < 7; j++) M[r][j] -= f * M[c][j];
    }
  }
  for (int i = 5; i >= 0; i--) {
    double s = M[i][6];
    for (int j = i + 1; j < 6; j++) s -= M[i][j] * x[j];
    x[i] = s / M[i][i];
  }
  return 0;
}

static void mat4_mul(const double A[16], const double B[16], double C[16]) {
  double t[16];
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) {
      double s = 0;
      for (int k = 0; k < 4; k++) s += A[i * 4 + k] * B[k * 4 + j];
      t[i * 4 + j] = s;
    }
  memcpy(C, t, sizeof(t));
}

/* lsq_registration_impl.hpp:53-79 (computeTransformation) + :125-172 (step_lm) */
int orc_reg_align(orc_reg* r, const float guess[16], float final_T[16], double final_H[36], int* converged,
                  int* lm_failed, orc_lm_trace* trace, int max_trace) {
  double x0[16];
  for (int i = 0; i < 16; i++) x0[i] = (double)guess[i];
  x0[12] = x0[13] = x0[14] = 0.0; x0[15] = 1.0; /* Eigen::Isometry3d(guess.cast<double>()) keeps the affine part */
  double lambda = -1.0;
  int conv = 0, failed = 0, iters = 0;
  r->n_lin = r->n_err = 0;
  if (final_H) { memset(final_H, 0, sizeof(double) * 36); for (int i = 0; i < 6; i++) final_H[i * 7] = 1.0; } /* :21 */
  if (orc_reg_prepare(r)) { if (lm_failed) *lm_failed = 1; if (converged) *converged = 0; return -1; }

  for (int it = 0; it < r->prm.max_iterations && !conv; it++) {
    iters = it + 1;
    double H[36], b[6], delta[16];
    double y0 = orc_reg_linearize(r, x0, H, b);             /* :128 */
    if (lambda < 0.0) {                                       /* :130-132 */
      double m = 0;
      for (int i = 0; i < 6; i++) if (fabs(H[i * 7]) > m) m = fabs(H[i * 7]);
      lambda = r->prm.lm_init_lambda_factor * m;
    }
    orc_lm_trace tr;
    memset(&tr, 0, sizeof(tr));
    tr.outer = it; tr.y0 = y0; tr.lambda_before = lambda; tr.n_corr = r->n_corr;
    double nu = 2.0;
    int ok = 0;
    for (int k = 0; k < r->prm.lm_max_iterations; k++) {     /* :135 */
      double A[36], nb[6], d[6];
      memcpy(A, H, sizeof(A));
      for (int i = 0; i < 6; i++) { A[i * 7] += lambda; nb[i] = -b[i]; }
      if (solve6(A, nb, d)) { for (int i = 0; i < 6; i++) d[i] = NAN; }
      double q[4], R[9];
      orc_so3_exp(d, q);                                      /* :139-141 */
      quat_to_R(q, R);
      memset(delta, 0, sizeof(delta));
      for (int a = 0; a < 3; a++) { for (int c = 0; c < 3; c++) delta[a * 4 + c] = R[a * 3 + c]; delta[a * 4 + 3] = d[3 + a]; }
      delta[15] = 1.0;
      double xi[16];
      mat4_mul(delta, x0, xi);                                /* :143 */
      double yi = orc_reg_compute_error(r, xi);               /* :144 */
      double den = 0;
      for (int i = 0; i < 6; i++) den += d[i] * (lambda * d[i] - b[i]);
      double rho = (y0 - yi) / den;                           /* :145 */
      tr.inner = k + 1; tr.yi = yi; tr.rho = rho;
      if (rho < 0) {                                          /* :155-163 */
        if (orc_is_converged(delta, r->prm.rotation_eps, r->prm.translation_eps)) { ok = 1; tr.accepted = 0; break; }
        lambda = nu * lambda;
        nu = 2 * nu;
        continue;
      }
      memcpy(x0, xi, sizeof(xi));                             /* :165 */
      double f = 1 - pow(2 * rho - 1, 3);
      lambda = lambda * (f > 1.0 / 3.0 ? f : 1.0 / 3.0);      /* :166 */
      if (final_H) memcpy(final_H, H, sizeof(double) * 36);   /* :167 */
      ok = 1; tr.accepted = 1;
      break;
    }
    tr.lambda_after = lambda;
    memcpy(tr.x, x0, sizeof(x0));
    if (trace && it < max_trace) trace[it] = tr;
    if (!ok) { failed = 1; break; }                           /* :69-72 "lm not converged!!" */
    conv = orc_is_converged(delta, r->prm.rotation_eps, r->prm.translation_eps); /* :74 */
  }
  for (int i = 0; i < 16; i++) final_T[i] = (float)x0[i];    /* :77 */
  if (converged) *converged = conv;
  if (lm_failed) *lm_failed = failed;
  return iters;
}

/* pcl::transformPointCloud in fp32 (lsq_registration_impl.hpp:78; SURVEY A.6 [3P-memory]) */
void orc_transform_f32(const float* pts, int n, int stride, const float T[16], float* out) {
  for (int i = 0; i < n; i++) {
    const float* p = pts + (size_t)i * stride;
    out[i * 3 + 0] = ((T[0] * p[0] + T[1] * p[1]) + T[2] * p[2]) + T[3];
    out[i * 3 + 1] = ((T[4] * p[0] + T[5] * p[1]) + T[6] * p[2]) + T[7];
    out[i * 3 + 2] = ((T[8] * p[0] + T[9] * p[1]) + T[10] * p[2]) + T[11];
  }
}

/* pcl::Registration::getFitnessScore(max_range = DBL_MAX) -- SURVEY A.6 [3P-memory];
 * called at RGC_odometer.cpp:1010.  mean squared 1-NN distance source->target, fp32 distances,
 * fp64 accumulation in source order. */
double orc_reg_fitness(orc_reg* r, const float final_T[16]) {
  if (!r->src || !r->tgt || r->n_tgt < 1) return DBL_MAX;
  if (!r->tgt_grid_ok) {
    if (grid_build(&r->tgt_grid, r->tgt, r->n_tgt, 3, 1) < 0) return DBL_MAX;
    r->tgt_grid_ok = 1;
  }
  float* tmp = (float*)malloc(sizeof(float) * 3 * (size_t)(r->n_src > 0 ? r->n_src : 1));
  orc_transform_f32(r->src, r->n_src, 3, final_T, tmp);
  float* d2 = (float*)malloc(sizeof(float) * (size_t)(r->n_src > 0 ? r->n_src : 1));
  int nt = clip_threads(r->prm.num_threads);
  (void)nt;
#pragma omp parallel for num_threads(nt) schedule(dynamic, 256)
  for (int i = 0; i < r->n_src; i++) {
    int id;
    float d;
    /* queries outside the target grid are handled by clamping inside grid_knn's ring walk */
    grid_knn(&r->tgt_grid, tmp + (size_t)i * 3, 1, &id, &d);
    d2[i] = d;
  }
  double score = 0.0;
  int nr = 0;
  for (int i = 0; i < r->n_src; i++) { score += (double)d2[i]; nr++; }
  free(tmp);
  free(d2);
  return nr > 0 ? score / nr : DBL_MAX;
}

/* ------------------------------------------------------------------------------------------
 * pcl::VoxelGrid<PointXYZI>::filter (RGC_odometer.cpp:976-991) -- SURVEY A.6 [3P-memory]:
 * leaf index = floor(p/leaf) - min_b ; idx = i + j*dx + k*dx*dy ; sort by idx ; centroid of all
 * fields per leaf (fp32 accumulation via Eigen::Vector4f-style running sum then divide) ; output
 * ordered by idx.
 * ---------------------------------------------------------------------------------------- */
typedef struct { int idx; int pt; } vg_pair;
static int cmp_vg(const void* a, const void* b) {
  const vg_pair* x = (const vg_pair*)a; const vg_pair* y = (const vg_pair*)b;
  if (x->idx != y->idx) return x->idx < y->idx ? -1 : 1;
  return x->pt < y->pt ? -1 : (x->pt > y->pt);
}

int orc_voxelgrid_filter(const float* p, int n, float leaf, float* out) {
  if (n <= 0) return 0;
  float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  for (int i = 0; i < n; i++) {
    const float* q = p + (size_t)i * 4;
    if (!isfinite(q[0]) || !isfinite(q[1]) || !isfinite(q[2])) continue;
    for (int a = 0; a < 3; a++) { if (q[a] < mn[a]) mn[a] = q[a]; if (q[a] > mx[a]) mx[a] = q[a]; }
  }
  const float inv = 1.0f / leaf;
  int minb[3], maxb[3], div[3];
  for (int a = 0; a < 3; a++) {
    minb[a] = (int)floorf(mn[a] * inv);
    maxb[a] = (int)floorf(mx[a] * inv);
    div[a] = maxb[a] - minb[a] + 1;
  }
  int64_t tot = (int64_t)div[0] * (int64_t)div[1] * (int64_t)div[2];
  if (tot > (int64_t)INT_MAX) return -1;
  vg_pair* pr = (vg_pair*)malloc(sizeof(vg_pair) * (size_t)n);
  int m = 0;
  for (int i = 0; i < n; i++) {
    const float* q = p + (size_t)i * 4;
    if (!isfinite(q[0]) || !isfinite(q[1]) || !isfinite(q[2])) continue;
    int i0 = (int)floorf(q[0] * inv) - minb[0];
    int i1 = (int)floorf(q[1] * inv) - minb[1];
    int i2 = (int)floorf(q[2] * inv) - minb[2];
    pr[m].idx = i0 + i1 * div[0] + i2 * div[0] * div[1];
    pr[m].pt = i;
    m++;
  }
  qsort(pr, (size_t)m, sizeof(vg_pair), cmp_vg);
  int no = 0;
  for (int s = 0; s < m;) {
    int e = s;
    float acc[4] = {0, 0, 0, 0};
    while (e < m && pr[e].idx == pr[s].idx) {
      const float* q = p + (size_t)pr[e].pt * 4;
      for (int a = 0; a < 4; a++) acc[a] += q[a];
      e++;
    }
    float cnt = (float)(e - s);
    for (int a = 0; a < 4; a++) out[(size_t)no * 4 + a] = acc[a] / cnt;
    no++;
    s = e;
  }
  free(pr);
  return no;
}

/* ------------------------------------------------------------------------------------------
 * f4: loop-closure ICP, pcl::IterativeClosestPoint as configured at src/RGC_mapping.cpp:2050-2069
 * (setMaxCorrespondenceDistance, setMaximumIterations(100), setTransformationEpsilon(1e-6),
 * setEuclideanFitnessEpsilon(1e-6), setRANSACIterations(0)), align() with the identity guess and getFitnessScore().
 * PCL is not installable here: the loop below restates IterativeClosestPoint::computeTransformation,
 * CorrespondenceEstimation::determineCorrespondences (1-NN, kept if d^2 <= max_dist^2), TransformationEstimationSVD
 * (Umeyama without scale) and DefaultConvergenceCriteria::hasConverged (PCL 1.8-1.10) [3P-memory].  PCL accumulates the
 * centroids and the 3x3 correlation in float in cloud order; here they are accumulated in fp64 (the GPU path sums in a
 * different order anyway) and the per-iteration transform is cast to float like PCL's Matrix4f.
 * ------------------------------------------------------------------------------------------ */
static void mat4f_mul(const float A[16], const float B[16], float C[16]) {
  float t[16];
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) {
      float s = 0.0f;
      for (int k = 0; k < 4; k++) s += A[i * 4 + k] * B[k * 4 + j];
      t[i * 4 + j] = s;
    }
  memcpy(C, t, sizeof(t));
}

/* rotation R and translation t minimising sum |R p + t - q|^2 from the sums n, sum p, sum q, sum p q^T (Kabsch / Umeyama) */
void orc_rigid_from_sums(double n, const double sp[3], const double sq[3], const double spq[9], double R[9], double t[3]) {
  double cp[3], cq[3], H[9];
  for (int a = 0; a < 3; a++) { cp[a] = sp[a] / n; cq[a] = sq[a] / n; }
  for (int a = 0; a < 3; a++)
    for (int b = 0; b < 3; b++) H[a * 3 + b] = spq[a * 3 + b] - n * cp[a] * cq[b]; /* sum (p - cp)(q - cq)^T */
  /* SVD of H through the eigen decomposition of H^T H: H = U S V^T, R = V diag(1, 1, det) U^T */
  double HtH[9] = {0}, ev[3], V[9];
  for (int a = 0; a < 3; a++)
    for (int b = 0; b < 3; b++)
      for (int k = 0; k < 3; k++) HtH[a * 3 + b] += H[k * 3 + a] * H[k * 3 + b];
  orc_eig3(HtH, ev, V); /* descending, V columns */
  double U[9];
  for (int j = 0; j < 2; j++) {
    double u[3] = {0, 0, 0};
    for (int a = 0; a < 3; a++)
      for (int k = 0; k < 3; k++) u[a] += H[a * 3 + k] * V[k * 3 + j];
    double nn = sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
    if (!(nn > 1e-300)) { /* rank deficient: any unit vector orthogonal to the previous columns */
      if (j == 0) { u[0] = 1; u[1] = 0; u[2] = 0; }
      else {
        const double a0 = fabs(U[0]), a1 = fabs(U[3]), a2 = fabs(U[6]);
        double e[3] = {a0 <= a1 && a0 <= a2 ? 1.0 : 0.0, a1 < a0 && a1 <= a2 ? 1.0 : 0.0, 0.0};
        if (e[0] == 0.0 && e[1] == 0.0) e[2] = 1.0;
        u[0] = U[3] * e[2] - U[6] * e[1]; u[1] = U[6] * e[0] - U[0] * e[2]; u[2] = U[0] * e[1] - U[3] * e[0];
      }
      nn = sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
    }
    for (int a = 0; a < 3; a++) U[a * 3 + j] = u[a] / nn;
  }
  /* third columns complete right-handed bases; the reflection fix of Umeyama is then implicit: R = V' U'^T with
   * V' = [v0 v1 v0 x v1], U' = [u0 u1 u0 x u1] is a proper rotation that agrees with V diag(1,1,det) U^T */
  U[2] = U[3] * U[7] - U[6] * U[4]; U[5] = U[6] * U[1] - U[0] * U[7]; U[8] = U[0] * U[4] - U[3] * U[1];
  double Vp[9];
  memcpy(Vp, V, sizeof(Vp));
  Vp[2] = V[3] * V[7] - V[6] * V[4]; Vp[5] = V[6] * V[1] - V[0] * V[7]; Vp[8] = V[0] * V[4] - V[3] * V[1];
  for (int a = 0; a < 3; a++)
    for (int b = 0; b < 3; b++) {
      double v = 0;
      for (int k = 0; k < 3; k++) v += Vp[a * 3 + k] * U[b * 3 + k];
      R[a * 3 + b] = v;
    }
  for (int a = 0; a < 3; a++) t[a] = cq[a] - (R[a * 3] * cp[0] + R[a * 3 + 1] * cp[1] + R[a * 3 + 2] * cp[2]);
}

int orc_icp_align(const float* src, int ns, int sstride, const float* tgt, int nt_, int tstride, const orc_icp_params* prm, float final_T[16],
                  orc_icp_result* res, int num_threads) {
  if (!src || !tgt || !prm || !final_T || !res || ns < 1 || nt_ < 1) return -1;
  grid_t g;
  if (grid_build(&g, tgt, nt_, tstride, 1) < 0) return -2;
  int nth = clip_threads(num_threads);
  (void)nth;
  float* cur = (float*)malloc(sizeof(float) * 3 * (size_t)ns);
  int* nn = (int*)malloc(sizeof(int) * (size_t)ns);
  float* d2 = (float*)malloc(sizeof(float) * (size_t)ns);
  for (int i = 0; i < ns; i++)
    for (int a = 0; a < 3; a++) cur[3 * i + a] = src[(size_t)i * sstride + a];
  float fin[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  const double max_d2 = prm->max_corr_dist * prm->max_corr_dist;
  const double rot_thr = 1.0 - prm->transformation_eps, trans_thr = prm->transformation_eps;
  double prev_mse = DBL_MAX;
  memset(res, 0, sizeof(*res));
  for (;;) {
#pragma omp parallel for num_threads(nth) schedule(dynamic, 256)
    for (int i = 0; i < ns; i++) grid_knn(&g, cur + 3 * (size_t)i, 1, &nn[i], &d2[i]);
    double cnt = 0, sp[3] = {0, 0, 0}, sq[3] = {0, 0, 0}, spq[9] = {0}, sd = 0;
    for (int i = 0; i < ns; i++) {
      if ((double)d2[i] > max_d2) continue;
      const float* p = cur + 3 * (size_t)i;
      const float* q = tgt + (size_t)nn[i] * tstride;
      cnt += 1;
      sd += (double)d2[i];
      for (int a = 0; a < 3; a++) {
        sp[a] += (double)p[a]; sq[a] += (double)q[a];
        for (int b = 0; b < 3; b++) spq[a * 3 + b] += (double)p[a] * (double)q[b];
      }
    }
    res->n_correspondences = (int)cnt;
    if (cnt < 3) { res->converged = 0; res->state = ORC_ICP_NO_CORRESPONDENCES; break; }
    double R[9], t[3];
    orc_rigid_from_sums(cnt, sp, sq, spq, R, t);
    float T[16] = {(float)R[0], (float)R[1], (float)R[2], (float)t[0], (float)R[3], (float)R[4], (float)R[5], (float)t[1],
                   (float)R[6], (float)R[7], (float)R[8], (float)t[2], 0, 0, 0, 1};
    float* nxt = (float*)malloc(sizeof(float) * 3 * (size_t)ns);
    orc_transform_f32(cur, ns, 3, T, nxt);
    free(cur);
    cur = nxt;
    mat4f_mul(T, fin, fin);
    res->iterations++;
    /* DefaultConvergenceCriteria::hasConverged */
    if (res->iterations >= prm->max_iterations) { res->converged = 1; res->state = ORC_ICP_ITERATIONS; break; }
    const double cos_angle = 0.5 * ((double)T[0] + (double)T[5] + (double)T[10] - 1.0);
    const double tr2 = (double)T[3] * T[3] + (double)T[7] * T[7] + (double)T[11] * T[11];
    if (cos_angle >= rot_thr && tr2 <= trans_thr) { res->converged = 1; res->state = ORC_ICP_TRANSFORM; break; }
    const double mse = sd / cnt;
    if (fabs(mse - prev_mse) < 1e-12) { res->converged = 1; res->state = ORC_ICP_ABS_MSE; break; }
    if (fabs(mse - prev_mse) / prev_mse < prm->fitness_eps) { res->converged = 1; res->state = ORC_ICP_REL_MSE; break; }
    prev_mse = mse;
  }
  /* getFitnessScore(): the source transformed by the final transformation (fp32), mean squared 1-NN distance */
  {
    float* tmp = (float*)malloc(sizeof(float) * 3 * (size_t)ns);
    float* s3 = (float*)malloc(sizeof(float) * 3 * (size_t)ns);
    for (int i = 0; i < ns; i++)
      for (int a = 0; a < 3; a++) s3[3 * i + a] = src[(size_t)i * sstride + a];
    orc_transform_f32(s3, ns, 3, fin, tmp);
#pragma omp parallel for num_threads(nth) schedule(dynamic, 256)
    for (int i = 0; i < ns; i++) grid_knn(&g, tmp + 3 * (size_t)i, 1, &nn[i], &d2[i]);
    double s = 0;
    for (int i = 0; i < ns; i++) s += (double)d2[i];
    res->fitness = s / (double)ns;
    free(tmp); free(s3);
  }
  memcpy(final_T, fin, sizeof(fin));
  free(cur); free(nn); free(d2);
  grid_free(&g);
  return 0;
}
