// rgc_frontend.hip -- gfx950 kernels of the scan feature + ground front-end:
// ScanRegistration::laserCloudHandler, /root/reference/rgc_slam/src/scanRegistration.cpp:89-730.
// The reference is one sequential callback; here every stage is data-parallel except the greedy per-sector pick
// (one lane per ring, after a cooperative LDS bitonic sort of each sector).
//
// fp32 stencils keep the reference's left-to-right association (compiled with -ffp-contract=off) so curvatures,
// hence labels, are bit-identical to the CPU path.  Documented clean-ups of reference quirks:
// per-frame zero-initialised state, sort ties broken by index, no 30000-point cap.
#include <limits.h>

#include "rgc_kernels.h"

namespace rgck {

constexpr int WAVE = 64;
constexpr int FE_T = 256;

__device__ __forceinline__ double fe_wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
  return v;
}

// ---- A1 + A2a: validity, ring id (:111-113, :141-183, :732-763) ----
__device__ __forceinline__ int ring_of(float x, float y, float z, int NS) {
  const float verticalAngle = (float)(atanf(z / sqrtf(x * x + y * y)) * 180 / M_PI);
  int scanID;
  if (NS == 16) {
    scanID = (int)((verticalAngle + 15) / 2 + 0.5);
    if (scanID > (NS - 1) || scanID < 0) return -1;
  } else if (NS == 32) {
    scanID = (int)((verticalAngle + 92.0 / 3.0) * 3.0 / 4.0);
    if (scanID > (NS - 1) || scanID < 0) return -1;
  } else {
    if (verticalAngle >= -8.83) scanID = (int)((2 - verticalAngle) * 3.0 + 0.5);
    else scanID = NS / 2 + (int)((-8.83 - verticalAngle) * 2.0 + 0.5);
    if (verticalAngle > 2 || verticalAngle < -24.33 || scanID > 50 || scanID < 0) return -1;
  }
  return scanID;
}

// ring[i] = -2: dropped by the NaN / range / self filter; -1: kept by them but outside the sensor's rings; >= 0: ring
// st[0] = first kept index, st[1] = last kept index (one atomicMin / atomicMax per WAVE: the first and last kept lane of its ballot --
// one pair per kept point was 23 k same-address atomics, most of this kernel's time)
// ... and, in the same launch, A2c pass 1 of the stable bucket by ring (:206-230): the rank of each point among the same-ring points of
// its block and the block's per-ring histogram.
__global__ void __launch_bounds__(FE_T)
k_fe_filter(const float* __restrict__ in, int stride_f, int n, FeParams p, int* __restrict__ ring, int* st, int* __restrict__ rank_in_block,
            int* __restrict__ blk_hist) {
  __shared__ int hist[FE_T / WAVE][64];
  const int i = blockIdx.x * FE_T + threadIdx.x;
  const int lane = threadIdx.x & (WAVE - 1), w = threadIdx.x / WAVE;
  for (int t = threadIdx.x; t < (FE_T / WAVE) * 64; t += FE_T) (&hist[0][0])[t] = 0;
  int r = -2;
  if (i < n) {
    const float* q = in + (size_t)i * stride_f;
    const float x = q[0], y = q[1], z = q[2];
    if (isfinite(x) && isfinite(y) && isfinite(z)) {
      const float dis = x * x + y * y + z * z;
      const float th1 = (float)p.min_range, th2 = (float)p.max_range;
      if (!(dis < th1 * th1) && !(dis > th2 * th2) && !(x < 0 && fabsf(y) < 0.5)) r = ring_of(x, y, z, p.n_scans);
    }
    ring[i] = r;
  }
  const unsigned long long kept = __ballot(r != -2);
  if (kept && lane == 0) {
    const int base = i;  // lane 0's index
    atomicMin(&st[0], base + __ffsll((long long)kept) - 1);
    atomicMax(&st[1], base + 63 - __clzll(kept));
  }
  __syncthreads();
  int rk = 0;
  unsigned long long todo = __ballot(r >= 0);
  while (todo) {
    const int leader = __ffsll((long long)todo) - 1;
    const int rr = __shfl(r, leader);
    const unsigned long long mask = __ballot(r == rr);
    if (r == rr) rk = __popcll(mask & ((1ull << lane) - 1ull));
    if (lane == leader) hist[w][rr] = __popcll(mask);
    todo &= ~mask;
  }
  __syncthreads();
  if (r >= 0) {
    for (int j = 0; j < w; j++) rk += hist[j][r];
    rank_in_block[i] = rk;
  }
  if (threadIdx.x < 64) {
    int sum = 0;
    for (int j = 0; j < FE_T / WAVE; j++) sum += hist[j][threadIdx.x];
    blk_hist[(size_t)blockIdx.x * 64 + threadIdx.x] = sum;
  }
}

__device__ __forceinline__ void start_end_ori(const float* __restrict__ in, int stride_f, const int* st, float& startOri, float& endOri) {
  const float* a = in + (size_t)st[0] * stride_f;
  const float* b = in + (size_t)st[1] * stride_f;
  startOri = -atan2f(a[1], a[0]);                                   // :117
  endOri = (float)(-atan2f(b[1], b[0]) + 2 * M_PI);                 // :118
  if (endOri - startOri > 3 * M_PI) endOri -= 2 * M_PI;             // :120-127
  else if (endOri - startOri < M_PI) endOri += 2 * M_PI;
}

// A2b: halfPassed becomes true AFTER the first ring-valid point whose (unwrapped) ori - startOri exceeds pi (:186-194):
// a prefix-OR, i.e. the minimum index satisfying the condition.  st[2] = that index (INT_MAX if none).
__global__ void k_fe_half(const float* __restrict__ in, int stride_f, int n, const int* __restrict__ ring, int* st) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  bool past = false;
  if (i < n && ring[i] >= 0) {
    float startOri, endOri;
    start_end_ori(in, stride_f, st, startOri, endOri);
    const float* q = in + (size_t)i * stride_f;
    float ori = -atan2f(q[1], q[0]);
    if (ori < startOri - M_PI / 2) ori += 2 * M_PI;
    else if (ori > startOri + M_PI * 3 / 2) ori -= 2 * M_PI;
    past = ori - startOri > M_PI;
  }
  const unsigned long long m = __ballot(past);  // the wave's minimum is its first such lane: one atomic per wave, not one per point of the second half
  if (m && (int)(threadIdx.x & (WAVE - 1)) == __ffsll((long long)m) - 1) atomicMin(&st[2], i);
}

// pass 2 (one block): exclusive prefix of the block histograms per ring, ring counts and ring starts.
// meta: [0..63] ring_count, [64..128] ring_start (65 entries)
// 64 rings x 16 chunks of blocks: a thread sums its chunk's histogram entries (independent loads), the chunk totals are prefixed per
// ring through LDS, then the thread writes its entries' exclusive prefixes.  (One thread per ring walking all ~110 blocks with a
// dependent load-store chain took 21 us of the front-end's 570.)
constexpr int HS_CH = 16;
__global__ void __launch_bounds__(64 * HS_CH) k_fe_hist_scan(int nblocks, int NS, int* __restrict__ blk_hist, int* __restrict__ meta) {
  __shared__ int part[HS_CH][64];
  __shared__ int cnt[64];
  const int r = threadIdx.x & 63, ch = threadIdx.x >> 6;
  const int per = (nblocks + HS_CH - 1) / HS_CH, b0 = ch * per, b1 = min(b0 + per, nblocks);
  int s = 0;
  for (int b = b0; b < b1; b++) s += blk_hist[(size_t)b * 64 + r];
  part[ch][r] = s;
  __syncthreads();
  int before = 0, total = 0;
#pragma unroll
  for (int j = 0; j < HS_CH; j++) {
    const int v = part[j][r];
    before += j < ch ? v : 0;
    total += v;
  }
  int run = before;
  for (int b = b0; b < b1; b++) {
    const int v = blk_hist[(size_t)b * 64 + r];
    blk_hist[(size_t)b * 64 + r] = run;
    run += v;
  }
  if (ch == 0) {
    cnt[r] = r < NS ? total : 0;
    meta[r] = cnt[r];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int a = 0;
    for (int j = 0; j < 64; j++) { meta[64 + j] = a; a += cnt[j]; }
    meta[128] = a;
  }
}

// pass 3: scatter into the ring-major cloud with intensity = ring + 0.1 * relTime (:196-213)
__global__ void k_fe_scatter(const float* __restrict__ in, int stride_f, int n, const int* __restrict__ ring, const int* __restrict__ rank_in_block,
                             const int* __restrict__ blk_hist, const int* __restrict__ meta, const int* __restrict__ st,
                             float4* __restrict__ C, int* __restrict__ inum2, int* __restrict__ z0, int* __restrict__ z1, int* __restrict__ z2,
                             int* __restrict__ z3) {
  const int i = blockIdx.x * FE_T + threadIdx.x;
  if (i >= n) return;
  z0[i] = 0; z1[i] = 0; z2[i] = 0; z3[i] = 0;  // the selection's flag and label arrays, zeroed per sweep (the sweep holds at most n points)
  const int r = ring[i];
  if (r < 0) return;
  float startOri, endOri;
  start_end_ori(in, stride_f, st, startOri, endOri);
  const float* q = in + (size_t)i * stride_f;
  float ori = -atan2f(q[1], q[0]);
  if (i <= st[2]) {  // !halfPassed (the point that flips the flag is still processed by this branch)
    if (ori < startOri - M_PI / 2) ori += 2 * M_PI;
    else if (ori > startOri + M_PI * 3 / 2) ori -= 2 * M_PI;
  } else {
    ori += 2 * M_PI;
    if (ori < endOri - M_PI * 3 / 2) ori += 2 * M_PI;
    else if (ori > endOri + M_PI / 2) ori -= 2 * M_PI;
  }
  const float relTime = (ori - startOri) / (endOri - startOri);
  const int d = meta[64 + r] + blk_hist[(size_t)blockIdx.x * 64 + r] + rank_in_block[i];
  C[d] = make_float4(q[0], q[1], q[2], (float)(r + 0.1 * relTime));  // scanPeriod = 0.1, :35
  inum2[d] = stride_f > 3 ? (int)q[3] : 0;                             // int point_intensity, :132,140
}

// ---- A3: range, incidence angle (:234-255) ----
// A3 / A4 / A6 in ONE launch (they were four: range + incidence angle, intensity smoothing, the curvature stencils, the occlusion mask --
// each a few microseconds of work behind a dependent-launch gap).  A workgroup stages its 256 points and ten neighbours on either side
// in LDS; the +-5 stencils of the smoothed intensity need the smoothing -- itself a +-5 stencil gated by the incidence angle, a +-5
// construction -- of the five points past the tile, hence ten.  Every value is computed by the expression the separate kernels used, in
// the same order; halo values are recomputed by the neighbouring workgroup for its own tile, identically.
//   range / incidence angle (:234-255), near-range intensity smoothing on the int intensities, truncating on every store like the
//   deque<int> (:257-268), curvature stencils (:270-306), occlusion / parallel-beam mask (:433-456; picked[] zeroed before)
__global__ void __launch_bounds__(FE_T)
k_fe_stencils(const float4* __restrict__ C, int cs_in, const int* __restrict__ csp, const int* __restrict__ inum2, float* __restrict__ range_vec,
              float* __restrict__ scan_angle, int* __restrict__ inum, float* __restrict__ curv, float* __restrict__ curv2, float* __restrict__ icurv,
              float* __restrict__ dsrc, float* __restrict__ osrc, int* __restrict__ picked) {
  constexpr int H = 10, W = FE_T + 2 * H;
  __shared__ float sx[W], sy[W], sz[W], sr[W], sa[W];
  __shared__ int si2[W], si[W];
  const int cs = csp ? min(cs_in, *csp) : cs_in;  // the sweep's size on the device (k_fe_hist_scan), cs_in = the launch's bound
  const int b0 = blockIdx.x * FE_T;
  if (b0 >= cs) return;
  for (int t = threadIdx.x; t < W; t += FE_T) {
    const int g = b0 - H + t;
    float x = 0.f, y = 0.f, z = 0.f, rg = 0.f;
    int iv = 0;
    if (g >= 0 && g < cs) {
      const float4 p4 = C[g];
      x = p4.x; y = p4.y; z = p4.z;
      rg = sqrtf(p4.x * p4.x + p4.y * p4.y + p4.z * p4.z);
      iv = inum2[g];
    }
    sx[t] = x; sy[t] = y; sz[t] = z; sr[t] = rg; si2[t] = iv;
  }
  __syncthreads();
  for (int t = 5 + threadIdx.x; t < W - 5; t += FE_T) {
    const int g = b0 - H + t;
    float v = 0.f;  // zero-initialised per frame (the reference leaks earlier frames' values here)
    if (g >= 5 && g < cs - 5 && sr[t] < 2) {
      const double a[3] = {sx[t + 5], sy[t + 5], sz[t + 5]}, b[3] = {sx[t - 5], sy[t - 5], sz[t - 5]}, p[3] = {sx[t], sy[t], sz[t]};
      const double c[3] = {(a[0] + b[0]) / 2, (a[1] + b[1]) / 2, (a[2] + b[2]) / 2};
      const double u[3] = {a[0] - b[0], a[1] - b[1], a[2] - b[2]}, w[3] = {p[0] - c[0], p[1] - c[1], p[2] - c[2]};
      const double nrm[3] = {u[1] * w[2] - u[2] * w[1], u[2] * w[0] - u[0] * w[2], u[0] * w[1] - u[1] * w[0]};
      const double nn = sqrt(nrm[0] * nrm[0] + nrm[1] * nrm[1] + nrm[2] * nrm[2]), pn = sqrt(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]);
      v = (float)((nrm[0] * p[0] + nrm[1] * p[1] + nrm[2] * p[2]) / (nn * pn));
      if (v < 0) v = -v;
    }
    sa[t] = v;
  }
  __syncthreads();
  for (int t = 5 + threadIdx.x; t < W - 5; t += FE_T) {
    const int g = b0 - H + t;
    int v = si2[t];
    if (g >= 5 && g < cs - 5 && sa[t] < 0.07 && sr[t] < 2) {
      v = (int)(0.9 * si2[t]);
      for (int j = -5; j < 6; j++)
        if (j != 0) v = (int)(v + 0.005 * si2[t + j]);
    }
    si[t] = v;
  }
  __syncthreads();
  const int i = b0 + threadIdx.x, t = H + threadIdx.x;
  if (i >= cs) return;
  range_vec[i] = sr[t];
  scan_angle[i] = sa[t];
  inum[i] = si[t];
  float cv = 0.f, cv2 = 0.f, icv = 0.f, ds = 0.f, os = 0.f;
  if (i >= 5 && i < cs - 5) {
    float dX = sx[t - 5], dY = sy[t - 5], dZ = sz[t - 5];
#pragma unroll
    for (int k = -4; k <= -1; k++) { dX = dX + sx[t + k]; dY = dY + sy[t + k]; dZ = dZ + sz[t + k]; }
    dX = dX - 10 * sx[t]; dY = dY - 10 * sy[t]; dZ = dZ - 10 * sz[t];
#pragma unroll
    for (int k = 1; k <= 5; k++) { dX = dX + sx[t + k]; dY = dY + sy[t + k]; dZ = dZ + sz[t + k]; }
    const int dIi = si[t - 5] + si[t - 4] + si[t - 3] + si[t - 2] + si[t - 1] - 10 * si[t] + si[t + 1] + si[t + 2] + si[t + 3] + si[t + 4] + si[t + 5];
    const float diffI = (float)dIi;
    float dis_factor = (float)(2.0 / (1.0 + sr[t] / 20.0));
    if (dis_factor < 0.2) dis_factor = 0.2f;
    cv = (dX * dX + dY * dY + dZ * dZ) * dis_factor;
    ds = (float)(0.5 + dis_factor);
    if (sa[t] < 0.07 && sr[t] < 2) {
      os = (float)(sa[t] * 10 + 0.6);
      icv = (float)((sa[t] + 0.3) * diffI);
    } else {
      os = 3;
      icv = diffI;
    }
    const float dr = (float)(sr[t - 5] + sr[t - 4] + sr[t - 3] + sr[t - 2] + sr[t - 1] - 10.0 * sr[t] + sr[t + 1] + sr[t + 2] + sr[t + 3] + sr[t + 4] + sr[t + 5]);
    cv2 = fabsf(dr * dis_factor);
    // occlusion / parallel beams
    const float d1 = sr[t], d2 = sr[t + 1];
    if (d1 - d2 > 0.04 * d2) {
      for (int k = -5; k <= 0; k++) picked[i + k] = 1;
    } else if (d2 - d1 > 0.04 * d1) {
      for (int k = 1; k <= 6; k++) if (i + k < cs) picked[i + k] = 1;
    }
  }
  curv[i] = cv; curv2[i] = cv2; icurv[i] = icv; dsrc[i] = ds; osrc[i] = os;
}

// ---- A5: ground marking (:308-353) ----
__device__ __forceinline__ int ring_of_index(const int* __restrict__ meta, int NS, int i) {  // meta[64..] ring starts
  int r = 0;
  for (int j = 1; j < NS; j++) r += (meta[64 + j] <= i);
  return r;
}
__constant__ float kGroundScanRange[16] = {2.66f, 3.04f, 3.56f, 4.30f, 5.44f, 7.41f, 11.63f, 27.12f, 0, 0, 0, 0, 0, 0, 0, 0};  // :40

__device__ __forceinline__ bool ground_seed(const float4* __restrict__ C, const float* __restrict__ range_vec, const int* __restrict__ meta,
                                            int ring, int c) {
  const int s0 = meta[64 + ring], sz = meta[ring];
  const int col = c - s0;
  if (sz < 11 || col < 5 || col >= sz - 5) return false;
  const float th = (float)(0.8 * (1.0 + ring / 6));  // integer division, :323
  const float dr = fabsf(range_vec[c] - kGroundScanRange[ring]);
  return dr < th && C[c].z < 0.3;
}

// mult[j] = how many times point j is pushed into the ground set (seed c = j - n, n in [-5, 4]); seedcnt[c] = pushes
// made by seed c (for the ordered ground list).  acc: block partial sums {W, Wx, Wy, Wz, Wxx, Wxy, Wxz, Wyy, Wyz, Wzz, count}
__global__ void __launch_bounds__(FE_T)
k_fe_ground(const float4* __restrict__ C, int cs_in, const int* __restrict__ csp, int NS, const float* __restrict__ range_vec, const int* __restrict__ meta,
            int* __restrict__ gmark, int* __restrict__ mult_out, int* __restrict__ seedcnt, double* __restrict__ partials) {
  const int cs = csp ? min(cs_in, *csp) : cs_in;  // the sweep's size on the device (k_fe_hist_scan), cs_in = the launch's bound
  const int j = blockIdx.x * FE_T + threadIdx.x;
  double acc[11];
#pragma unroll
  for (int a = 0; a < 11; a++) acc[a] = 0.0;
  int mult = 0, sc = 0;
  const int gend = meta[64 + (NS < 7 ? NS : 7)];  // rings 0..6 only (groundScanInd = 7, :34)
  if (j < cs && j < gend) {
    const int ring = ring_of_index(meta, NS, j);
    const float th = (float)(0.8 * (1.0 + ring / 6));
    const float rj = range_vec[j];
    for (int nn = -5; nn < 5; nn++) {       // j = c + nn
      const int c = j - nn;
      if (c < 0 || c >= cs) continue;
      if (ground_seed(C, range_vec, meta, ring, c) && fabsf(rj - range_vec[c]) < th / 2) mult++;
    }
    if (ground_seed(C, range_vec, meta, ring, j)) {
      for (int nn = -5; nn < 5; nn++)
        if (fabsf(range_vec[j + nn] - rj) < th / 2) sc++;
    }
    if (mult > 0) {
      const double w = (1.5 - ring / 6) * (double)mult;  // groundweight, :325
      const float4 p = C[j];
      const double x = p.x, y = p.y, z = p.z;
      acc[0] = w; acc[1] = w * x; acc[2] = w * y; acc[3] = w * z;
      acc[4] = w * x * x; acc[5] = w * x * y; acc[6] = w * x * z; acc[7] = w * y * y; acc[8] = w * y * z; acc[9] = w * z * z;
      acc[10] = (double)mult;
    }
  }
  if (j < cs) { gmark[j] = mult > 0 ? 1 : 0; mult_out[j] = mult; seedcnt[j] = sc; }
  else if (j < cs_in) seedcnt[j] = 0;  // the scan of the seed counts runs over the launch's bound
  __shared__ double red[FE_T / WAVE][11];
  const int lane = threadIdx.x & (WAVE - 1), w = threadIdx.x / WAVE;
#pragma unroll
  for (int a = 0; a < 11; a++) {
    const double v = fe_wave_sum(acc[a]);
    if (lane == 0) red[w][a] = v;
  }
  __syncthreads();
  if (threadIdx.x < 11) {
    double s = 0;
    for (int t = 0; t < FE_T / WAVE; t++) s += red[t][threadIdx.x];
    partials[(size_t)blockIdx.x * 11 + threadIdx.x] = s;
  }
}

// distance pass (:386-402): sums {sum dw, sum dw * n.p} with multiplicity
__global__ void __launch_bounds__(FE_T)
k_fe_ground_dist(const float4* __restrict__ C, int cs_in, const int* __restrict__ csp, const int* __restrict__ mult, const double* __restrict__ fit, double* __restrict__ partials) {
  // fit: centre (3), normal (3), eigenvectors (9), ground present (1) -- written by k_fe_fold_fit, never seen by the host in between
  const int cs = csp ? min(cs_in, *csp) : cs_in;  // the sweep's size on the device (k_fe_hist_scan), cs_in = the launch's bound
  const int j = blockIdx.x * FE_T + threadIdx.x;
  double a0 = 0, a1 = 0;
  if (fit[15] != 0.0 && j < cs && mult[j] > 0) {
    const double cx = fit[0], cy = fit[1], cz = fit[2], nx = fit[3], ny = fit[4], nz = fit[5];
    const float4 p = C[j];
    const double d[3] = {(double)p.x - cx, (double)p.y - cy, (double)p.z - cz};
    const double dl = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    double dw = dl == 0 ? 1.0 : 1 - 100 * fabs((nx * d[0] + ny * d[1] + nz * d[2]) / dl);
    if (dw < 0) dw = 0.1;
    a0 = dw * mult[j];
    a1 = dw * mult[j] * (nx * (double)p.x + ny * (double)p.y + nz * (double)p.z);
  }
  __shared__ double red[FE_T / WAVE][2];
  const int lane = threadIdx.x & (WAVE - 1), w = threadIdx.x / WAVE;
  a0 = fe_wave_sum(a0); a1 = fe_wave_sum(a1);
  if (lane == 0) { red[w][0] = a0; red[w][1] = a1; }
  __syncthreads();
  if (threadIdx.x < 2) {
    double s = 0;
    for (int t = 0; t < FE_T / WAVE; t++) s += red[t][threadIdx.x];
    partials[(size_t)blockIdx.x * 2 + threadIdx.x] = s;
  }
}

// Plane through the weighted ground set (:358-377) on the device, so that the host does not have to see the sums before the distance
// pass can start: weighted centroid, covariance, symmetric eigen-decomposition (cyclic Jacobi, eigenvalues ascending like
// Eigen::SelfAdjointEigenSolver), normal = the smallest eigenvector oriented towards the centroid.  One lane; ~2 us.
// g11: {W, Wx, Wy, Wz, Wxx, Wxy, Wxz, Wyy, Wyz, Wzz, count};  fit: centre (3), normal (3), V row-major (9), present (1)
__device__ void fe_ground_fit(const double* g11, double* __restrict__ fit) {
  const long long gsize = (long long)(g11[10] + 0.5);
  fit[15] = gsize > 0 ? 1.0 : 0.0;
  if (gsize <= 0) return;
  const double W = g11[0];
  const double ctr[3] = {g11[1] / W, g11[2] / W, g11[3] / W};
  double A[3][3], U[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
  A[0][0] = g11[4] / W - ctr[0] * ctr[0]; A[0][1] = A[1][0] = g11[5] / W - ctr[0] * ctr[1]; A[0][2] = A[2][0] = g11[6] / W - ctr[0] * ctr[2];
  A[1][1] = g11[7] / W - ctr[1] * ctr[1]; A[1][2] = A[2][1] = g11[8] / W - ctr[1] * ctr[2]; A[2][2] = g11[9] / W - ctr[2] * ctr[2];
  for (int sweep = 0; sweep < 60; sweep++) {
    const double off = A[0][1] * A[0][1] + A[0][2] * A[0][2] + A[1][2] * A[1][2];
    const double dg = A[0][0] * A[0][0] + A[1][1] * A[1][1] + A[2][2] * A[2][2];
    if (off <= 1e-40 * dg || off == 0.0) break;
#pragma unroll
    for (int p = 0; p < 2; p++)
#pragma unroll
      for (int q = p + 1; q < 3; q++) {
        if (A[p][q] == 0.0) continue;
        const double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double cc = 1.0 / sqrt(t * t + 1.0), ss = t * cc;
#pragma unroll
        for (int k = 0; k < 3; k++) { const double a = A[k][p], b = A[k][q]; A[k][p] = cc * a - ss * b; A[k][q] = ss * a + cc * b; }
#pragma unroll
        for (int k = 0; k < 3; k++) { const double a = A[p][k], b = A[q][k]; A[p][k] = cc * a - ss * b; A[q][k] = ss * a + cc * b; }
#pragma unroll
        for (int k = 0; k < 3; k++) { const double a = U[k][p], b = U[k][q]; U[k][p] = cc * a - ss * b; U[k][q] = ss * a + cc * b; }
      }
  }
  // columns in ascending eigenvalue order (a three-element sorting network on the column indices, static indexing only)
  double e0 = A[0][0], e1 = A[1][1], e2 = A[2][2];
  double c0[3] = {U[0][0], U[1][0], U[2][0]}, c1[3] = {U[0][1], U[1][1], U[2][1]}, c2[3] = {U[0][2], U[1][2], U[2][2]};
  auto cswap = [](double& ea, double& eb, double (&ca)[3], double (&cb)[3]) {
    if (eb < ea) { double t = ea; ea = eb; eb = t; for (int k = 0; k < 3; k++) { t = ca[k]; ca[k] = cb[k]; cb[k] = t; } }
  };
  cswap(e0, e1, c0, c1); cswap(e0, e2, c0, c2); cswap(e1, e2, c1, c2);
  double nrm[3] = {c0[0], c0[1], c0[2]};
  const double nl = sqrt(nrm[0] * nrm[0] + nrm[1] * nrm[1] + nrm[2] * nrm[2]);
  for (int a = 0; a < 3; a++) nrm[a] /= nl;
  if (ctr[0] * nrm[0] + ctr[1] * nrm[1] + ctr[2] * nrm[2] < 0) for (int a = 0; a < 3; a++) nrm[a] = -nrm[a];  // :374-377
  for (int a = 0; a < 3; a++) { fit[a] = ctr[a]; fit[3 + a] = nrm[a]; fit[6 + a * 3] = c0[a]; fit[6 + a * 3 + 1] = c1[a]; fit[6 + a * 3 + 2] = c2[a]; }
}

// the fold of k_fe_ground's block sums (eleven columns, one wave each, the fixed order of k_fe_fold) and the plane fit in ONE launch
__global__ void __launch_bounds__(11 * WAVE) k_fe_fold_fit(const double* __restrict__ partials, int nrows, double* __restrict__ out11, double* __restrict__ fit) {
  __shared__ double g[11];
  const int a = threadIdx.x / WAVE, lane = threadIdx.x & (WAVE - 1);
  double sum = 0;
  for (int r = lane; r < nrows; r += WAVE) sum += partials[(size_t)r * 11 + a];
  sum = fe_wave_sum(sum);
  if (lane == 0) { out11[a] = sum; g[a] = sum; }
  __syncthreads();
  if (threadIdx.x == 0) fe_ground_fit(g, fit);
}

// fixed-order fold of per-block rows (deterministic)
__global__ void __launch_bounds__(WAVE) k_fe_fold(const double* __restrict__ partials, int nrows, int ncols, double* __restrict__ out) {
  const int a = blockIdx.x, lane = threadIdx.x;
  double s = 0;
  for (int r = lane; r < nrows; r += WAVE) s += partials[(size_t)r * ncols + a];
  s = fe_wave_sum(s);
  if (lane == 0) out[a] = s;
}

// ground points with duplicates in the reference's push order (/laser_cloud_ground, :336)
__global__ void k_fe_ground_list(const float4* __restrict__ C, int cs_in, const int* __restrict__ csp, int NS, const float* __restrict__ range_vec, const int* __restrict__ meta,
                                 const int* __restrict__ seedcnt, const int* __restrict__ seedpos, float4* __restrict__ out, int cap) {
  const int cs = csp ? min(cs_in, *csp) : cs_in;  // the sweep's size on the device (k_fe_hist_scan), cs_in = the launch's bound
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= cs || seedcnt[c] == 0) return;
  const int ring = ring_of_index(meta, NS, c);
  const float th = (float)(0.8 * (1.0 + ring / 6));
  int pos = seedpos[c];
  const float rc = range_vec[c];
  for (int nn = -5; nn < 5; nn++) {
    if (fabsf(range_vec[c + nn] - rc) < th / 2) {
      if (pos < cap) out[pos] = C[c + nn];
      pos++;
    }
  }
}

// ---- A7: per ring, six sectors: sort by curvature / intensity curvature, greedy pick with +-5 suppression (:469-644) ----
constexpr int SEC_MAX = 2048;   // points per sector (ring of <= 12288 points)
constexpr int SEL_T = 256;
constexpr int SELP_T = 384;  // k_fe_select: six waves, one per sector of the ring

struct Key { float v; int i; };
__device__ __forceinline__ bool key_less(const Key& a, const Key& b) { return a.v < b.v || (a.v == b.v && a.i < b.i); }

__device__ void bitonic_sort(Key* k, int n2) {  // ascending by (value, index); n2 power of two; whole block
  for (int size = 2; size <= n2; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      __syncthreads();
      for (int t = threadIdx.x; t < n2 / 2; t += SEL_T) {
        const int lo = (t / stride) * (stride * 2) + (t % stride), hi = lo + stride;
        const bool up = ((lo & size) == 0);
        const Key a = k[lo], b = k[hi];
        if (key_less(b, a) == up) { k[lo] = b; k[hi] = a; }
      }
    }
  }
  __syncthreads();
}

// The two sorts of every (ring, sector) -- by curvature and by intensity curvature (:482-483) -- depend on nothing the greedy passes
// change, so they do not belong on the 16 workgroups that must walk their ring's sectors in order: one workgroup per (ring, sector,
// array), all at once, the sorted point indices left in memory for k_fe_select (the two sorts used to be half of that kernel's time).
__global__ void __launch_bounds__(SEL_T)
k_fe_sort(int NS, const int* __restrict__ meta, const float* __restrict__ curv, const float* __restrict__ icurv, int* __restrict__ sorted_curv,
          int* __restrict__ sorted_icurv) {
  __shared__ Key k[SEC_MAX];
  const int ring = blockIdx.x / 6, j = blockIdx.x % 6;
  const int S = meta[64 + ring] + 5, E = meta[64 + ring + 1] - 5;
  if (E - S < 10) return;
  const int sp = S + (E - S) * j / 6, ep = S + (E - S) * (j + 1) / 6 - 1;
  const int cnt = ep - sp + 1;
  if (cnt > SEC_MAX || cnt <= 0) return;  // k_fe_select reports the oversize sector
  const float* __restrict__ key = blockIdx.y ? icurv : curv;
  int* __restrict__ out = blockIdx.y ? sorted_icurv : sorted_curv;
  int n2 = 1;
  while (n2 < cnt) n2 <<= 1;
  for (int t = threadIdx.x; t < n2; t += SEL_T) k[t] = t < cnt ? Key{key[sp + t], sp + t} : Key{INFINITY, INT_MAX};
  bitonic_sort(k, n2);
  for (int t = threadIdx.x; t < cnt; t += SEL_T) out[sp + t] = k[t].i;
}

// slots: per (ring, sector): sharp 20, flat 40, inten 20 indices + 3 counts (83 ints)
constexpr int SLOT = 83;
constexpr int SEL_MARGIN = 5;  // a pick touches ind +- 5 (and compares ind + l with ind + l -+ 1, both inside +- 5)

// One workgroup per ring; the sectors of a ring are processed in order (a pick suppresses up to five neighbours, possibly
// across a sector border).  Per sector: every lane stages the sector's window (points, curvatures, flags; +-5 points of
// margin) in LDS and the block sorts the two key arrays; then ONE lane runs the three greedy passes of :487-641 entirely out of
// LDS -- the passes are inherently sequential (each pick changes what the next candidate may be), and run from global memory
// they were a chain of ~10 dependent loads per pick (1.4 ms per sweep); finally the block writes the flags back.
// Dynamic LDS: n2 keys x 2, then the window arrays (host sizes it from the largest ring).
__global__ void __launch_bounds__(SELP_T)
k_fe_select(const float4* __restrict__ C, int NS, const int* __restrict__ meta, const float* __restrict__ curv, const float* __restrict__ curv2,
            const float* __restrict__ icurv, const int* __restrict__ inum, const int* __restrict__ gmark, int* __restrict__ picked,
            int* __restrict__ ipicked, int* __restrict__ label, int* __restrict__ ilabel, int* __restrict__ slots, int* flags, int sec_cap,
            const int* __restrict__ sorted_curv, const int* __restrict__ sorted_icurv, int group) {
  // sec_cap: points of `group` consecutive sectors (6, 3, 2 or 1: as many as the LDS holds) -- their window is staged, and written
  // back, ONCE, and the serial passes walk the group's sectors in order out of it (staging per sector was a third of the kernel)
  extern __shared__ __align__(16) unsigned char sel_lds[];
  const int wcap = sec_cap + 2 * SEL_MARGIN;
  int* ks = reinterpret_cast<int*>(sel_lds);   // point indices of the sector in ascending curvature (k_fe_sort)
  int* ki = ks + sec_cap;                      // ... in ascending intensity curvature
  float* wx = reinterpret_cast<float*>(ki + sec_cap);
  float* wy = wx + wcap;
  float* wz = wy + wcap;
  float* wc = wz + wcap;    // curvature
  float* wc2 = wc + wcap;   // curvature2
  float* wic = wc2 + wcap;  // intensity curvature
  int* wnum = reinterpret_cast<int*>(wic + wcap);
  signed char* wpick = reinterpret_cast<signed char*>(wnum + wcap);
  signed char* wipick = wpick + wcap;
  signed char* wgm = wipick + wcap;
  signed char* wlab = wgm + wcap;
  signed char* wilab = wlab + wcap;
  signed char* wrun = wilab + wcap;   // how far a pick at this point suppresses: points to the right | points to the left << 4 (:517-533)
  signed char* wirun = wrun + wcap;   // the same for the intensity pass (:625-639)
  signed char* wpick0 = wirun + wcap;  // the two flag arrays as staged (six sectors at once: a sector that must start over)
  signed char* wipick0 = wpick0 + wcap;
  signed char* xp = wipick0 + wcap;    // marks dropped outside the marking wave's sector
  signed char* xip = xp + wcap;
  const int ring = blockIdx.x;
  const int S = meta[64 + ring] + 5, E = meta[64 + ring + 1] - 5;  // scanStartInd / scanEndInd, :223,229
  for (int j = 0; j < 6; j++) {
    int* sl = slots + ((size_t)ring * 6 + j) * SLOT;
    if (threadIdx.x < 3) sl[80 + threadIdx.x] = 0;
  }
  if (E - S < 10) return;  // :471
#ifdef RGC_LAB
  long long lab_t[5] = {0, 0, 0, 0, 0}, lab_prev = wall_clock64();
  __shared__ int lab_redo;
  if (threadIdx.x == 0) lab_redo = 0;
#define FE_LAB(k) do { const long long now_ = wall_clock64(); lab_t[k] += now_ - lab_prev; lab_prev = now_; } while (0)
#else
#define FE_LAB(k)
#endif
  for (int j0 = 0; j0 < 6; j0 += group) {
    const int sp0 = S + (E - S) * j0 / 6, epl = S + (E - S) * (j0 + group) / 6 - 1;  // the group's first and last point
    const int gcnt = epl - sp0 + 1;
    bool oversize = gcnt > sec_cap;
    for (int j = j0; j < j0 + group; j++) oversize |= (S + (E - S) * (j + 1) / 6 - 1) - (S + (E - S) * j / 6) + 1 > SEC_MAX;
    if (oversize) { if (threadIdx.x == 0) atomicOr(flags, 2); return; }
    const int w0 = sp0 - SEL_MARGIN, wn = gcnt + 2 * SEL_MARGIN;  // window [w0, w0 + wn): inside this ring (S = start + 5)
    __syncthreads();
    for (int t = threadIdx.x; t < wn; t += SELP_T) {
      const int g = w0 + t;
      const float4 p = C[g];
      wx[t] = p.x; wy[t] = p.y; wz[t] = p.z;
      wc[t] = curv[g]; wc2[t] = curv2[g]; wic[t] = icurv[g];
      wnum[t] = inum[g];
      wpick[t] = (signed char)picked[g]; wipick[t] = (signed char)ipicked[g];
      wgm[t] = (signed char)(gmark[g] == 1);
      wlab[t] = (signed char)label[g]; wilab[t] = (signed char)ilabel[g];
    }
    for (int t = threadIdx.x; t < gcnt; t += SELP_T) { ks[t] = sorted_curv[sp0 + t]; ki[t] = sorted_icurv[sp0 + t]; }  // sorted sector by sector
    __syncthreads();
    FE_LAB(0);
    // The reach of a pick's suppression is a property of the points alone (consecutive gaps <= 0.05 m^2, resp. intensity steps <= 35,
    // up to five on each side): computed here for every point of the sector by the whole workgroup, so that the serial pick loop
    // below reads one byte instead of testing ten gaps per pick.
    for (int t = SEL_MARGIN + threadIdx.x; t < SEL_MARGIN + gcnt; t += SELP_T) {
      auto far_p = [&](int a, int b) {
        const float dx = wx[a] - wx[b], dy = wy[a] - wy[b], dz = wz[a] - wz[b];
        return dx * dx + dy * dy + dz * dz > 0.05;
      };
      auto far_i = [&](int a, int b) { return fabsf((float)(wnum[a] - wnum[b])) > 35; };
      int rp = 0, rm = 0, ip = 0, im = 0;
      while (rp < 5 && !far_p(t + rp + 1, t + rp)) rp++;
      while (rm < 5 && !far_p(t - rm - 1, t - rm)) rm++;
      while (ip < 5 && !far_i(t + ip + 1, t + ip)) ip++;
      while (im < 5 && !far_i(t - im - 1, t - im)) im++;
      wrun[t] = (signed char)(rp | (rm << 4));
      wirun[t] = (signed char)(ip | (im << 4));
    }
    __syncthreads();
    FE_LAB(1);
    // one sector's three passes by the calling wave; own_only: marks that fall outside the sector go to xp / xip instead of the flags
    auto run_sector = [&](int j, bool own_only) {
      const int sp = S + (E - S) * j / 6, ep = S + (E - S) * (j + 1) / 6 - 1;  // :478-480
      const int cnt = ep - sp + 1;
      const int* const ksj = ks + (sp - sp0);
      const int* const kij = ki + (sp - sp0);
      // The three greedy passes of :487-641 by ONE WAVE.  A pass walks the sorted candidates in order and a pick suppresses
      // up to ten neighbours, so picks are sequential -- but only picks: 64 candidates at a time, every lane tests the
      // static conditions of its candidate, then the earliest candidate that is still unsuppressed is picked (ballot +
      // find-first), its lane marks the neighbours in LDS, and the remaining lanes look again.  Serial steps = picks
      // (<= 21 / 40 / 21 per sector), not candidates (hundreds).
      const int lane = threadIdx.x & (WAVE - 1);
      const int tlo = own_only ? sp - w0 : 0, thi = own_only ? ep - w0 : wn - 1;   // the window positions whose flags this wave owns
      int* sl = slots + ((size_t)ring * 6 + j) * SLOT;
      auto wave_fence = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
      };
      // run: the precomputed reach of a pick's suppression at every window point (:517-533 and twins).  Returns the picks made.
      const bool mark_right = lane < 5;
      const int mark_off = lane < 5 ? lane + 1 : -(lane - 4);           // lanes 0..4: +1..+5, lanes 5..9: -1..-5
      const int mark_rank = lane < 5 ? lane : (lane < 10 ? lane - 5 : 99);  // how far out this lane's mark is (99: this lane marks nothing)
      auto greedy = [&](const int* keys, bool descending, signed char* flag, signed char* xflag, int limit, auto&& static_ok, auto&& on_pick, const signed char* run) {
        int count = 0;
        bool stop = false;
        for (int base = 0; base < cnt && !stop; base += WAVE) {
          const int kk = base + lane;
          const bool valid = kk < cnt;
          const int k = valid ? (descending ? cnt - 1 - kk : kk) : 0;
          const int ind = keys[k], w = ind - w0;
          const bool ok = valid && static_ok(w);
          const int reach = run[w];
          // This candidate's suppression state lives in a REGISTER for the batch: a pick marks window positions wb - nm .. wb + np, and
          // every lane sees whether its own candidate is among them by comparing -- no LDS read (and no wait for the marks to land)
          // inside the pick loop, which was two LDS round trips per pick (~340 cycles; 70 of the kernel's 138 us were this loop).
          // The marks still go to LDS for the following batches and passes; one fence per batch.
          int mine = flag[w];
          // (the LDS reads above are waited for HERE: left to the compiler, the wait lands in the loop header below and every pick then
          // also waits for the previous pick's LDS stores to land)
          __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
          for (;;) {
            const unsigned long long mask = __ballot(ok && mine == 0);
            if (!mask) break;
            count++;
            if (count > limit) { stop = true; break; }  // the reference's `else break`: not even marked
            const int b = __builtin_amdgcn_readfirstlane(__ffsll((long long)mask) - 1);  // wave-uniform: v_readlane, not an LDS permute
            const int wb = __builtin_amdgcn_readlane(w, b);
            if (lane == b) {
              on_pick(count, ind, w);
              flag[w] = 1;
            }
            // the ten neighbours side by side: lanes 0..4 mark wb+1..wb+5, lanes 5..9 wb-1..wb-5, as far as the pick reaches
            // (mark_off / mark_rank are per-lane constants: one select, one compare and one masked byte store per pick)
            const int rb = __builtin_amdgcn_readlane(reach, b);
            const int np = rb & 15, nm = rb >> 4;
            if (mark_rank < (mark_right ? np : nm)) {
              const int q = wb + mark_off;
              if (q >= tlo && q <= thi) flag[q] = 1; else xflag[q] = 1;
            }
            if (w >= wb - nm && w <= wb + np) mine = 1;
          }
          wave_fence();
        }
        return count > limit ? limit : count;
      };
      // sharp: largest curvature first (:487-536); the 21st pick is labelled "less sharp" and gets no slot
      const int nsh_picks = greedy(ksj, true, wpick, xp, 21, [&](int w) { return wgm[w] == 0 && wc[w] > 0.1 && wc2[w] > 0.3; },
                                   [&](int count, int ind, int w) { if (count <= 20) { wlab[w] = 2; sl[count - 1] = ind; } else { wlab[w] = 1; } },
                                   wrun);
      const int nsh = nsh_picks > 20 ? 20 : nsh_picks;
      wave_fence();
      // flat: smallest curvature first (:540-583)
      const int nfl = greedy(ksj, false, wpick, xp, 40, [&](int w) { return wc[w] < 0.3 && wc2[w] < 0.4; },
                             [&](int count, int ind, int w) { wlab[w] = -1; sl[20 + count - 1] = ind; }, wrun);
      wave_fence();
      // intensity: largest intensity curvature first, not on points already labelled sharp (:594-641)
      const int nin_picks = greedy(kij, true, wipick, xip, 21, [&](int w) { return wgm[w] == 0 && wic[w] > 65 && wlab[w] != 2 && wlab[w] != 1; },
                                   [&](int count, int ind, int w) { if (count <= 20) { wilab[w] = 2; sl[60 + count - 1] = ind; } else { wilab[w] = 1; } },
                                   wirun);
      if (lane == 0) { sl[80] = nsh; sl[81] = nfl; sl[82] = nin_picks > 20 ? 20 : nin_picks; }
      wave_fence();
    };
    const int wv = threadIdx.x / WAVE, ln = threadIdx.x & (WAVE - 1);
    bool parallel = group == 6;
    for (int j = 0; j < 6 && parallel; j++) parallel = (S + (E - S) * (j + 1) / 6 - 1) - (S + (E - S) * j / 6) + 1 >= 12;  // marks reach one sector only
    if (!parallel) {
      if (wv == 0) for (int j = j0; j < j0 + group; j++) run_sector(j, false);
    } else {
      // THE SIX SECTORS AT ONCE, one wave each.  A sector's passes depend on the sectors before it only through the marks their picks
      // drop on its first five points, and such a mark changes the outcome only if it lands on a point this sector picked (a point
      // that was not picked -- failed the thresholds, already suppressed, behind the quota -- stays unpicked).  So: every wave runs its
      // sector on the flags as staged, marks that leave the sector are kept aside (xp / xip); then, in ring order, a sector that finds
      // one of its picks marked by its (final) predecessor starts over from the staged flags plus those marks.  Same picks, same
      // order, same flags as the serial walk; a 16-beam ring of six sectors takes one sector's time plus the repeats (about a third).
      for (int t = threadIdx.x; t < wn; t += SELP_T) { wpick0[t] = wpick[t]; wipick0[t] = wipick[t]; xp[t] = 0; xip[t] = 0; }
      __syncthreads();
      if (wv < 6) run_sector(wv, true);
      __syncthreads();
      FE_LAB(4);
      for (int j = 1; j < 6; j++) {
        if (wv == j) {
          const int sp = S + (E - S) * j / 6, ep = S + (E - S) * (j + 1) / 6 - 1;
          const int tlo = sp - w0, thi = ep - w0;
          const int q = tlo + (ln < 5 ? ln : 0);
          const bool hit = ln < 5 && ((xp[q] && !wpick0[q] && wlab[q] != 0) || (xip[q] && !wipick0[q] && wilab[q] != 0));
          const bool redo = __any(hit);
          if (redo) {
            for (int t = tlo + ln; t <= thi; t += WAVE) { wpick[t] = wpick0[t]; wipick[t] = wipick0[t]; wlab[t] = 0; wilab[t] = 0; }
            if (ln < 5) { xp[thi + 1 + ln] = 0; xip[thi + 1 + ln] = 0; xp[tlo - 1 - ln] = 0; xip[tlo - 1 - ln] = 0; }   // this sector's own marks outside it
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
          }
          if (ln < 5) { wpick[q] |= xp[q]; wipick[q] |= xip[q]; }   // the predecessor's marks
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
          __builtin_amdgcn_wave_barrier();
          if (redo) run_sector(j, true);
#ifdef RGC_LAB
          if (redo && ln == 0) atomicAdd(&lab_redo, 1);
#endif
        }
        __syncthreads();
      }
      for (int t = threadIdx.x; t < wn; t += SELP_T) { wpick[t] |= xp[t]; wipick[t] |= xip[t]; }   // every mark that left its sector
    }
    __syncthreads();
    FE_LAB(2);
    for (int t = threadIdx.x; t < wn; t += SELP_T) {  // flags back to memory: the next sector's window overlaps this one's margin
      const int g = w0 + t;
      picked[g] = wpick[t]; ipicked[g] = wipick[t];
      label[g] = wlab[t]; ilabel[g] = wilab[t];
    }
    FE_LAB(3);
  }
#ifdef RGC_LAB
  if (threadIdx.x == 0 && (ring == 3 || ring == 12)) printf("k_fe_select ring 3 / 12 (10 ns units): load %lld reach %lld six sectors at once %lld in-order pass %lld (%d sectors started over) writeback %lld\n", lab_t[0], lab_t[1], lab_t[4], lab_t[2], lab_redo, lab_t[3]);
#endif
}

// A8: emit the feature clouds in the reference's order (ring, sector, pick order): x,y,z,intensity,normal_x weight.
// counts: [0] sharp (own), [1] flat, [2] inten
__global__ void __launch_bounds__(128) k_fe_emit(const float4* __restrict__ C, int NS, const int* __restrict__ slots, const float* __restrict__ dsrc,
                                                 const float* __restrict__ osrc, float* __restrict__ sharp, float* __restrict__ flat,
                                                 float* __restrict__ inten, int cap, int* __restrict__ counts) {
  // one workgroup per (ring, sector) unit, one lane per pick (20 sharp + 40 flat + 20 intensity slots); the unit's output offsets =
  // the picks of the units before it, summed by the workgroup itself (<= 383 units x 3 counts, strided loads + one reduction)
  __shared__ int red[3][128 / WAVE];
  __shared__ int off[3];
  const int nu = NS * 6, u = blockIdx.x, t = threadIdx.x;
  int a = 0, b = 0, c = 0;
  for (int v = t; v < u; v += 128) { a += slots[(size_t)v * SLOT + 80]; b += slots[(size_t)v * SLOT + 81]; c += slots[(size_t)v * SLOT + 82]; }
  for (int o = WAVE / 2; o > 0; o >>= 1) { a += __shfl_down(a, o); b += __shfl_down(b, o); c += __shfl_down(c, o); }
  if ((t & (WAVE - 1)) == 0) { red[0][t / WAVE] = a; red[1][t / WAVE] = b; red[2][t / WAVE] = c; }
  __syncthreads();
  if (t < 3) { int v = 0; for (int w = 0; w < 128 / WAVE; w++) v += red[t][w]; off[t] = v; }
  __syncthreads();
  const int* sl = slots + (size_t)u * SLOT;
  const int ns = sl[80], nf = sl[81], ni = sl[82];
  if (u == nu - 1 && t == 0) { counts[0] = off[0] + ns; counts[1] = off[1] + nf; counts[2] = off[2] + ni; }
  if (t < 20) {
    if (t < ns) { const int ind = sl[t], d = off[0] + t;
      if (d < cap) { const float4 p = C[ind]; float* f = sharp + (size_t)d * 5; f[0] = p.x; f[1] = p.y; f[2] = p.z; f[3] = p.w; f[4] = dsrc[ind] + 1; } }  // :501
  } else if (t < 60) {
    const int k = t - 20;
    if (k < nf) { const int ind = sl[20 + k], d = off[1] + k;
      if (d < cap) { const float4 p = C[ind]; float* f = flat + (size_t)d * 5; f[0] = p.x; f[1] = p.y; f[2] = p.z; f[3] = p.w; f[4] = dsrc[ind]; } }     // :554
  } else if (t < 80) {
    const int k = t - 60;
    if (k < ni) { const int ind = sl[60 + k], d = off[2] + k;
      if (d < cap) { const float4 p = C[ind]; float* f = inten + (size_t)d * 5; f[0] = p.x; f[1] = p.y; f[2] = p.z; f[3] = p.w; f[4] = osrc[ind]; } }    // :609
  }
}

// ------------------------------------------------------------------------------------------------
static inline int nblk(long long n, int t) { return (int)((n + t - 1) / t); }

int fe_blocks(int n) { return nblk(n, FE_T); }
int fe_slot_ints() { return SLOT; }

void fe_filter(hipStream_t s, const float* in, int stride_f, int n, FeParams p, int* ring, int* st, int* rank_in_block, int* blk_hist) {
  hipLaunchKernelGGL(k_fe_filter, dim3(nblk(n, FE_T)), dim3(FE_T), 0, s, in, stride_f, n, p, ring, st, rank_in_block, blk_hist);
}
void fe_half(hipStream_t s, const float* in, int stride_f, int n, const int* ring, int* st) {
  hipLaunchKernelGGL(k_fe_half, dim3(nblk(n, FE_T)), dim3(FE_T), 0, s, in, stride_f, n, ring, st);
}
void fe_bucket(hipStream_t s, const float* in, int stride_f, int n, int NS, const int* ring, int* rank_in_block, int* blk_hist, int* meta,
               const int* st, float4* C, int* inum2, int* z0, int* z1, int* z2, int* z3) {
  const int nb = nblk(n, FE_T);
  hipLaunchKernelGGL(k_fe_hist_scan, dim3(1), dim3(64 * HS_CH), 0, s, nb, NS, blk_hist, meta);
  hipLaunchKernelGGL(k_fe_scatter, dim3(nb), dim3(FE_T), 0, s, in, stride_f, n, ring, rank_in_block, blk_hist, meta, st, C, inum2, z0, z1, z2, z3);
}
void fe_stencils(hipStream_t s, const float4* C, int cs, const int* csp, float* range_vec, float* scan_angle, const int* inum2, int* inum, float* curv,
                 float* curv2, float* icurv, float* dsrc, float* osrc, int* picked) {
  hipLaunchKernelGGL(k_fe_stencils, dim3(nblk(cs, FE_T)), dim3(FE_T), 0, s, C, cs, csp, inum2, range_vec, scan_angle, inum, curv, curv2, icurv, dsrc, osrc,
                     picked);
}
void fe_ground(hipStream_t s, const float4* C, int cs, const int* csp, int NS, const float* range_vec, const int* meta, int* gmark, int* mult, int* seedcnt,
               double* partials, double* out11, double* fit) {
  const int nb = nblk(cs, FE_T);
  hipLaunchKernelGGL(k_fe_ground, dim3(nb), dim3(FE_T), 0, s, C, cs, csp, NS, range_vec, meta, gmark, mult, seedcnt, partials);
  hipLaunchKernelGGL(k_fe_fold_fit, dim3(1), dim3(11 * WAVE), 0, s, partials, nb, out11, fit);
}
void fe_ground_dist(hipStream_t s, const float4* C, int cs, const int* csp, const int* mult, const double* fit, double* partials, double* out2) {
  const int nb = nblk(cs, FE_T);
  hipLaunchKernelGGL(k_fe_ground_dist, dim3(nb), dim3(FE_T), 0, s, C, cs, csp, mult, fit, partials);
  hipLaunchKernelGGL(k_fe_fold, dim3(2), dim3(WAVE), 0, s, partials, nb, 2, out2);
}
void fe_ground_list(hipStream_t s, const float4* C, int cs, const int* csp, int NS, const float* range_vec, const int* meta, const int* seedcnt,
                    const int* seedpos, float4* out, int cap) {
  hipLaunchKernelGGL(k_fe_ground_list, dim3(nblk(cs, FE_T)), dim3(FE_T), 0, s, C, cs, csp, NS, range_vec, meta, seedcnt, seedpos, out, cap);
}
void fe_select(hipStream_t s, const float4* C, int NS, const int* meta, const float* curv, const float* curv2, const float* icurv, const int* inum,
               const int* gmark, int* picked, int* ipicked, int* label, int* ilabel, int* slots, int* flags, int max_ring, int* sorted_curv,
               int* sorted_icurv) {
  hipLaunchKernelGGL(k_fe_sort, dim3(NS * 6, 2), dim3(SEL_T), 0, s, NS, meta, curv, icurv, sorted_curv, sorted_icurv);
  // LDS sized from the largest ring: the two sorted index lists of a sector + the per-sector window arrays
  // as many consecutive sectors per staging as fit (a 16- or 64-beam ring of ~2000 points: all six)
  int group = 6, sec_cap = 0;
  size_t lds = 0;
  for (;;) {
    sec_cap = (int)(((long long)max_ring * group + 5) / 6) + 2;
    if (sec_cap > SEC_MAX * group) sec_cap = SEC_MAX * group;
    lds = sizeof(int) * 2 * (size_t)sec_cap + (size_t)(sec_cap + 2 * SEL_MARGIN) * (7 * 4 + 11) + 64;
    if (lds <= 150 * 1024 || group == 1) break;
    group = group == 6 ? 3 : (group == 3 ? 2 : 1);
  }
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void*)k_fe_select, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
    attr_done = true;
  }
  hipLaunchKernelGGL(k_fe_select, dim3(NS), dim3(SELP_T), lds, s, C, NS, meta, curv, curv2, icurv, inum, gmark, picked, ipicked, label, ilabel, slots, flags,
                     sec_cap, sorted_curv, sorted_icurv, group);
}
void fe_emit(hipStream_t s, const float4* C, int NS, const int* slots, const float* dsrc, const float* osrc, float* sharp, float* flat, float* inten,
             int cap, int* counts) {
  hipLaunchKernelGGL(k_fe_emit, dim3(NS * 6), dim3(128), 0, s, C, NS, slots, dsrc, osrc, sharp, flat, inten, cap, counts);
}

}  // namespace rgck
