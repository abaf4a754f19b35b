// rgc_pre.hip -- gfx950 kernels for the stages either side of the registration operator in the odometer's per-frame
// body: de-skew (B2), pcl::VoxelGrid leaf-centroid down-sampling (B3) and the fp64 rigid transform used to re-express
// the sub-map in the new body frame (B9).  Reference citations are relative to /root/reference/rgc_slam/.
#include <limits.h>

#include "rgc_kernels.h"

namespace rgck {

constexpr int WAVE = 64;

// ------------------------------------------------------------------------------------------------
// B2  vg_ICP::adjustDistortion (src/RGC_odometer.cpp:1441-1481): every point is moved to the END of the sweep,
//   s = 1 - frac(intensity) / SCAN_PERIOD               (fp32, as the reference's expression types make it)
//   q_s = Identity.slerp(s, q_last_curr^-1),  p' = q_s * (p - s * t_last_curr)     (fp64, stored as fp32)
// Eigen's slerp and quaternion-vector product are restated below [3P-memory].
// ------------------------------------------------------------------------------------------------
__global__ void k_deskew(float* __restrict__ xyzi, int stride_f, int n, Quat qinv, double tx, double ty, double tz) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float* p = xyzi + (size_t)i * stride_f;
  const float inten = p[3];
  const float sf = 1 - (inten - (float)(int)inten) / 0.1f;  // SCAN_PERIOD = 0.1f, RGC_odometer.cpp:323
  const double s = (double)sf;
  // Quaterniond::Identity().slerp(s, qinv)
  const double d = qinv.w;  // dot(identity, qinv)
  const double absD = fabs(d);
  double scale0, scale1;
  if (absD >= 1.0 - 2.220446049250313e-16) {
    scale0 = 1.0 - s;
    scale1 = s;
  } else {
    const double theta = acos(absD), sinTheta = sin(theta);
    scale0 = sin((1.0 - s) * theta) / sinTheta;
    scale1 = sin(s * theta) / sinTheta;
  }
  if (d < 0) scale1 = -scale1;
  const double qx = scale1 * qinv.x, qy = scale1 * qinv.y, qz = scale1 * qinv.z, qw = scale0 + scale1 * qinv.w;
  const double vx = (double)p[0] - s * tx, vy = (double)p[1] - s * ty, vz = (double)p[2] - s * tz;
  // Eigen: uv = 2 * (q.vec x v);  v + w * uv + q.vec x uv
  double ux = qy * vz - qz * vy, uy = qz * vx - qx * vz, uz = qx * vy - qy * vx;
  ux += ux; uy += uy; uz += uz;
  p[0] = (float)(vx + qw * ux + (qy * uz - qz * uy));
  p[1] = (float)(vy + qw * uy + (qz * ux - qx * uz));
  p[2] = (float)(vz + qw * uz + (qx * uy - qy * ux));
}

// B9  vg_ICP::transformPointCloud (src/RGC_odometer.cpp:1495-1514): point_w = q * p + t in fp64, stored fp32,
// intensity copied.
__global__ void k_transform_q(const float* __restrict__ in, int stride_f, int n, Quat q, double tx, double ty, double tz,
                              float* __restrict__ out, int ostride_f) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float* p = in + (size_t)i * stride_f;
  const double vx = (double)p[0], vy = (double)p[1], vz = (double)p[2];
  double ux = q.y * vz - q.z * vy, uy = q.z * vx - q.x * vz, uz = q.x * vy - q.y * vx;
  ux += ux; uy += uy; uz += uz;
  float* o = out + (size_t)i * ostride_f;
  o[0] = (float)(vx + q.w * ux + (q.y * uz - q.z * uy) + tx);
  o[1] = (float)(vy + q.w * uy + (q.z * ux - q.x * uz) + ty);
  o[2] = (float)(vz + q.w * uz + (q.x * uy - q.y * ux) + tz);
  if (ostride_f > 3) o[3] = stride_f > 3 ? p[3] : 0.f;
}

// ------------------------------------------------------------------------------------------------
// B3  pcl::VoxelGrid<PointXYZI>::filter (src/RGC_odometer.cpp:976-991; SURVEY A.6 [3P-memory]):
//   ijk = floor(p * inv_leaf) - min_b ; idx = i + j*dx + k*dx*dy ; one output per occupied leaf = mean of ALL fields
//   (fp32 running sum / count) ; output ordered by idx.  Inside a leaf the sum runs in ascending point index (PCL's
//   std::sort leaves that order unspecified).  Counting sort over the rows of the leaf grid (over the leaves themselves for a dense
//   cloud) + compaction by leaf heads; see the chain below.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int leaf_coord(float x, float inv) { return (int)floorf(x * inv); }

__global__ void __launch_bounds__(256) k_vg_bbox(const float* __restrict__ in, int stride_f, int n, float inv, int* mm6, int* flags) {
  __shared__ int red[256 / WAVE][6];
  int lo[3] = {INT_MAX, INT_MAX, INT_MAX}, hi[3] = {INT_MIN, INT_MIN, INT_MIN};
  int bad = 0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const float* p = in + (size_t)i * stride_f;
#pragma unroll
    for (int a = 0; a < 3; a++) {
      const float v = p[a];
      if (!isfinite(v) || fabsf(v * inv) > 1.0e9f) { bad = 1; continue; }
      const int c = leaf_coord(v, inv);
      lo[a] = min(lo[a], c);
      hi[a] = max(hi[a], c);
    }
  }
  const int lane = threadIdx.x & (WAVE - 1), w = threadIdx.x / WAVE;
#pragma unroll
  for (int a = 0; a < 3; a++) {
    int l = lo[a], h = hi[a];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { l = min(l, __shfl_xor(l, o)); h = max(h, __shfl_xor(h, o)); }
    if (lane == 0) { red[w][a] = l; red[w][3 + a] = h; }
  }
  __syncthreads();
  if (threadIdx.x < 6) {
    int v = red[0][threadIdx.x];
    for (int j = 1; j < 256 / WAVE; j++) v = threadIdx.x < 3 ? min(v, red[j][threadIdx.x]) : max(v, red[j][threadIdx.x]);
    if (threadIdx.x < 3) { if (v != INT_MAX) atomicMin(&mm6[threadIdx.x], v); }
    else { if (v != INT_MIN) atomicMax(&mm6[threadIdx.x], v); }
  }
  if (bad) atomicOr(flags, 1);
}

// ---- the filter as ONE chain of launches.  A 30 k-point sweep at 0.2 m leaves spans ten million leaves (zero-filling and scanning a
// dense leaf array was most of the filter's time):  Leaves are ordered by idx = i + j dx + k dx dy, i.e. by (k, j) row first and by i
// inside the row: the counting sort runs over the ROWS (dy x dz entries, tens of thousands), and inside a row the points are ranked by
// (i, point index) -- rows hold few points when the grid is sparse.  Same output, bit for bit.
//
// The order of the leaves -- (k, j, i) lexicographic -- does not depend on WHICH box the grid spans, only that it holds every point: the
// filter may therefore run on a box kept from the previous cloud of the same leaf size (padded), without asking the host for this
// cloud's bounding box first.  `edge` > 0 marks such a speculative box: a point outside it sets flag bit 1 (the host repeats the filter
// with the exact box), a point within `edge` leaves of its faces sets bit 2 (the next call measures the box again); a non-finite point
// sets bit 0 in either mode.  Seven launches, one read-back: count, scan of the rows (two launches, the counters are left at zero for
// the next cloud), placement, rank, leaf heads + their block scan, centroids.
constexpr int VG_SCAN_T = 256, VG_SCAN_V = 8, VG_SCAN_B = VG_SCAN_T * VG_SCAN_V;

__global__ void __launch_bounds__(256)
k_vg_rows_count(const float* __restrict__ in, int stride_f, int n, float inv, LeafGrid g, int edge, int seg_shift, int nseg, int* __restrict__ row_of,
                int* __restrict__ lx, int* cnt, int* __restrict__ slot, int* flags) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & (WAVE - 1);
  const bool valid = i < n;
  int r = -1 - lane;  // lanes past the end: distinct negative keys, they never extend a neighbour's run
  bool near = false;
  if (valid) {
    const float* p = in + (size_t)i * stride_f;
    const float x = p[0], y = p[1], z = p[2];
    const bool fin = fabsf(x * inv) <= 1.0e9f && fabsf(y * inv) <= 1.0e9f && fabsf(z * inv) <= 1.0e9f;  // false for NaN / inf too
    int cx = fin ? leaf_coord(x, inv) - g.minb[0] : 0, cy = fin ? leaf_coord(y, inv) - g.minb[1] : 0, cz = fin ? leaf_coord(z, inv) - g.minb[2] : 0;
    const bool inside = cx >= 0 && cx < g.div[0] && cy >= 0 && cy < g.div[1] && cz >= 0 && cz < g.div[2];
    if (!fin || !inside) {  // parked in leaf 0 so that nothing is written out of bounds; the host discards the result
      atomicOr(flags, fin ? 2 : 1);
      cx = cy = cz = 0;
    } else {
      near = cx < edge || cx >= g.div[0] - edge || cy < edge || cy >= g.div[1] - edge || cz < edge || cz >= g.div[2] - edge;
    }
    // The counting sort's bucket is a SEGMENT of 2^seg_shift leaves of a grid row: the whole row for a sweep (seg_shift = 31: tens of
    // thousands of buckets, a few hundred points in the fullest), the leaf itself for a dense cloud (seg_shift = 0), and in between for a
    // large cloud in a box too big for leaf buckets (a keyframe store: the ranking pass below is quadratic in a bucket's population, and
    // a ground-level row of such a store holds thousands of points -- 140 us of a 0.19 ms filter).  Buckets in (k, j, segment) order and
    // points by (leaf x, index) inside one give the order of idx = i + j dx + k dx dy whatever the segment length.
    r = (cy + cz * g.div[1]) * nseg + (cx >> seg_shift);
    row_of[i] = r;
    lx[i] = seg_shift >= 13 ? cx : (cx & ((1 << seg_shift) - 1));
  }
  const unsigned long long nm = __ballot(near);
  if (nm && lane == __ffsll((long long)nm) - 1) atomicOr(flags, 4);
  // consecutive points of a sweep fall into the same row: one atomicAdd per RUN of equal rows inside the wave, not one per point
  // (same-address atomics cost ~12 ns each on this part); the returned count is the run's first arrival-order slot inside its row.
  const int prev = __shfl_up(r, 1);
  const bool head = lane == 0 || r != prev;
  const unsigned long long hm = __ballot(head);
  const int head_lane = 63 - __clzll(hm & ((2ull << lane) - 1ull));
  const unsigned long long above = head_lane == 63 ? 0ull : (hm >> (head_lane + 1));
  const int run_len = above ? __ffsll((long long)above) : WAVE - head_lane;
  int base = 0;
  if (head && valid) base = atomicAdd(&cnt[r], run_len);
  base = __shfl(base, head_lane);
  if (valid) slot[i] = base + (lane - head_lane);
}
// arrival-order placement: tmp[row start + slot] = one 64-bit record per point.  kVgPacked (leaf x < 2^13: the host checks the grid):
//   {leaf x : 13 | point index : 27 | arrival slot in the row : 12 | row population : 12}  -- its top 40 bits are the key the rank compares,
//   and the ranking pass finds the row's extent in the record itself (s - slot, + population) instead of looking the point's row and the
//   row's bounds up: two dependent random reads per point into tables of millions of entries, which is what that pass cost (60 us of a
//   1.3 M-point keyframe store's filter).  A row of more than 4095 points leaves the population field 0: looked up as before.
// otherwise {leaf x : 32 | point index : 32}.
constexpr int kVgIdxBits = 27, kVgCntBits = 12;
constexpr unsigned long long kVgCntMask = (1ull << kVgCntBits) - 1ull;
template <bool kVgPacked>
__global__ void k_vg_rows_place(int n, const int* __restrict__ row_of, const int* __restrict__ lx, const int* __restrict__ slot,
                                const int* __restrict__ start, unsigned long long* __restrict__ tmp) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int r = row_of[i], s0 = start[r], sl = slot[i];
  if (kVgPacked) {
    const int cnt = start[r + 1] - s0;
    const unsigned long long tail = cnt <= (int)kVgCntMask ? (((unsigned long long)(unsigned)sl << kVgCntBits) | (unsigned)cnt) : 0ull;
    tmp[s0 + sl] = ((unsigned long long)(unsigned)lx[i] << (kVgIdxBits + 2 * kVgCntBits)) | ((unsigned long long)(unsigned)i << (2 * kVgCntBits)) | tail;
  } else {
    tmp[s0 + sl] = ((unsigned long long)(unsigned)lx[i] << 32) | (unsigned)i;
  }
}
// final slot of a point = row start + number of same-row points that precede it in (leaf x, point index) order;
// order[slot] = point index, leaf[slot] = (row start, leaf x) packed: equal values = same leaf
template <bool kVgPacked>
__global__ void k_vg_rows_rank(int n, const int* __restrict__ row_of, const int* __restrict__ start, const unsigned long long* __restrict__ tmp,
                               int* __restrict__ order, unsigned long long* __restrict__ leaf) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n) return;
  const unsigned long long me = tmp[s];
  constexpr int kKeyShift = kVgPacked ? 2 * kVgCntBits : 0;
  const int i = kVgPacked ? (int)((me >> kKeyShift) & ((1ull << kVgIdxBits) - 1ull)) : (int)(unsigned)me;
  const unsigned lxv = kVgPacked ? (unsigned)(me >> (kVgIdxBits + kKeyShift)) : (unsigned)(me >> 32);
  int s0, s1;
  if (kVgPacked && (me & kVgCntMask) != 0) {
    s0 = s - (int)((me >> kVgCntBits) & kVgCntMask);
    s1 = s0 + (int)(me & kVgCntMask);
  } else {
    const int r = row_of[i];
    s0 = start[r];
    s1 = start[r + 1];
  }
  const unsigned long long key = me >> kKeyShift;
  int rank = 0, t = s0;
  for (; t + 8 <= s1; t += 8) {  // eight independent loads in flight: a ring of the sweep inside one row is hundreds of members
    unsigned long long o[8];
#pragma unroll
    for (int u = 0; u < 8; u++) o[u] = tmp[t + u];
#pragma unroll
    for (int u = 0; u < 8; u++) rank += ((o[u] >> kKeyShift) < key);
  }
  for (; t < s1; t++) rank += ((tmp[t] >> kKeyShift) < key);
  order[s0 + rank] = i;
  leaf[s0 + rank] = ((unsigned long long)(unsigned)s0 << 32) | lxv;
}
// pos[s] = number of leaf heads before s inside this block of 2048 sorted slots; block_sums[b] = heads in block b
__global__ void __launch_bounds__(VG_SCAN_T)
k_vg_rows_heads(const unsigned long long* __restrict__ leaf, int n, int* __restrict__ pos, int* __restrict__ block_sums) {
  __shared__ int wsum[VG_SCAN_T / WAVE];
  const int base = blockIdx.x * VG_SCAN_B + threadIdx.x * VG_SCAN_V;
  unsigned long long prev = (base > 0 && base <= n) ? leaf[base - 1] : ~0ull;
  int v[VG_SCAN_V], sum = 0;
#pragma unroll
  for (int j = 0; j < VG_SCAN_V; j++) {
    v[j] = 0;
    if (base + j < n) {
      const unsigned long long cur = leaf[base + j];
      v[j] = (base + j == 0 || cur != prev) ? 1 : 0;
      prev = cur;
    }
    sum += v[j];
  }
  const int lane = threadIdx.x & (WAVE - 1), w = threadIdx.x / WAVE;
  int inc = sum;
#pragma unroll
  for (int o = 1; o < WAVE; o <<= 1) {
    const int t = __shfl_up(inc, o);
    if (lane >= o) inc += t;
  }
  if (lane == WAVE - 1) wsum[w] = inc;
  __syncthreads();
  int before = 0, tot = 0;
#pragma unroll
  for (int j = 0; j < VG_SCAN_T / WAVE; j++) {
    if (j < w) before += wsum[j];
    tot += wsum[j];
  }
  int ex = before + inc - sum;
#pragma unroll
  for (int j = 0; j < VG_SCAN_V; j++) {
    if (base + j < n) pos[base + j] = ex;
    ex += v[j];
  }
  if (threadIdx.x == 0) block_sums[blockIdx.x] = tot;
}
// one lane per sorted slot; the head of a leaf sums its points in ascending point index (fp32, like the dense path) and writes output
// number pos + (heads in the scan blocks before this one), which every workgroup adds up for itself (<= 4096 values).
// res[0] = the flags of this run (and the live word res[1] goes back to zero for the next run), res[2] = number of leaves.
__global__ void __launch_bounds__(256)
k_vg_rows_centroid(const float* __restrict__ in, int stride_f, int n, const int* __restrict__ order, const unsigned long long* __restrict__ leaf,
                   const int* __restrict__ pos, const int* __restrict__ block_sums, float* __restrict__ out, int* res) {
  __shared__ int part[256 / WAVE];
  const int sb = (int)((blockIdx.x * 256u) / VG_SCAN_B);
  int acc = 0;
  for (int j = threadIdx.x; j < sb; j += 256) acc += block_sums[j];
#pragma unroll
  for (int o = WAVE / 2; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  if ((threadIdx.x & (WAVE - 1)) == 0) part[threadIdx.x / WAVE] = acc;
  __syncthreads();
  int pre = 0;
#pragma unroll
  for (int w = 0; w < 256 / WAVE; w++) pre += part[w];
  const int s = blockIdx.x * 256 + threadIdx.x;
  if (s >= n) return;
  const unsigned long long me = leaf[s];
  const bool head = s == 0 || leaf[s - 1] != me;
  const int op = pos[s] + pre;
  if (s == n - 1) {
    res[2] = op + (head ? 1 : 0);
    res[0] = res[1];
    res[1] = 0;
  }
  if (!head) return;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  int t = s;
  do {
    const float* p = in + (size_t)order[t] * stride_f;
    a0 += p[0]; a1 += p[1]; a2 += p[2];
    a3 += stride_f > 3 ? p[3] : 0.f;
    t++;
  } while (t < n && leaf[t] == me);
  const float cnt = (float)(t - s);
  float* o = out + (size_t)op * 4;
  o[0] = a0 / cnt; o[1] = a1 / cnt; o[2] = a2 / cnt; o[3] = a3 / cnt;
}

static inline int nblk(long long n, int t) { return (int)((n + t - 1) / t); }

void deskew(hipStream_t s, float* xyzi, int stride_f, int n, Quat qinv, const double t[3]) {
  hipLaunchKernelGGL(k_deskew, dim3(nblk(n, 256)), dim3(256), 0, s, xyzi, stride_f, n, qinv, t[0], t[1], t[2]);
}
void transform_q(hipStream_t s, const float* in, int stride_f, int n, Quat q, const double t[3], float* out, int ostride_f) {
  hipLaunchKernelGGL(k_transform_q, dim3(nblk(n, 256)), dim3(256), 0, s, in, stride_f, n, q, t[0], t[1], t[2], out, ostride_f);
}
void vg_bbox(hipStream_t s, const float* in, int stride_f, int n, float inv, int* mm6, int* flags) {
  hipLaunchKernelGGL(k_vg_bbox, dim3(min(nblk(n, 256), 1024)), dim3(256), 0, s, in, stride_f, n, inv, mm6, flags);
}
int vg_segments(const LeafGrid& g, int seg_shift) { return seg_shift >= 31 ? 1 : (int)(((long long)g.div[0] + (1ll << seg_shift) - 1) >> seg_shift); }
void vg_rows(hipStream_t s, const float* in, int stride_f, int n, float inv, LeafGrid g, int edge, int seg_shift, int* row_of, int* lx, int* slot_then_pos, int* cnt,
             int* start, void* row_block_sums, unsigned long long* tmp, int* order, unsigned long long* leaf, int* head_block_sums, float* out,
             int* res) {
  const int nseg = vg_segments(g, seg_shift);
  const int nr1 = g.div[1] * g.div[2] * nseg + 1;
  hipLaunchKernelGGL(k_vg_rows_count, dim3(nblk(n, 256)), dim3(256), 0, s, in, stride_f, n, inv, g, edge, seg_shift, nseg, row_of, lx, cnt, slot_then_pos, res + 1);
  scan_cells(s, cnt, start, nr1, row_block_sums, nullptr, nullptr, 0, nullptr);
  if ((seg_shift <= 13 || g.div[0] <= (1 << 13)) && n <= (1 << kVgIdxBits)) {  // (the packed record holds 13 bits of leaf x inside the bucket and 27 of point index)
    hipLaunchKernelGGL(k_vg_rows_place<true>, dim3(nblk(n, 256)), dim3(256), 0, s, n, row_of, lx, slot_then_pos, start, tmp);
    hipLaunchKernelGGL(k_vg_rows_rank<true>, dim3(nblk(n, 256)), dim3(256), 0, s, n, row_of, start, tmp, order, leaf);
  } else {
    hipLaunchKernelGGL(k_vg_rows_place<false>, dim3(nblk(n, 256)), dim3(256), 0, s, n, row_of, lx, slot_then_pos, start, tmp);
    hipLaunchKernelGGL(k_vg_rows_rank<false>, dim3(nblk(n, 256)), dim3(256), 0, s, n, row_of, start, tmp, order, leaf);
  }
  hipLaunchKernelGGL(k_vg_rows_heads, dim3(nblk(n, VG_SCAN_B)), dim3(VG_SCAN_T), 0, s, leaf, n, slot_then_pos, head_block_sums);
  hipLaunchKernelGGL(k_vg_rows_centroid, dim3(nblk(n, 256)), dim3(256), 0, s, in, stride_f, n, order, leaf, slot_then_pos, head_block_sums, out, res);
}


// ------------------------------------------------------------------------------------------------
// f3  sensor_msgs/PointCloud2 <-> device arrays (pcl::fromROSMsg at src/scanRegistration.cpp:107-108, pcl::toROSMsg at
// :689-727 and src/RGC_odometer.cpp:1348).  One lane per point: the fields are gathered from the message's byte layout
// (offset + PointField datatype per field, either endianness) straight into the float4 {x, y, z, intensity} the front-end
// consumes, plus the Velodyne driver's `ring` / `time` fields when present -- no PCL, no host-side repacking.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double pc2_load(const unsigned char* p, int type, int big) {
  unsigned char b[8];
  const int size = (type == 1 || type == 2) ? 1 : (type == 3 || type == 4) ? 2 : (type == 8 ? 8 : 4);
  for (int i = 0; i < size; i++) b[i] = big ? p[size - 1 - i] : p[i];
  switch (type) {  // sensor_msgs/PointField: INT8=1 UINT8=2 INT16=3 UINT16=4 INT32=5 UINT32=6 FLOAT32=7 FLOAT64=8
    case 1: return (double)(signed char)b[0];
    case 2: return (double)b[0];
    case 3: return (double)(short)(b[0] | (b[1] << 8));
    case 4: return (double)(unsigned short)(b[0] | (b[1] << 8));
    case 5: return (double)(int)(b[0] | (b[1] << 8) | (b[2] << 16) | ((unsigned)b[3] << 24));
    case 6: return (double)(unsigned)(b[0] | (b[1] << 8) | (b[2] << 16) | ((unsigned)b[3] << 24));
    case 7: return (double)__uint_as_float(b[0] | (b[1] << 8) | (b[2] << 16) | ((unsigned)b[3] << 24));
    case 8: {
      unsigned long long u = 0;
      for (int i = 0; i < 8; i++) u |= (unsigned long long)b[i] << (8 * i);
      return __longlong_as_double((long long)u);
    }
  }
  return 0.0;
}

__global__ void k_pc2_unpack(const unsigned char* __restrict__ data, int n, Pc2Layout L, float4* __restrict__ xyzi, int* __restrict__ ring,
                             float* __restrict__ time) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const unsigned char* p = data + (size_t)i * L.point_step;
  float v[4];
#pragma unroll
  for (int f = 0; f < 4; f++) v[f] = L.off[f] >= 0 ? (float)pc2_load(p + L.off[f], L.type[f], L.big_endian) : 0.0f;
  xyzi[i] = make_float4(v[0], v[1], v[2], v[3]);
  if (ring) ring[i] = L.off[4] >= 0 ? (int)pc2_load(p + L.off[4], L.type[4], L.big_endian) : -1;
  if (time) time[i] = L.off[5] >= 0 ? (float)pc2_load(p + L.off[5], L.type[5], L.big_endian) : 0.0f;
}

// pcl::toROSMsg of a PointXYZI (kind 0: 32-byte points, x y z @0 4 8, data[3] = 1.0f @12, intensity @16) or PointXYZINormal
// cloud (kind 1: 48-byte points, normal_x y z @16 20 24, intensity @32, curvature @36); in = cols floats per point:
// x, y, z, intensity[, normal_x]
__global__ void k_pc2_pack(const float* __restrict__ in, int cols, int n, int kind, unsigned char* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float* q = in + (size_t)i * cols;
  const int step = kind == 0 ? 32 : 48;
  float* o = reinterpret_cast<float*>(out + (size_t)i * step);
  for (int w = 0; w < step / 4; w++) o[w] = 0.0f;
  o[0] = q[0]; o[1] = q[1]; o[2] = q[2]; o[3] = 1.0f;
  if (kind == 0) {
    o[4] = q[3];
  } else {
    o[4] = cols > 4 ? q[4] : 0.0f;  // normal_x carries the feature weight (scanRegistration.cpp:501,554,609)
    o[8] = q[3];                    // intensity
  }
}

void pc2_unpack(hipStream_t s, const unsigned char* data, int n, const Pc2Layout& L, float4* xyzi, int* ring, float* time) {
  if (n > 0) hipLaunchKernelGGL(k_pc2_unpack, dim3((n + 255) / 256), dim3(256), 0, s, data, n, L, xyzi, ring, time);
}
void pc2_pack(hipStream_t s, const float* in, int cols, int n, int kind, unsigned char* out) {
  if (n > 0) hipLaunchKernelGGL(k_pc2_pack, dim3((n + 255) / 256), dim3(256), 0, s, in, cols, n, kind, out);
}

}  // namespace rgck
