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
//   std::sort leaves that order unspecified).  Counting sort over the dense leaf grid + first-in-leaf compaction.
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

__global__ void k_vg_count(const float* __restrict__ in, int stride_f, int n, float inv, LeafGrid g, int* __restrict__ cell_of, int* cnt) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float* p = in + (size_t)i * stride_f;
  const int c = (leaf_coord(p[0], inv) - g.minb[0]) + (leaf_coord(p[1], inv) - g.minb[1]) * g.div[0] +
                (leaf_coord(p[2], inv) - g.minb[2]) * g.div[0] * g.div[1];
  cell_of[i] = c;
  atomicAdd(&cnt[c], 1);
}

// final slot of a point = leaf start + number of same-leaf points with a smaller index (deterministic);
// order[slot] = original index, first[slot] = 1 for the first point of each leaf
__global__ void k_vg_rank(int n, const int* __restrict__ cell_of, const int* __restrict__ start, const int* __restrict__ order_tmp,
                          int* __restrict__ order, int* __restrict__ first) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n) return;
  const int i = order_tmp[s];
  const int c = cell_of[i];
  const int s0 = start[c], s1 = start[c + 1];
  int rank = 0;
  for (int t = s0; t < s1; t++) rank += (order_tmp[t] < i);
  order[s0 + rank] = i;
  first[s0 + rank] = (rank == 0) ? 1 : 0;
}

__global__ void k_vg_centroid(const float* __restrict__ in, int stride_f, int n, const int* __restrict__ cell_of,
                              const int* __restrict__ start, const int* __restrict__ order, const int* __restrict__ first,
                              const int* __restrict__ outpos, float* __restrict__ out, int* n_out) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n) return;
  if (s == n - 1) *n_out = outpos[s] + first[s];
  if (!first[s]) return;
  const int c = cell_of[order[s]];
  const int s1 = start[c + 1];
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  for (int t = s; t < s1; t++) {
    const float* p = in + (size_t)order[t] * stride_f;
    a0 += p[0]; a1 += p[1]; a2 += p[2];
    a3 += stride_f > 3 ? p[3] : 0.f;
  }
  const float cnt = (float)(s1 - s);
  float* o = out + (size_t)outpos[s] * 4;
  o[0] = a0 / cnt; o[1] = a1 / cnt; o[2] = a2 / cnt; o[3] = a3 / cnt;
}

// ---- the same filter for SPARSE leaf grids (a 30 k-point sweep at 0.2 m leaves spans ten million leaves: zero-filling and scanning the
// dense leaf array was most of the filter's time).  Leaves are ordered by idx = i + j dx + k dx dy, i.e. by (k, j) row first and by i
// inside the row: the counting sort runs over the ROWS (dy x dz entries, tens of thousands), and inside a row the points are ranked by
// (i, point index) -- rows hold few points when the grid is sparse.  Same output, bit for bit.
__global__ void k_vg_count_rows(const float* __restrict__ in, int stride_f, int n, float inv, LeafGrid g, int* __restrict__ row_of,
                                int* __restrict__ lx, int* cnt, int* __restrict__ slot) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float* p = in + (size_t)i * stride_f;
  const int r = (leaf_coord(p[1], inv) - g.minb[1]) + (leaf_coord(p[2], inv) - g.minb[2]) * g.div[1];
  row_of[i] = r;
  lx[i] = leaf_coord(p[0], inv) - g.minb[0];
  slot[i] = atomicAdd(&cnt[r], 1);  // arrival order inside the row: the placement needs no second atomic
}
// final slot of a point = row start + number of same-row points that precede it in (leaf x, point index) order; key[slot] = its leaf x
__global__ void k_vg_rank_rows(int n, const int* __restrict__ row_of, const int* __restrict__ lx, const int* __restrict__ start,
                               const int* __restrict__ order_tmp, int* __restrict__ order, int* __restrict__ key) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n) return;
  const int i = order_tmp[s];
  const int r = row_of[i], x = lx[i];
  const int s0 = start[r], s1 = start[r + 1];
  int rank = 0;
  for (int t = s0; t < s1; t++) {
    const int j = order_tmp[t], xj = lx[j];
    rank += (xj < x || (xj == x && j < i)) ? 1 : 0;
  }
  order[s0 + rank] = i;
  key[s0 + rank] = x;
}
// first[s] = 1 where a new leaf begins in the sorted order (new row or new leaf x)
__global__ void k_vg_first_rows(int n, const int* __restrict__ row_of, const int* __restrict__ order, const int* __restrict__ key,
                                int* __restrict__ first) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n) return;
  first[s] = (s == 0 || key[s] != key[s - 1] || row_of[order[s]] != row_of[order[s - 1]]) ? 1 : 0;
}
__global__ void k_vg_centroid_rows(const float* __restrict__ in, int stride_f, int n, const int* __restrict__ order, const int* __restrict__ first,
                                   const int* __restrict__ outpos, float* __restrict__ out, int* n_out) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n) return;
  if (s == n - 1) *n_out = outpos[s] + first[s];
  if (!first[s]) return;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  int t = s;
  do {  // the leaf's points in ascending point index, like the dense path
    const float* p = in + (size_t)order[t] * stride_f;
    a0 += p[0]; a1 += p[1]; a2 += p[2];
    a3 += stride_f > 3 ? p[3] : 0.f;
    t++;
  } while (t < n && !first[t]);
  const float cnt = (float)(t - s);
  float* o = out + (size_t)outpos[s] * 4;
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
void vg_count(hipStream_t s, const float* in, int stride_f, int n, float inv, LeafGrid g, int* cell_of, int* cnt) {
  hipLaunchKernelGGL(k_vg_count, dim3(nblk(n, 256)), dim3(256), 0, s, in, stride_f, n, inv, g, cell_of, cnt);
}
void vg_rank(hipStream_t s, int n, const int* cell_of, const int* start, const int* order_tmp, int* order, int* first) {
  hipLaunchKernelGGL(k_vg_rank, dim3(nblk(n, 256)), dim3(256), 0, s, n, cell_of, start, order_tmp, order, first);
}
void vg_count_rows(hipStream_t s, const float* in, int stride_f, int n, float inv, LeafGrid g, int* row_of, int* lx, int* cnt, int* slot) {
  hipLaunchKernelGGL(k_vg_count_rows, dim3(nblk(n, 256)), dim3(256), 0, s, in, stride_f, n, inv, g, row_of, lx, cnt, slot);
}
void vg_rank_rows(hipStream_t s, int n, const int* row_of, const int* lx, const int* start, const int* order_tmp, int* order, int* key, int* first) {
  hipLaunchKernelGGL(k_vg_rank_rows, dim3(nblk(n, 256)), dim3(256), 0, s, n, row_of, lx, start, order_tmp, order, key);
  hipLaunchKernelGGL(k_vg_first_rows, dim3(nblk(n, 256)), dim3(256), 0, s, n, row_of, order, key, first);
}
void vg_centroid_rows(hipStream_t s, const float* in, int stride_f, int n, const int* order, const int* first, const int* outpos, float* out,
                      int* n_out) {
  hipLaunchKernelGGL(k_vg_centroid_rows, dim3(nblk(n, 256)), dim3(256), 0, s, in, stride_f, n, order, first, outpos, out, n_out);
}
void vg_centroid(hipStream_t s, const float* in, int stride_f, int n, const int* cell_of, const int* start, const int* order,
                 const int* first, const int* outpos, float* out, int* n_out) {
  hipLaunchKernelGGL(k_vg_centroid, dim3(nblk(n, 256)), dim3(256), 0, s, in, stride_f, n, cell_of, start, order, first, outpos, out, n_out);
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
