// rgc_kernels.hip -- gfx950 (MI355X / CDNA4) kernels of the RGC-SLAM scan-to-map registration path.
//
// Written for 64-wide wavefronts; compiled with -ffp-contract=off so that dist2() is the same ((dx*dx + dy*dy) + dz*dz) FLANN's
// L2_Simple<float> produces on the reference's x86-64 build: wherever the ORDER of two candidates decides a neighbour set it is
// decided by that expression (the bulk kNN kernels rank by FMA-contracted keys and fall back to it when two keys are too close
// to tell), so the exact-kNN neighbour sets match the CPU path (fast_gicp_impl.hpp:254).  Everything downstream of the raw fp32
// points is fp64, like the reference (points are cast to double at fast_gicp_impl.hpp:258, fast_vgicp_impl.hpp:84).
//
// Reference citations are relative to /root/reference/rgc_slam/.
//
// Contents, in file order (one translation unit on purpose: the device helpers -- wave reductions, the register top-k chain,
// the row walkers, the Jacobi solver, the block reduction -- are shared by nearly every kernel and stay inlinable):
//   grid build      k_bbox, k_count, k_cells_reduce + k_cells_scan_{write,sums}, k_place, k_rank_gather (+ k_scan_*, a plain 32-bit
//                   exclusive scan: the front-end's ground list; the leaf filter of rgc_pre.hip uses the cell scan)
//   C2 kNN + cov    Chain, sp_piece_table, knn_point_sp (map: one lane per query, one pass), knn_point_split (scan: four lanes per
//                   query), k_knn_sp (the bulk launch of either), TopK, coop_kth, k_knn_coop (deferred queries, one wave per query)
//   C3 voxel map    k_voxel_build
//   C4-C7 solve     linearize_point, error_point, block_reduce_store, block_fold_rows_pre, lm_step_decide, k_lm_step (default
//                   driver), k_linearize / k_error / k_fold / k_lm_try (public fine seam)
//   C8, f4          nn_search, k_fitness(_lm), k_icp_accumulate, k_transform_f32
//   f1              k_mapreg_associate, k_mapreg_terms, k_mapreg_fold
//   launch wrappers at the end (namespace rgck, declared in rgc_kernels.h)
#include "rgc_kernels.h"
#include "rgc_lm.h"
#include <hip/hip_ext.h>

#include <limits.h>
#include <stddef.h>

#include <type_traits>

namespace rgck {

constexpr int WAVE = 64;

__device__ __forceinline__ int voxel_coord1(float x, double res) { return (int)floor((double)x / res - 0.5); }
// Cell of grid g (absolute, before - minc) along one axis: floor(x / res - 0.5), without the fp64 division when the cell size is a
// power of two (the product is then the same double).
__device__ __forceinline__ int cell_coord(float x, const Grid& g) {
  const double u = g.inv_res != 0.0 ? (double)x * g.inv_res - 0.5 : (double)x / g.res - 0.5;
  return (int)floor(u);
}
// lower wall of cell c (relative) along axis a; the upper wall is + g.res
__device__ __forceinline__ double cell_wall(const Grid& g, int a, int c) { return ((double)(c + g.minc[a]) + 0.5) * g.res; }

__device__ __forceinline__ int wave_min(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ int wave_max(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
  return v;
}

__device__ __forceinline__ void wave_lds_fence() {  // LDS is in-order within a wave: only the compiler must not reorder
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// DPP quad permutes (registers only): lane l of every quad of lanes reads lane (CTRL >> 2 l) & 3 of its quad
template <int CTRL>
__device__ __forceinline__ int quad_perm_i(int v) { return __builtin_amdgcn_mov_dpp(v, CTRL, 0xf, 0xf, true); }
template <int CTRL>
__device__ __forceinline__ double quad_perm_f64(double v) {
  const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
constexpr int kQuadSwap1 = 0xB1, kQuadSwap2 = 0x4E;  // lanes 0<->1, 2<->3 / lanes 0,1<->2,3
__device__ __forceinline__ int quad_max_i(int v) {
  v = max(v, quad_perm_i<kQuadSwap1>(v));
  return max(v, quad_perm_i<kQuadSwap2>(v));
}
__device__ __forceinline__ int quad_or_i(int v) {
  v |= quad_perm_i<kQuadSwap1>(v);
  return v | quad_perm_i<kQuadSwap2>(v);
}
__device__ __forceinline__ double quad_sum_f64(double v) {  // (l0 + l1) + (l2 + l3) in every lane of the quad... up to the order of each pair
  v += quad_perm_f64<kQuadSwap1>(v);
  return v + quad_perm_f64<kQuadSwap2>(v);
}

// ------------------------------------------------------------------------------------------------
// grid build
// ------------------------------------------------------------------------------------------------
// The scan's (source) preprocessing runs on a second stream underneath the map's VALU-bound kNN launch; its waves share
// CUs with that kernel and would get ~1/8 of the issue slots.  Raising the wave priority lets the small latency-chained
// source kernels finish while the big kernel soaks up what is left.
// Three levels: the map's throughput kernels stay at 0, the scan's preparation runs at 2, the solve's steps and the score -- the
// frame's critical chain, which with two contexts share the chip with the NEXT scan's preparation -- at 3.
#ifndef RGC_SCAN_PRIO
#define RGC_SCAN_PRIO 2
#endif
__device__ __forceinline__ void wave_prio(int hi) {
  if (hi >= 2) __builtin_amdgcn_s_setprio(3);
  else if (hi) __builtin_amdgcn_s_setprio(RGC_SCAN_PRIO);
}

// Developer build (-DRGC_LAB_TURN): the turn-around of a dependent sequence -- from the moment a solve's deciding launch posts its final pose to
// the first wave of the next map's counting pass (k_count<true>) -- in ticks of the 100 MHz wall clock: [0] sum, [1] count, [2] max,
// [3] the post's stamp, [4] the launch start's stamp (deciding launch entry -> post), [5] sum of that.  Printed by rgc_destroy's caller via lab_turn().
#ifdef RGC_LAB_TURN
__device__ unsigned long long g_lab_turn[8];
void lab_turn(unsigned long long* out8) {
  (void)hipDeviceSynchronize();
  (void)hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_lab_turn), sizeof(g_lab_turn));
  unsigned long long z[8] = {0};
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_lab_turn), z, sizeof(z));
}
#endif
// Developer build (-DRGC_LAB): how often the wave-level loops of the map's bulk kNN kernel run -- with the kernel's ISA that gives the
// executed instruction mix (scripts/isa_mix.py).  One count per WAVE (its first active lane).  Compiled out of the product.
#ifdef RGC_LAB
__device__ unsigned long long g_lab_iter[8];  // 0 waves, 1 quads processed in the scan loop, 2 chain inserts (drain rounds), 3 Newton steps, 4 Jacobi fallbacks, 5 exact tie-breaks
#define LAB_COUNT(slot)                                                                                    \
  do {                                                                                                     \
    const unsigned long long m_ = __ballot(true);                                                          \
    if ((int)(threadIdx.x & 63) == __ffsll((long long)m_) - 1) atomicAdd(&g_lab_iter[slot], 1ull);         \
  } while (0)
void lab_iters(unsigned long long* out8) {
  (void)hipDeviceSynchronize();
  (void)hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_lab_iter), sizeof(g_lab_iter));
  unsigned long long z[8] = {0};
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_lab_iter), z, sizeof(z));
}
__device__ int g_lab_decl[16];  // developer build: why knn_point_seeded declined, per LANE (1 no seed, 2 crowded row, 3 fewer than k, 4 more than k + 1, 5 k keys undecided, 6 k + 2 may contend, 7 three contenders, 8 exact tie)
#define LAB_DECLINE(r) atomicAdd(&g_lab_decl[r], 1)
void lab_declines(int* out16) {
  (void)hipDeviceSynchronize();
  (void)hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_lab_decl), sizeof(g_lab_decl));
  int z[16] = {0};
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_lab_decl), z, sizeof(z));
}
#else
#define LAB_COUNT(slot) do {} while (0)
#ifdef RGC_LAB_BLK
__device__ int g_lab_blk_why[16384];  // developer build: OR of (1 << reason) over the lanes of a workgroup whose seeded search declined
#define LAB_DECLINE(r) atomicOr(&g_lab_blk_why[blockIdx.x < 16384 ? blockIdx.x : 0], 1 << (r))
#else
#define LAB_DECLINE(r) do {} while (0)
#endif
#endif

// Bounding box in CELL coordinates.  voxel_coord1 is monotone, so the per-thread work is a float min/max (the fp64
// division runs once per thread, not once per coordinate); four points are in flight per lane; a block only issues
// its six same-address atomics when it would still move the global bound (a relaxed read first) -- with ~1000 blocks the
// serialised atomics used to cost more than the 16 MB read.
__global__ void __launch_bounds__(256) k_bbox(const float* __restrict__ in, int stride_f, int n, double res, int* mm6, int* flags, int prio) {
  wave_prio(prio);
  __shared__ int red[256 / WAVE][6];
  float flo[3] = {INFINITY, INFINITY, INFINITY}, fhi[3] = {-INFINITY, -INFINITY, -INFINITY};
  int bad = 0;
  const int T = gridDim.x * blockDim.x;
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  auto take = [&](float v, int a) {
    bad |= !(fabsf(v) <= 1.0e8f);  // NaN, inf and absurd coordinates
    flo[a] = fminf(flo[a], v);
    fhi[a] = fmaxf(fhi[a], v);
  };
  for (; i + 3 * T < n; i += 4 * T) {
    float v[4][3];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const float* p = in + (size_t)(i + u * T) * stride_f;
      v[u][0] = p[0]; v[u][1] = p[1]; v[u][2] = p[2];
    }
#pragma unroll
    for (int u = 0; u < 4; u++) { take(v[u][0], 0); take(v[u][1], 1); take(v[u][2], 2); }
  }
  for (; i < n; i += T) {
    const float* p = in + (size_t)i * stride_f;
    take(p[0], 0); take(p[1], 1); take(p[2], 2);
  }
  int lo[3], hi[3];
#pragma unroll
  for (int a = 0; a < 3; a++) {
    const bool any_pt = flo[a] <= fhi[a] && !bad;  // a flagged cloud is rejected by the host: its bounds do not matter
    lo[a] = any_pt ? voxel_coord1(flo[a], res) : INT_MAX;
    hi[a] = any_pt ? voxel_coord1(fhi[a], res) : INT_MIN;
  }
  const int lane = threadIdx.x & (WAVE - 1), w = threadIdx.x / WAVE;
#pragma unroll
  for (int a = 0; a < 3; a++) {
    const int l = wave_min(lo[a]), h = wave_max(hi[a]);
    if (lane == 0) { red[w][a] = l; red[w][3 + a] = h; }
  }
  __syncthreads();
  if (threadIdx.x < 6) {
    int v = red[0][threadIdx.x];
    for (int j = 1; j < 256 / WAVE; j++) v = threadIdx.x < 3 ? min(v, red[j][threadIdx.x]) : max(v, red[j][threadIdx.x]);
    const int cur = __hip_atomic_load(&mm6[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (threadIdx.x < 3) { if (v < cur) atomicMin(&mm6[threadIdx.x], v); }
    else { if (v > cur) atomicMax(&mm6[threadIdx.x], v); }
  }
  if (bad) atomicOr(flags, 1);
}

// Cell order: x fastest (a grid row = the cells of one (y, z) = one contiguous range of the sorted array), then y, then z.
// -DRGC_Y_SLOWEST=1 makes it x, z, y: a lidar map is flat (hundreds of cells in x and y, ten or twenty in z), so with y slowest a
// contiguous run of queries keeps its candidates in a ~0.5 MB window of the array, and with one long run per XCD (-DRGC_XCD_RUN=496)
// the bulk kNN launch re-reads almost nothing: 43 MB of HBM traffic per 1 M-query launch instead of 73 (algorithmic 36) -- but the
// launch is 3 % SLOWER (0.163-0.169 ms against 0.157: the populated neighbour rows of a surface are the y-neighbours, which that
// order moves apart) and the frame 2-3 % (scripts/exp_xcd_run.sh).  The kernel is bound by instruction issue, not bytes: the default
// keeps the faster order.
#ifndef RGC_Y_SLOWEST
#define RGC_Y_SLOWEST 0
#endif
__device__ __forceinline__ int cell_index(const Grid& g, int cx, int cy, int cz) {
  return RGC_Y_SLOWEST ? (cy * g.dim[2] + cz) * g.dim[0] + cx : (cz * g.dim[1] + cy) * g.dim[0] + cx;
}
// which (dy, dz) the r-th row of a D x D block is, rows counted in MEMORY order
__device__ __forceinline__ constexpr int row_dy(int r, int D, int R) { return (RGC_Y_SLOWEST ? r / D : r % D) - R; }
__device__ __forceinline__ constexpr int row_dz(int r, int D, int R) { return (RGC_Y_SLOWEST ? r % D : r / D) - R; }

// guard (nullable): the grid was NOT derived from this cloud (a speculative grid kept from the previous one): a point with
// non-finite / absurd coordinates sets bit 0, a point outside the grid bit 1 -- the host re-prepares the cloud when it
// learns of it; such a point is parked in cell 0 so that nothing is written out of bounds meanwhile.
// q * p + t, the arithmetic of k_transform_q (rgc_pre.hip; vg_ICP::transformPointCloud, src/RGC_odometer.cpp:1495-1514)
__device__ __forceinline__ float4 reframe_point(const Reframe& rf, int i) {
  const float* p = rf.src + (size_t)i * rf.src_stride_f;
  const Quat& q = rf.q;
  const double vx = (double)p[0], vy = (double)p[1], vz = (double)p[2];
  double ux = q.y * vz - q.z * vy, uy = q.z * vx - q.x * vz, uz = q.x * vy - q.y * vx;
  ux += ux; uy += uy; uz += uz;
  return make_float4((float)(vx + q.w * ux + (q.y * uz - q.z * uy) + rf.t[0]), (float)(vy + q.w * uy + (q.z * ux - q.x * uz) + rf.t[1]),
                     (float)(vz + q.w * uz + (q.x * uy - q.y * ux) + rf.t[2]), rf.src_stride_f > 3 ? p[3] : 0.f);
}

template <bool kReframe>
__global__ void k_count(const float* __restrict__ in, int stride_f, int n, Grid g, int* __restrict__ cell_of, int* __restrict__ slot_of,
                        int* cnt, int* guard, int prio, Reframe rf) {
  wave_prio(prio);
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const bool valid = i < n;
  const int lane = threadIdx.x & (WAVE - 1);
  int c = -1 - lane;
#ifdef RGC_LAB_TURN
  if (kReframe && i == 0) {
    const unsigned long long now = wall_clock64(), post = atomicExch(&g_lab_turn[3], 0ull);
    if (post) { atomicAdd(&g_lab_turn[0], now - post); atomicAdd(&g_lab_turn[1], 1ull); atomicMax(&g_lab_turn[2], now - post); }
  }
#endif  // lanes past the end: distinct negative keys, so they never extend a neighbour's run
  if (valid) {
    float pt[3];
    if (kReframe) {  // the cloud is produced here (stride 4 floats) and counted from registers
      if (rf.copy) {  // the neighbour-list cache's guard: is this still the map its lists were made for?  (x, y, z, bit for bit)
        const float* sp = rf.src + (size_t)i * rf.src_stride_f;
        const float4 old = rf.copy[i];
        if (__float_as_int(old.x) != __float_as_int(sp[0]) || __float_as_int(old.y) != __float_as_int(sp[1]) || __float_as_int(old.z) != __float_as_int(sp[2])) {
          rf.copy[i] = make_float4(sp[0], sp[1], sp[2], 0.f);
          *rf.epoch = rf.frame;  // (every thread that finds a difference stores the same value)
        }
        if (rf.force && i == 0) *rf.epoch = rf.frame;
      }
      const float4 w = reframe_point(rf, i);
      *reinterpret_cast<float4*>(const_cast<float*>(in) + (size_t)i * 4) = w;
      pt[0] = w.x; pt[1] = w.y; pt[2] = w.z;
    } else {
      const float* pp = in + (size_t)i * stride_f;
      pt[0] = pp[0]; pt[1] = pp[1]; pt[2] = pp[2];
    }
    const float* p = pt;
    int cx, cy, cz;
    if (guard) {
      const float x = p[0], y = p[1], z = p[2];
      const bool fin = fabsf(x) <= 1.0e8f && fabsf(y) <= 1.0e8f && fabsf(z) <= 1.0e8f;  // false for NaN too
      cx = fin ? cell_coord(x, g) - g.minc[0] : 0;
      cy = fin ? cell_coord(y, g) - g.minc[1] : 0;
      cz = fin ? cell_coord(z, g) - g.minc[2] : 0;
      const bool inside = cx >= 0 && cx < g.dim[0] && cy >= 0 && cy < g.dim[1] && cz >= 0 && cz < g.dim[2];
      if (!fin || !inside) { atomicOr(guard, fin ? 2 : 1); cx = cy = cz = 0; }
    } else {
      cx = cell_coord(p[0], g) - g.minc[0];
      cy = cell_coord(p[1], g) - g.minc[1];
      cz = cell_coord(p[2], g) - g.minc[2];
    }
    c = cell_index(g, cx, cy, cz);
    cell_of[i] = c;
  }
  // One atomicAdd per DISTINCT cell of the wave, not per point: the lanes that fall into the same cell are found with one ballot per
  // distinct cell (the first lane still unassigned names the cell, every lane compares), the first of them adds the group's size and the
  // returned count is the group's first (arrival-order) slot inside its cell -- the placement pass needs no second atomic.
  // Same-address atomics are what this pass costs (~12 ns each across the XCDs, serialised per address): a leaf-ordered map puts a wave
  // into ~6 cells; a raw sweep arrives in FIRING order -- consecutive points are the 16 lasers of one azimuth, 16 different cells, and
  // the next azimuth hits the same 16 again -- so grouping only CONSECUTIVE lanes (round 1-3) left it one atomic per point, hundreds of
  // them on the crowded cells next to the sensor: the 30 k-point scan's pass took as long as the 1 M-point map's (21 us) and slowed
  // every other kernel that was waiting for an atomic meanwhile (the solve's tickets).
  unsigned long long todo = __ballot(valid);
  int lead = lane, rank = 0, group = 1;
  while (todo) {
    const int l0 = __ffsll((long long)todo) - 1;
    const int cl = __shfl(c, l0);
    const unsigned long long same = __ballot(valid && c == cl);
    if (valid && c == cl) {
      lead = l0;
      rank = __popcll(same & ((1ull << lane) - 1ull));
      group = __popcll(same);
    }
    todo &= ~same;
  }
  int base = 0;
  if (valid && lane == lead) base = atomicAdd(&cnt[c], group);
  base = __shfl(base, lead);
  if (valid) slot_of[i] = base + rank;
}

// three-kernel exclusive scan: 2048 items per block (256 threads x 8)
constexpr int SCAN_T = 256, SCAN_V = 8, SCAN_B = SCAN_T * SCAN_V;

__device__ __forceinline__ int block_exclusive_scan(int v, int* total) {
  __shared__ int wsum[SCAN_T / WAVE];
  int lane = threadIdx.x & (WAVE - 1), w = threadIdx.x / WAVE;
  int inc = v;
#pragma unroll
  for (int o = 1; o < WAVE; o <<= 1) {
    int t = __shfl_up(inc, o);
    if (lane >= o) inc += t;
  }
  if (lane == WAVE - 1) wsum[w] = inc;
  __syncthreads();
  int base = 0, tot = 0;
#pragma unroll
  for (int j = 0; j < SCAN_T / WAVE; j++) {
    int s = wsum[j];
    if (j < w) base += s;
    tot += s;
  }
  __syncthreads();
  *total = tot;
  return base + inc - v;
}

__global__ void k_scan_block(const int* __restrict__ in, int* __restrict__ out, int n, int* __restrict__ block_sums, int prio) {
  wave_prio(prio);
  int base = blockIdx.x * SCAN_B + threadIdx.x * SCAN_V;
  int v[SCAN_V], s = 0;
#pragma unroll
  for (int j = 0; j < SCAN_V; j++) {
    v[j] = (base + j < n) ? in[base + j] : 0;
    s += v[j];
  }
  int tot;
  int ex = block_exclusive_scan(s, &tot);
#pragma unroll
  for (int j = 0; j < SCAN_V; j++) {
    if (base + j < n) out[base + j] = ex;
    ex += v[j];
  }
  if (threadIdx.x == 0) block_sums[blockIdx.x] = tot;
}

__global__ void k_scan_sums(int* sums, int nb, int prio) {
  wave_prio(prio);  // single block, in-place exclusive scan
  __shared__ int carry_s;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  for (int base = 0; base < nb; base += SCAN_T) {
    int i = base + threadIdx.x;
    int v = i < nb ? sums[i] : 0;
    int tot;
    int ex = block_exclusive_scan(v, &tot);
    int carry = carry_s;
    if (i < nb) sums[i] = ex + carry;
    __syncthreads();
    if (threadIdx.x == 0) carry_s = carry + tot;
    __syncthreads();
  }
}

__global__ void k_scan_add(int* out, int n, const int* __restrict__ block_sums, int prio) {
  wave_prio(prio);
  int base = blockIdx.x * SCAN_B + threadIdx.x * SCAN_V;
  int add = block_sums[blockIdx.x];
#pragma unroll
  for (int j = 0; j < SCAN_V; j++)
    if (base + j < n) out[base + j] += add;
}

// ---- cell scan: start[] AND the dense voxel numbering in one pass --------------------------------------------------
// value per cell = count | (occupied << 32): one 64-bit exclusive scan yields the cell's first sorted slot (low half) and
// the number of occupied cells before it (high half) = its voxel id.  Voxel ids therefore come out dense, deterministic and
// in cell order without a single atomic (one same-address atomic per wave used to bound the voxel kernel: ~12 ns each
// across the 8 XCDs).  n <= 2^27 points: the halves cannot carry into each other.
__device__ __forceinline__ unsigned long long block_exclusive_scan64(unsigned long long v, unsigned long long* total) {
  __shared__ unsigned long long wsum64[SCAN_T / WAVE];
  const int lane = threadIdx.x & (WAVE - 1), w = threadIdx.x / WAVE;
  unsigned long long inc = v;
#pragma unroll
  for (int o = 1; o < WAVE; o <<= 1) {
    const unsigned long long t = __shfl_up(inc, o);
    if (lane >= o) inc += t;
  }
  if (lane == WAVE - 1) wsum64[w] = inc;
  __syncthreads();
  unsigned long long base = 0, tot = 0;
#pragma unroll
  for (int j = 0; j < SCAN_T / WAVE; j++) {
    const unsigned long long s = wsum64[j];
    if (j < w) base += s;
    tot += s;
  }
  __syncthreads();
  *total = tot;
  return base + inc - v;
}

// A thread's SCAN_V = 8 consecutive entries as two 16-byte accesses where the whole group lies inside the array and the array is 16-byte
// aligned (hipMalloc's are): eight 4-byte accesses per thread at a stride of 32 bytes touch every cache line of the wave's 2 KB four
// times over and cost four times the instructions.  Past the end: zeros.
__device__ __forceinline__ void load_cells8(const int* __restrict__ a, int base, int n, bool wide, int (&v)[8]) {
  if (wide && base + 8 <= n) {
    const int4 lo = *reinterpret_cast<const int4*>(a + base), hi = *reinterpret_cast<const int4*>(a + base + 4);
    v[0] = lo.x; v[1] = lo.y; v[2] = lo.z; v[3] = lo.w; v[4] = hi.x; v[5] = hi.y; v[6] = hi.z; v[7] = hi.w;
  } else {
#pragma unroll
    for (int j = 0; j < 8; j++) v[j] = (base + j < n) ? a[base + j] : 0;
  }
}
__device__ __forceinline__ bool aligned16(const void* p) { return (reinterpret_cast<unsigned long long>(p) & 15ull) == 0ull; }

// cnt has n entries (cells + 1 sentinel of 0); cell_voxel (nullable) has n - 1.
__device__ __forceinline__ unsigned long long scan_value(int v) { return (unsigned long long)(unsigned)v | ((unsigned long long)(v > 0) << 32); }
// Reduce, then scan (round 3; the first pass used to write provisional start[] / cell_voxel[] that the second read, corrected and
// wrote again: 58 MB of traffic for a 2.4 M-cell grid, now 38):
//   pass 1 (k_cells_reduce): every workgroup's total of its SCAN_B entries -- reads only (and the scan grid's crowding figure);
//   pass 2 (k_cells_scan_write): the totals of the workgroups before this one (summed by the workgroup itself for <= 4096 of them,
//           else taken from k_cells_scan_sums), the entries once more, their exclusive scan, and start[] / cell_voxel[]
//           written ONCE with their final values; the counters are consumed here (left at zero for the next cloud).
// A grid of one workgroup needs pass 2 only.
__global__ void __launch_bounds__(SCAN_T) k_cells_reduce(const int* __restrict__ cnt, int n, unsigned long long* __restrict__ block_sums, int prio,
                                                         float* __restrict__ sum_sq) {
  wave_prio(prio);
  __shared__ unsigned long long part[SCAN_T / WAVE];
  static_assert(SCAN_V == 8, "load_cells8");
  const int base = blockIdx.x * SCAN_B + threadIdx.x * SCAN_V;
  int v[SCAN_V];
  unsigned long long s = 0;
  load_cells8(cnt, base, n, aligned16(cnt), v);
#pragma unroll
  for (int j = 0; j < SCAN_V; j++) s += scan_value(v[j]);
  if (sum_sq) {  // sum of count^2 = the work of every point scanning its own cell: how crowded the cells are (a heuristic, float is plenty)
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < SCAN_V; j++) q += (float)v[j] * (float)v[j];
    for (int o = WAVE / 2; o > 0; o >>= 1) q += __shfl_down(q, o);
    if ((threadIdx.x & (WAVE - 1)) == 0 && q > 0.f) atomicAdd(sum_sq, q);
  }
  for (int o = WAVE / 2; o > 0; o >>= 1) s += __shfl_down(s, o);
  if ((threadIdx.x & (WAVE - 1)) == 0) part[threadIdx.x / WAVE] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long t = 0;
#pragma unroll
    for (int w = 0; w < SCAN_T / WAVE; w++) t += part[w];
    block_sums[blockIdx.x] = t;
  }
}

__global__ void k_cells_scan_sums(unsigned long long* sums, int nb, int* __restrict__ nvox, int prio) {
  wave_prio(prio);  // single block, in-place exclusive scan
  __shared__ unsigned long long carry_s;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  for (int base = 0; base < nb; base += SCAN_T) {
    const int i = base + threadIdx.x;
    const unsigned long long v = i < nb ? sums[i] : 0;
    unsigned long long tot;
    const unsigned long long ex = block_exclusive_scan64(v, &tot);
    const unsigned long long carry = carry_s;
    if (i < nb) sums[i] = ex + carry;
    __syncthreads();
    if (threadIdx.x == 0) carry_s = carry + tot;
    __syncthreads();
  }
  if (threadIdx.x == 0 && nvox) *nvox = (int)(carry_s >> 32);
}

// kSelf: block_sums holds the workgroups' TOTALS and this workgroup sums those before it itself (strided loads + one reduction: no
// single-workgroup kernel in between); else block_sums has been scanned in place (k_cells_scan_sums) and holds the prefix.
template <bool kSelf>
__global__ void __launch_bounds__(SCAN_T) k_cells_scan_write(int* __restrict__ cnt, int* __restrict__ start, int n,
                                                             const unsigned long long* __restrict__ block_sums, int nb, int* __restrict__ cell_voxel,
                                                             int* __restrict__ nvox, int prio) {
  wave_prio(prio);
  __shared__ unsigned long long part[SCAN_T / WAVE];
  __shared__ unsigned long long pre_s;
  const int base = blockIdx.x * SCAN_B + threadIdx.x * SCAN_V;
  int v[SCAN_V];  // the entries are fetched first: their loads overlap the prefix of the totals
  unsigned long long s = 0;
  const bool wide = aligned16(cnt) && aligned16(start) && (!cell_voxel || aligned16(cell_voxel));
  load_cells8(cnt, base, n, wide, v);
#pragma unroll
  for (int j = 0; j < SCAN_V; j++) s += scan_value(v[j]);
  if (kSelf) {
    unsigned long long acc = 0;
    for (int j = threadIdx.x; j < (int)blockIdx.x; j += SCAN_T) acc += block_sums[j];
    for (int o = WAVE / 2; o > 0; o >>= 1) acc += __shfl_down(acc, o);
    if ((threadIdx.x & (WAVE - 1)) == 0) part[threadIdx.x / WAVE] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
      unsigned long long t = 0;
      for (int w = 0; w < SCAN_T / WAVE; w++) t += part[w];
      pre_s = t;
    }
    __syncthreads();
  } else if (threadIdx.x == 0) {
    pre_s = block_sums[blockIdx.x];
  }
  const bool whole = wide && base + SCAN_V <= n - 1;  // all eight entries are cells (the sentinel, entry n - 1, has no cell_voxel)
  // consumed: the counters are left clean for the next cloud, no fill kernel per frame
  if (whole) {
    const int4 z = make_int4(0, 0, 0, 0);
    if (v[0] | v[1] | v[2] | v[3]) *reinterpret_cast<int4*>(cnt + base) = z;
    if (v[4] | v[5] | v[6] | v[7]) *reinterpret_cast<int4*>(cnt + base + 4) = z;
  } else {
#pragma unroll
    for (int j = 0; j < SCAN_V; j++)
      if (base + j < n && v[j]) cnt[base + j] = 0;
  }
  unsigned long long tot;
  unsigned long long ex = block_exclusive_scan64(s, &tot);  // (its barriers publish pre_s)
  ex += pre_s;
  if (kSelf && threadIdx.x == 0 && (int)blockIdx.x == nb - 1 && nvox) *nvox = (int)((pre_s + tot) >> 32);
  if (whole) {
    int st8[SCAN_V], vx8[SCAN_V];
#pragma unroll
    for (int j = 0; j < SCAN_V; j++) {
      st8[j] = (int)(unsigned)ex;
      vx8[j] = v[j] > 0 ? (int)(ex >> 32) : -1;
      ex += scan_value(v[j]);
    }
    *reinterpret_cast<int4*>(start + base) = make_int4(st8[0], st8[1], st8[2], st8[3]);
    *reinterpret_cast<int4*>(start + base + 4) = make_int4(st8[4], st8[5], st8[6], st8[7]);
    if (cell_voxel) {
      *reinterpret_cast<int4*>(cell_voxel + base) = make_int4(vx8[0], vx8[1], vx8[2], vx8[3]);
      *reinterpret_cast<int4*>(cell_voxel + base + 4) = make_int4(vx8[4], vx8[5], vx8[6], vx8[7]);
    }
    return;
  }
#pragma unroll
  for (int j = 0; j < SCAN_V; j++) {
    const int i = base + j;
    if (i < n) {
      start[i] = (int)(unsigned)ex;
      if (cell_voxel && i < n - 1) cell_voxel[i] = v[j] > 0 ? (int)(ex >> 32) : -1;
    }
    ex += scan_value(v[j]);
  }
}

// placement without atomics: slot_of[] came back from k_count's atomicAdd.  The record left at the point's (arrival-order) position in its
// cell carries everything the ranking pass needs -- the original index, this position inside the cell and the cell's population -- so
// that k_rank_gather finds its cell's extent WITHOUT going back to cell_of[] and start[] (two dependent memory round trips of its five;
// the pass is a chain of round trips, not bandwidth: 27 -> us at 1 M points):
//   bits 0..26 original index (n <= 2^27), 27..44 position in the cell, 45..62 population; a cell of 2^18 points or more (a degenerate
//   cloud) stores population 0 and the ranking pass takes the long way for it.
constexpr int kOrdIdxBits = 27, kOrdCntBits = 18;
constexpr unsigned long long kOrdIdxMask = (1ull << kOrdIdxBits) - 1ull;
constexpr int kOrdCntMax = (1 << kOrdCntBits) - 1;
// The neighbour-list cache (KnnCache): does this frame search every query and rebuild the lists?  The map changed (k_count<true> found a
// point that differs from the library's copy, or was told not to trust it), or the previous rebuild ran out of room in a todo list.
// (overflow: two words, written by frame f at [f & 1] and read by frame f + 1 there: a rebuild that overflows again must not change what the
// workgroups of its own launch that start later read)
__device__ __forceinline__ bool cache_redo(const KnnCache& kc) { return *kc.epoch == kc.frame || kc.overflow[(kc.frame - 1) & 1] == kc.frame - 1; }
__global__ void k_place(int n, const int* __restrict__ cell_of, const int* __restrict__ slot_of, const int* __restrict__ start,
                        unsigned long long* __restrict__ order_tmp, int prio, KnnCache kc) {
  wave_prio(prio);
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  // the neighbour-list cache's todo lists are rebuilt by a frame that searches everything (k_knn_sp, the launch after the next): emptied here
  if (kc.nbr && i < kTodoLists && cache_redo(kc)) kc.todo_cnt[i] = 0;
  if (i >= n) return;
  const int c = cell_of[i], slot = slot_of[i];
  const int s0 = start[c], cnt = start[c + 1] - s0;
  const bool big = cnt > kOrdCntMax;
  order_tmp[s0 + slot] = (unsigned long long)(unsigned)i | ((unsigned long long)(big ? 0 : slot) << kOrdIdxBits) |
                         ((unsigned long long)(big ? 0 : cnt) << (kOrdIdxBits + kOrdCntBits));  // unordered inside the cell; k_rank_gather makes the order deterministic
}

// deterministic placement: a point's final slot = cell start + number of same-cell points with a smaller index.
// Sorted points are stored as float4 {x, y, z, original index (int bits)}: one 16-byte load per candidate; P holds n + 4 entries.
__global__ void k_rank_gather(const float* __restrict__ in, int stride_f, int n, const int* __restrict__ cell_of,
                              const int* __restrict__ start, const unsigned long long* __restrict__ order_tmp, float4* __restrict__ P, int* zero_me,
                              int prio, KnnCache kc) {
  wave_prio(prio);
  int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s == 0 && zero_me) { zero_me[0] = 0; zero_me[1] = 0; zero_me[2] = 0; }  // the deferred-query counter of the kNN launch that follows, and the lazy target's two list sizes behind it
  if (s < 4) P[n + s] = make_float4(1.0e30f, 1.0e30f, 1.0e30f, __int_as_float(-1));  // sentinels: infinitely far from every query (k_knn_sp's last quad)
  if (s >= n) return;
  const unsigned long long rec = order_tmp[s];
  const int i = (int)(rec & kOrdIdxMask);
  // the neighbour-list cache (nullable): a frame that rebuilds the lists makes this frame's position the point's RANK; any other frame
  // says where each rank sits now (pos_of) and which rank each position holds (qrank)
  const bool ranked = kc.nbr && !cache_redo(kc);
  int my_rank = 0;
  if (ranked) my_rank = kc.rank_of[i];  // (its round trip overlaps the cell members' below)
  const int cnt = (int)(rec >> (kOrdIdxBits + kOrdCntBits)) & kOrdCntMax;
  int s0, s1;
  if (cnt > 0) {
    s0 = s - ((int)(rec >> kOrdIdxBits) & kOrdCntMax);
    s1 = s0 + cnt;
  } else {  // a cell too crowded for the record's fields
    const int c = cell_of[i];
    s0 = start[c];
    s1 = start[c + 1];
  }
  const float* p = in + (size_t)i * stride_f;
  const float px = p[0], py = p[1], pz = p[2];  // (issued beside the members' loads: nothing below depends on them until the store)
  const unsigned* lo = reinterpret_cast<const unsigned*>(order_tmp);  // the records' low words: index bits 0..26, position bits above
  constexpr unsigned kLoMask = (unsigned)kOrdIdxMask;
  int rank = 0;
  int t = s0;
  for (; t + 8 <= s1; t += 8) {  // eight independent loads in flight: a crowded cell (hundreds of members) is a long serial loop otherwise
    unsigned o[8];
#pragma unroll
    for (int u = 0; u < 8; u++) o[u] = lo[2 * (size_t)(t + u)];
#pragma unroll
    for (int u = 0; u < 8; u++) rank += ((int)(o[u] & kLoMask) < i);
  }
  if (t < s1) {  // 1..7 left: clamped loads (the last member counted again would be wrong: masked by position)
    unsigned o[7];
#pragma unroll
    for (int u = 0; u < 7; u++) o[u] = lo[2 * (size_t)min(t + u, s1 - 1)];
#pragma unroll
    for (int u = 0; u < 7; u++) rank += (t + u < s1 && (int)(o[u] & kLoMask) < i);
  }
  P[s0 + rank] = make_float4(px, py, pz, __int_as_float(i));
  if (kc.nbr) {
    if (ranked) { kc.pos_of[my_rank] = s0 + rank; kc.qrank[s0 + rank] = my_rank; }
    else kc.rank_of[i] = s0 + rank;
  }
}

// ------------------------------------------------------------------------------------------------
// symmetric 3x3 eigen solver (cyclic Jacobi, fp64) -> eigenvector of the smallest eigenvalue.
// Stands in for Eigen::JacobiSVD on the symmetric PSD neighbourhood covariance (fast_gicp_impl.hpp:273):
// U diag(1,1,1e-3) V^T = I - 0.999 n n^T with n that eigenvector.
// ------------------------------------------------------------------------------------------------
// One Jacobi rotation in the (P,Q) plane, R = the third index.  Same rotation as the textbook two-sided product
// A <- G^T A G (t = sgn(theta) / (|theta| + sqrt(theta^2 + 1)), theta = (a_qq - a_pp) / (2 a_pq)) written in its closed
// form: only the five entries that change are touched, and t comes from ONE division and ONE square root
// (t = 2 |a_pq| sgn(theta) / (|d| + sqrt(d^2 + 4 a_pq^2)), d = a_qq - a_pp) -- fp64 divisions and roots are ~15 VALU ops each
// and this routine runs once per point of the map.
template <int P, int Q, int R>
__device__ __forceinline__ void jacobi_rot(double (&A)[3][3], double (&V)[3][3]) {
  const double apq = A[P][Q];
  if (apq == 0.0) return;
  const double d = A[Q][Q] - A[P][P];
  const bool pos = (d == 0.0) || ((d > 0.0) == (apq > 0.0));  // sign of theta, with theta = 0 counted positive
  const double t = (pos ? 2.0 : -2.0) * fabs(apq) / (fabs(d) + sqrt(d * d + 4.0 * apq * apq));
  const double c = rsqrt(t * t + 1.0), s = t * c;
  const double tap = t * apq;
  A[P][P] -= tap;
  A[Q][Q] += tap;
  A[P][Q] = 0.0;
  A[Q][P] = 0.0;
  const double arp = A[R][P], arq = A[R][Q];
  const double nrp = c * arp - s * arq, nrq = s * arp + c * arq;
  A[R][P] = nrp; A[P][R] = nrp;
  A[R][Q] = nrq; A[Q][R] = nrq;
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const double vkp = V[k][P], vkq = V[k][Q];
    V[k][P] = c * vkp - s * vkq;
    V[k][Q] = s * vkp + c * vkq;
  }
}

__device__ __forceinline__ void min_eigenvector(const double S[6], double n[3]) {
  double A[3][3] = {{S[0], S[1], S[2]}, {S[1], S[3], S[4]}, {S[2], S[4], S[5]}};
  double V[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
  for (int sweep = 0; sweep < 60; sweep++) {
    double off = A[0][1] * A[0][1] + A[0][2] * A[0][2] + A[1][2] * A[1][2];
    double diag = A[0][0] * A[0][0] + A[1][1] * A[1][1] + A[2][2] * A[2][2];
    if (off <= 1e-40 * diag || off == 0.0) break;
    jacobi_rot<0, 1, 2>(A, V);
    jacobi_rot<0, 2, 1>(A, V);
    jacobi_rot<1, 2, 0>(A, V);
  }
  double e0 = A[0][0], e1 = A[1][1], e2 = A[2][2];
  // column of the smallest eigenvalue; on exact ties take the LAST one in descending sort order like the
  // oracle's stable selection (ord[] keeps index order for equal values, so the smallest is the highest index)
  int m = 0;
  double em = e0;
  if (e1 <= em) { m = 1; em = e1; }
  if (e2 <= em) { m = 2; em = e2; }
  n[0] = m == 0 ? V[0][0] : (m == 1 ? V[0][1] : V[0][2]);
  n[1] = m == 0 ? V[1][0] : (m == 1 ? V[1][1] : V[1][2]);
  n[2] = m == 0 ? V[2][0] : (m == 1 ? V[2][1] : V[2][2]);
}

// The same eigenvector without sweeps, for the bulk kernel (one call per map point): the smallest root of the characteristic
// polynomial by Newton's iteration from 0 -- for a symmetric positive semi-definite matrix all three roots are real and
// non-negative, so the iteration climbs monotonically to the smallest one, quadratically once it is close (a planar
// neighbourhood: lambda_3 << lambda_2, four or five steps) -- then the null vector of S - lambda I as the largest cross
// product of two of its rows.  About 150 fp64 instructions instead of the ~1700 of five or six Jacobi sweeps.
// Returns false -- the caller falls back to min_eigenvector -- when the iteration does not settle or the two smallest
// eigenvalues are closer than ~1e-3 of the trace (rank of S - lambda I drops to 1, the cross products vanish): there the
// eigenvector is ill-conditioned and only the SAME algorithm reproduces the oracle's choice.
__device__ __forceinline__ bool min_eigenvector_direct(const double S[6], double n[3]) {
  const double tr = S[0] + S[3] + S[5];
  if (!(tr > 0.0)) return false;
  const double inv = 1.0 / tr;  // scaled to trace 1: every entry and every eigenvalue is in [0, 1]
  const double a = S[0] * inv, b = S[1] * inv, c = S[2] * inv, d = S[3] * inv, e = S[4] * inv, f = S[5] * inv;
  const double m0 = d * f - e * e, m1 = b * f - c * e, m2 = b * e - c * d;
  const double c1 = (a * d - b * b) + (a * f - c * c) + m0;  // sum of the principal 2x2 minors
  const double c0 = a * m0 - b * m1 + c * m2;                // determinant
  // p(x) = x^3 - x^2 + c1 x - c0
  double x = 0.0;
  bool settled = false;
  for (int it = 0; it < 12; it++) {
    LAB_COUNT(3);
    const double pv = ((x - 1.0) * x + c1) * x - c0;
    const double dp = (3.0 * x - 2.0) * x + c1;
    const double dx = pv * __builtin_amdgcn_rcp(dp);  // an approximate reciprocal is enough: the iteration corrects itself
    x -= dx;
    // a step this small is the last one that matters (the next would be ~dx^2 / gap); the rounding noise of pv / dp stays below
    // it while the two smallest eigenvalues are at least ~1e-3 apart, which the cross-product test below insists on
    if (!(fabs(dx) > 1.0e-13)) { settled = dp > 0.0; break; }  // also leaves on NaN
  }
  if (!settled) return false;
  const double r0[3] = {a - x, b, c}, r1[3] = {b, d - x, e}, r2[3] = {c, e, f - x};
  const double u[3] = {r0[1] * r1[2] - r0[2] * r1[1], r0[2] * r1[0] - r0[0] * r1[2], r0[0] * r1[1] - r0[1] * r1[0]};
  const double v[3] = {r0[1] * r2[2] - r0[2] * r2[1], r0[2] * r2[0] - r0[0] * r2[2], r0[0] * r2[1] - r0[1] * r2[0]};
  const double w[3] = {r1[1] * r2[2] - r1[2] * r2[1], r1[2] * r2[0] - r1[0] * r2[2], r1[0] * r2[1] - r1[1] * r2[0]};
  const double nu = u[0] * u[0] + u[1] * u[1] + u[2] * u[2], nv = v[0] * v[0] + v[1] * v[1] + v[2] * v[2],
               nw = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
  const bool pu = nu >= nv && nu >= nw, pv2 = !pu && nv >= nw;
  const double best = pu ? nu : (pv2 ? nv : nw);
  if (!(best >= 1.0e-7)) return false;  // |cross| ~ (lambda_1 - lambda_3)(lambda_2 - lambda_3) / 2 at trace 1
  const double s = rsqrt(best);
  n[0] = (pu ? u[0] : (pv2 ? v[0] : w[0])) * s;
  n[1] = (pu ? u[1] : (pv2 ? v[1] : w[1])) * s;
  n[2] = (pu ? u[2] : (pv2 ? v[2] : w[2])) * s;
  return true;
}

// ------------------------------------------------------------------------------------------------
// C2  exact k-nearest neighbours + covariance + normal (fast_gicp_impl.hpp:241-298): helpers shared by the bulk kernels (knn_point_sp,
// knn_point_split: one or four lanes per query, one pass, below), the cooperative kernel (one wave per deferred query) and the
// nearest-neighbour search of the fitness score / ICP.
// ------------------------------------------------------------------------------------------------
// distance from the query to the faces of the cube of cells [c-r, c+r] that are not grid borders; 1e300 if none
__device__ __forceinline__ double cube_bound(const Grid& g, const int c[3], const double q[3], int r) {
  double bound = 1.0e300;
#pragma unroll
  for (int a = 0; a < 3; a++) {
    if (c[a] - r > 0) bound = fmin(bound, q[a] - cell_wall(g, a, c[a] - r));
    if (c[a] + r < g.dim[a] - 1) bound = fmin(bound, cell_wall(g, a, c[a] + r + 1) - q[a]);
  }
  return bound;
}

// ---- shared pieces of the exact search ----------------------------------------------------------------------
template <int KC>
struct TopK {  // the KC smallest squared distances, ascending, in registers (static indexing only)
  // Kept as fp32 BIT PATTERNS: squared distances are >= +0 (never -0, never NaN: non-finite input is rejected by k_bbox),
  // so they order exactly like signed integers, and v_min_i32/v_max_i32 need none of the NaN-quieting v_max_f32 x,x
  // that fminf/fmaxf cost on a loop-carried value (3 ops per slot instead of 2).
  int a[KC];
  __device__ __forceinline__ void init() {
#pragma unroll
    for (int j = 0; j < KC; j++) a[j] = 0x7f800000;  // +inf
  }
  // One compare-exchange per slot, IN PLACE: the carried value ping-pongs between two registers and a[j] is a
  // read-write operand, so the insert is exactly 2 KC VALU ops.  (Written in C++ the conditional insert becomes a phi
  // per slot and the allocator materialises it as 20 extra v_mov; one asm statement per slot gets an s_nop after each.)
  static_assert(KC % 2 == 0, "the carried value must end up back in its first register");
  __device__ __forceinline__ void insert(float xf) {
    int x = __float_as_int(xf), t;
    int j = 0;
#pragma unroll
    for (; j + 10 <= KC; j += 10)
      asm("v_max_i32 %11, %0, %10\n\tv_min_i32 %0, %0, %10\n\tv_max_i32 %10, %1, %11\n\tv_min_i32 %1, %1, %11\n\tv_max_i32 %11, %2, %10\n\tv_min_i32 %2, %2, %10\n\tv_max_i32 %10, %3, %11\n\tv_min_i32 %3, %3, %11\n\tv_max_i32 %11, %4, %10\n\tv_min_i32 %4, %4, %10\n\tv_max_i32 %10, %5, %11\n\tv_min_i32 %5, %5, %11\n\tv_max_i32 %11, %6, %10\n\tv_min_i32 %6, %6, %10\n\tv_max_i32 %10, %7, %11\n\tv_min_i32 %7, %7, %11\n\tv_max_i32 %11, %8, %10\n\tv_min_i32 %8, %8, %10\n\tv_max_i32 %10, %9, %11\n\tv_min_i32 %9, %9, %11"
          : "+v"(a[j]), "+v"(a[j + 1]), "+v"(a[j + 2]), "+v"(a[j + 3]), "+v"(a[j + 4]), "+v"(a[j + 5]), "+v"(a[j + 6]), "+v"(a[j + 7]),
            "+v"(a[j + 8]), "+v"(a[j + 9]), "+v"(x), "=&v"(t));
#pragma unroll
    for (; j < KC; j += 2)
      asm("v_max_i32 %3, %0, %2\n\tv_min_i32 %0, %0, %2\n\tv_max_i32 %2, %1, %3\n\tv_min_i32 %1, %1, %3"
          : "+v"(a[j]), "+v"(a[j + 1]), "+v"(x), "=&v"(t));
  }
  __device__ __forceinline__ bool improves(float x) const { return __float_as_int(x) < a[KC - 1]; }
  __device__ __forceinline__ float at(int j) const { return __int_as_float(a[j]); }  // j must be a compile-time constant
  // a[k-1] without dynamic register indexing (a select chain is turned back into an indexed scratch access by
  // the compiler): the chain is ascending, so a[k-1] is the maximum of the first k entries
  __device__ __forceinline__ float kth(int k) const {
    if (k == KC) return __int_as_float(a[KC - 1]);
    int t = a[0];
#pragma unroll
    for (int j = 1; j < KC; j++) t = (j < k) ? max(t, a[j]) : t;
    return __int_as_float(t);
  }
};

typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float dist2(float px, float py, float pz, const float4& c) {
  // (dx*dx + dy*dy) + dz*dz like flann::L2_Simple<float>; never contracted (-ffp-contract=off).  x and y go through the
  // packed fp32 pipe as ONE pair -- the loaded {x, y} already sits in an aligned register pair -- so the distance is
  // v_pk_add, v_pk_mul, v_sub, v_mul, v_add, v_add: six VALU ops per candidate (each lane rounds exactly as before).
  const f32x2 cxy = {c.x, c.y}, pxy = {px, py};
  const f32x2 d = pxy - cxy;
  const f32x2 dd = d * d;
  const float dz = pz - c.z;
  return (dd.x + dd.y) + dz * dz;
}

// sorted point at a 32-bit BYTE offset: base in SGPRs + one VGPR offset, no 64-bit address arithmetic per candidate
// (n <= 2^27 points is enforced at the API, so 16 n fits)
__device__ __forceinline__ float4 point_at(const float4* __restrict__ P, unsigned byte_off) {
  return *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(P) + byte_off);
}

// Cells of one (y,z) grid row are consecutive in the sorted array, so the candidates of the cube of cells
// [c-r, c+r]^3 are (2r+1)^2 CONTIGUOUS ranges: two start[] loads per row instead of two per cell, and no walk
// through empty cells (the 1-NN searches of C8 / f4).
template <typename F>
__device__ __forceinline__ void for_each_cube_row(const Grid& g, const int c[3], int r, const int* __restrict__ start, F&& f) {
  const int z0 = max(c[2] - r, 0), z1 = min(c[2] + r, g.dim[2] - 1);
  const int y0 = max(c[1] - r, 0), y1 = min(c[1] + r, g.dim[1] - 1);
  const int x0 = max(c[0] - r, 0), x1 = min(c[0] + r, g.dim[0] - 1);
  if (x0 > x1) return;
  for (int z = z0; z <= z1; z++) {
    int y = y0;
    for (; y + 3 <= y1; y += 4) {  // four rows per step: eight independent start[] loads in flight
      int a[4], b[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        a[u] = start[cell_index(g, x0, y + u, z)];
        b[u] = start[cell_index(g, x1, y + u, z) + 1];
      }
#pragma unroll
      for (int u = 0; u < 4; u++) f(a[u], b[u]);
    }
    for (; y <= y1; y++) f(start[cell_index(g, x0, y, z)], start[cell_index(g, x1, y, z) + 1]);
  }
}

#ifndef RGC_KNN_T
#define RGC_KNN_T 256
#endif
#ifndef RGC_SP_ONE_EXIT
#define RGC_SP_ONE_EXIT 0  // (1: the unseeded search's candidate loop with one exit like the seeded one's -- 0.172 against 0.156 ms per 1 M-query launch: it loses there)
#endif
constexpr int KNN_T = RGC_KNN_T;  // (64- and 128-thread workgroups: the same launch time, round 3)
#ifndef RGC_XCD_RUN
#define RGC_XCD_RUN 16
#endif

// Queries the lane-per-query kernel cannot finish -- a sparse neighbourhood whose k-th neighbour is not provably inside the 3x3x3
// block (the search cube must grow), several candidates at exactly the k-th distance (the original index decides), more candidates
// than the keys can number -- are DEFERRED: (query, an upper bound of the k-th distance if one is known, whether the block has been
// scanned) goes to a list that k_knn_coop handles with one WAVE per query, so a wave never idles 63 lanes behind one expensive query.
struct Deferred {
  int* idx;    // query index i, or ~i when the 3x3x3 block (radius 1) has already been scanned
  float* thr;  // k-th distance seen so far (INFINITY if fewer than k candidates)
  int* cnt;
  const int* guard;  // speculative grid only: non-zero = some point did not fit the grid, the cloud will be prepared again -- do nothing
  // lazy target (rgc_set_target_lazy): the map's bulk launch searches only the queries LISTED here -- the points of the cells within the
  // margin of a cell the scan falls into at the guess, compacted by k_footprint (qlist[0 .. *nq)); null: every query, in cell order
  const int* qlist;
  const int* nq;
  // seeds (nullable; a map that is a rigid re-expression of a buffer seen before, rgc_set_target_reframed): per ORIGINAL point index an
  // upper bound of the k-th squared distance the last search of that point found -- where the next search starts pruning (knn_point_seeded).
  // Any value is safe: a search that does not find its k neighbours under the bound runs again without it.
  float* seed;
  float seed_slack;  // how much a k-th distance may have grown since (the coordinates' fp32 rounding in two different frames), metres
  // the neighbour-list cache (KnnCache, rgc_kernels.h; nbr == nullptr: none): the exact searches of a frame that rebuilds the lists write
  // them (the certificate is the top bit of a list's first entry), knn_point_cached reads them
  KnnCache cache;
  int cache_nb;  // workgroups at the FRONT of the map's bulk launch that search the queries on the todo lists (no certificate)
  // the scan's deferred queries resolved INSIDE its bulk launch (coop_stream): the launch's last coop_blocks workgroups take entries as
  // they are published -- slot e = {enc, thr} in one 64-bit word (kSlotEmpty until then, put back by the reader) -- and leave when all
  // bulk workgroups have counted themselves out (*done) and the list is exhausted.  coop_blocks == 0: the lists above, a launch of its own.
  int coop_blocks;
  int* done;
  unsigned long long* slots;
  // knn_point_split only (the four-lane search).  1, a SCAN: every lane sums the moments of its quarter of the neighbours, in key order, and
  // the quad adds up (the scan's sorted order follows ITS grid, whose cell size follows the crowding of the context's previous scan:
  // no order of a scan's neighbours is a property of the cloud alone).  0, a MAP that takes this search (the sparse-map launch): ascending
  // position in the sorted array, the dense kernels' expression -- a map's grid is the voxel grid whichever route prepares it, and its
  // covariances are the same bits through every one of them.
  int split_sums;
};
constexpr unsigned long long kSlotEnd = 0x8080808080808081ull;    // "no entry will ever appear here": written behind the list by the last bulk workgroup
constexpr unsigned long long kSlotEmpty = 0x8080808080808080ull;  // (what hipMemset can write; its low word is no valid entry: |enc| <= 2^27)

// ------------------------------------------------------------------------------------------------
// C2  exact k-nearest neighbours + covariance + normal (fast_gicp_impl.hpp:241-298), the bulk kernel: one lane per query, queries
// in cell order (neighbouring lanes share cells, hence cache lines).  A query's 3x3x3 block of cells is a handful of PIECES --
// contiguous ranges of the sorted array: the nine grid rows of three cells (map), or eight rows and the own row cut into the own
// cell and its two neighbours (scan: 11 pieces) -- each with a lower bound of the distance to anything in it.  Every candidate is
// looked at ONCE and the selection costs a few instructions per candidate:
//  * key = fp32 bit pattern of the squared distance with its low KB bits replaced by the candidate's ORDINAL in the lane's
//    candidate stream.  Keys order like distances (up to 2^-13 relative at KB = 10), and the winning keys name the neighbours: no
//    second pass over the candidates to collect them.
//  * the k + 2 smallest keys live in a sorted register chain whose insert is ONE v_med3_i32 per slot
//    (new a[j] = med3(a[j-1], a[j], x): the slots do not depend on each other), half of the compare-exchange form.
//  * LAZY insertion: a candidate whose key is below the chain's tail is only APPENDED to a small per-lane LDS buffer
//    (compare + masked store); the buffers are drained into the chains when one fills up.  A wave then pays one insert per
//    buffered key of its fullest lane -- not one per candidate that improves ANY of its 64 lanes, which is nearly every one.
//    (min / max / med3 / compare / select are HALF-rate instructions on gfx950 -- profiles/r02_valu_issue.jsonl -- so the insert is
//    what the kernel must ration.)
//  * pieces are visited nearest first and one whose distance bound is not below the chain's tail is skipped: the ball clipping of
//    an exact search, decided with the bound the scan has reached.  A crowded cell next to the sensor is done after its own piece.
//  * the distance may use FMA here: keys only have to ORDER candidates that are at least two key buckets apart.  Where the k-th
//    and (k+1)-th keys are closer than that the two candidates are compared by their exact, uncontracted distances
//    (flann::L2_Simple's expression); a third contender or an exact tie (index order decides) sends the query to the
//    cooperative kernel, as do blocks that cannot prove the k-th distance and blocks with more candidates than ordinals.
//  * the loads of the next four candidates are in flight while four are processed (two register sets taking turns).
// Pieces are walked as ONE per-lane stream of quads (a lane moves to its next piece when the current one is exhausted), so a
// wave's trip count is its longest lane's total, not the sum of per-piece maxima.  A piece's last quad may read up to three
// points past the piece: real points of cells outside the block (pieces that touch in memory hand their tails over, below), so
// they are legitimate candidates and need no masking.
// kTarget separates the two instantiations by NAME (map vs scan) for the profiles and picks the ordinal width.
// ------------------------------------------------------------------------------------------------
#ifndef RGC_ROW_SKIP
#define RGC_ROW_SKIP 1
#endif
constexpr bool kRowSkip = RGC_ROW_SKIP != 0;  // the map's search (whole rows, no piece machinery) skips rows by their distance bound
constexpr int kPieceQuads = 1023;  // a piece's length in quads shares its table entry with the piece's distance bound (fp32, low 10 bits cut)
#ifndef RGC_SPBUF
#define RGC_SPBUF 12
#endif
constexpr int kSpBuf = RGC_SPBUF;   // keys waiting to enter the chain, per lane
#ifndef RGC_SPLOW
#define RGC_SPLOW 8
#endif
constexpr int kSpLow = RGC_SPLOW;   // the map's kernel drains a full buffer down to this many keys (0: to the bottom), knn_point_sp
static_assert(kSpLow >= 0 && kSpLow <= kSpBuf - 4, "a quad of four keys must fit above the drained level");
// Block geometry and LDS layout of one instantiation.  R = block radius in cells, kClip as described at knn_point_sp.
// LDS per lane, as columns [slot][lane]: the append buffer, the piece table (padded with INT_MAX for the ordinal -> piece search)
// and the pieces' first positions in the sorted array.
template <int R, bool kClip>
struct SpShape {
  static constexpr int D = 2 * R + 1;
  static constexpr int NROW = D * D;               // grid rows of the block
  static constexpr int NP = kClip ? NROW + 2 : NROW;  // pieces: the own row in three parts when clipping
  static constexpr int CUM = !kClip ? NP : (NP < 16 ? 16 : (NP < 32 ? 32 : 64));  // table size; with clipping searchable: a power of two > NP
  static constexpr int LDS = kSpBuf + CUM + NP;    // ints per lane
};
// visiting order, nearest first: the own cell and its row, then the other rows by the Chebyshev ring and the distance of their offset
template <int R, bool kClip>
struct SpOrder {
  int p[SpShape<R, kClip>::NP];
  constexpr SpOrder() : p{} {
    constexpr int D = 2 * R + 1, NROW = D * D, OWN = NROW / 2;
    int n = 0;
    if (kClip) { p[n++] = OWN + 1; p[n++] = OWN; p[n++] = OWN + 2; }
    else p[n++] = OWN;
    for (int ring = 1; ring <= R; ring++)
      for (int d2 = 1; d2 <= 2 * R * R; d2++)
        for (int r = 0; r < NROW; r++) {
          const int dy = row_dy(r, D, R), dz = row_dz(r, D, R);
          const int ay = dy < 0 ? -dy : dy, az = dz < 0 ? -dz : dz;
          if ((ay > az ? ay : az) == ring && dy * dy + dz * dz == d2) p[n++] = (kClip && r > OWN) ? r + 2 : r;
        }
  }
};

// Batcher's odd-even merge sort for 32 inputs with the comparators that touch wires >= 24 removed (those wires would hold +inf):
// 132 compare-exchanges sort 24 values (checked on all 2^24 zero-one inputs).  (a << 5 | b), a < b.
constexpr int kSort24N = 132;
__device__ constexpr unsigned short kSort24[kSort24N] = {1, 67, 2, 35, 34, 133, 199, 134, 167, 166, 4, 70, 68, 37, 103, 101, 34, 100, 166, 265, 331, 266, 299, 298, 397, 463, 398, 431, 430, 268, 334, 332, 301, 367, 365, 298, 364, 430, 8, 140, 136, 74, 206, 202, 68, 200, 332, 41, 173, 169, 107, 239, 235, 101, 233, 365, 34, 100, 166, 232, 298, 364, 430, 529, 595, 530, 563, 562, 661, 727, 662, 695, 694, 532, 598, 596, 565, 631, 629, 562, 628, 694, 596, 629, 562, 628, 694, 16, 272, 148, 404, 136, 400, 82, 338, 214, 470, 202, 466, 68, 200, 332, 464, 596, 49, 305, 181, 437, 169, 433, 115, 371, 247, 503, 235, 499, 101, 233, 365, 497, 629, 34, 100, 166, 232, 298, 364, 430, 496, 562, 628, 694};

// The same network cut to 20 wires: 101 compare-exchanges (checked on all 2^20 zero-one inputs).  It puts the k = 20 winners of the map's
// full search into ascending position in the sorted array: the order the moments are summed in (below).
constexpr int kSort20N = 101;
__device__ constexpr unsigned short kSort20[kSort20N] = {1, 67, 2, 35, 34, 133, 199, 134, 167, 166, 4, 70, 68, 37, 103, 101, 34, 100, 166, 265, 331, 266, 299, 298, 397, 463, 398, 431, 430, 268, 334, 332, 301, 367, 365, 298, 364, 430, 8, 140, 136, 74, 206, 202, 68, 200, 332, 41, 173, 169, 107, 239, 235, 101, 233, 365, 34, 100, 166, 232, 298, 364, 430, 529, 595, 530, 563, 562, 16, 272, 136, 400, 82, 338, 202, 466, 68, 200, 332, 464, 49, 305, 169, 433, 115, 371, 235, 499, 101, 233, 365, 497, 34, 100, 166, 232, 298, 364, 430, 496, 562};

// The full 32-wire network (191 compare-exchanges) for the k > 20 instances (KC = 32).
constexpr int kSort32N = 191;
__device__ constexpr unsigned short kSort32[kSort32N] = {1, 67, 2, 35, 34, 133, 199, 134, 167, 166, 4, 70, 68, 37, 103, 101, 34, 100, 166, 265, 331, 266, 299, 298, 397, 463, 398, 431, 430, 268, 334, 332, 301, 367, 365, 298, 364, 430, 8, 140, 136, 74, 206, 202, 68, 200, 332, 41, 173, 169, 107, 239, 235, 101, 233, 365, 34, 100, 166, 232, 298, 364, 430, 529, 595, 530, 563, 562, 661, 727, 662, 695, 694, 532, 598, 596, 565, 631, 629, 562, 628, 694, 793, 859, 794, 827, 826, 925, 991, 926, 959, 958, 796, 862, 860, 829, 895, 893, 826, 892, 958, 536, 668, 664, 602, 734, 730, 596, 728, 860, 569, 701, 697, 635, 767, 763, 629, 761, 893, 562, 628, 694, 760, 826, 892, 958, 16, 280, 272, 148, 412, 404, 136, 400, 664, 82, 346, 338, 214, 478, 470, 202, 466, 730, 68, 200, 332, 464, 596, 728, 860, 49, 313, 305, 181, 445, 437, 169, 433, 697, 115, 379, 371, 247, 511, 503, 235, 499, 763, 101, 233, 365, 497, 629, 761, 893, 34, 100, 166, 232, 298, 364, 430, 496, 562, 628, 694, 760, 826, 892, 958};

// idx[0 .. KC) into ascending order (entries that hold no neighbour: INT_MAX, they end up behind the others).  Neighbour positions in
// ascending position in the sorted array are the order EVERY route sums a neighbourhood's moments in (sp_normal_of): the covariance is
// then a function of the neighbour set alone -- whichever kernel found it, on whichever grid.
template <int KC>
__device__ __forceinline__ void sort_positions(int (&idx)[KC]) {
  static_assert(KC == 20 || KC == 32, "a 20- or a 32-input network");
  constexpr int N = KC == 20 ? kSort20N : kSort32N;
#pragma unroll
  for (int e = 0; e < N; e++) {
    const int ce = KC == 20 ? kSort20[e] : kSort32[e];
    const int a = ce >> 5, b = ce & 31;
    const int lo_ = min(idx[a], idx[b]);
    idx[b] = max(idx[a], idx[b]);
    idx[a] = lo_;
  }
}

// a = med3(below, a, x), IN PLACE: the chain's registers stay where they are across the loops they are carried through (with a
// separate output operand the compiler shuffles all of them at every loop boundary)
__device__ __forceinline__ void med3_inplace(int& a, int below, int x) { asm("v_med3_i32 %0, %1, %0, %2" : "+v"(a) : "v"(below), "v"(x)); }

template <int L>
struct Chain {  // the L smallest keys, ascending, in registers
  int a[L];
  __device__ __forceinline__ void init() {
#pragma unroll
    for (int j = 0; j < L; j++) a[j] = INT_MAX;
  }
  __device__ __forceinline__ void insert(int x) {  // INT_MAX is a no-op
#pragma unroll
    for (int j = L - 1; j >= 1; j--) med3_inplace(a[j], a[j - 1], x);
    asm("v_min_i32 %0, %0, %1" : "+v"(a[0]) : "v"(x));
  }
  // a[idx] for a run-time idx, without indexing (an indexed private array goes to scratch memory): the chain is ascending,
  // so a[idx] is the largest of the first idx + 1 entries
  __device__ __forceinline__ int at(int idx) const {
    int t = a[0];
#pragma unroll
    for (int j = 1; j < L; j++) t = (j <= idx) ? max(t, a[j]) : t;
    return t;
  }
};

__device__ __forceinline__ float dist2_fma(float px, float py, float pz, float cx, float cy, float cz) {
  const float dx = px - cx, dy = py - cy, dz = pz - cz;
  return __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
}

// LDS-address-space pointers (32 bits): through a generic int* every address computation is a 64-bit add
typedef __attribute__((address_space(3))) int lds_int;

// The pieces of query (c, q)'s block, in registers and in MEMORY order: [lo, hi) of the sorted array and a lower bound of the squared
// distance from the query to anything in the piece (rounded down a little).
// kXCut (the seeded search, R = 1, whole rows): tauf is known before the first candidate is fetched, so a row is also cut in x -- the cell
// left (right) of the query's is dropped when nothing in it can be below tauf (its x gap and the row's y / z gaps together): two more
// start[] look-ups per row, a quarter fewer candidates.
template <bool kClip, int R, bool kXCut = false>
__device__ __forceinline__ void sp_pieces(const int* __restrict__ start, const Grid& g, const int (&c)[3], const double (&q)[3],
                                          int (&lo)[SpShape<R, kClip>::NP], int (&hi)[SpShape<R, kClip>::NP], float (&min2)[SpShape<R, kClip>::NP],
                                          float tauf = 0.f) {
  static_assert(!kXCut || (!kClip && R == 1), "the x cut: whole rows of three cells");
  using Shape = SpShape<R, kClip>;
  constexpr int D = Shape::D, NROW = Shape::NROW, NP = Shape::NP, OWN = NROW / 2;
  // ---- the block's pieces in MEMORY order: the D x D grid rows in the order of cell_index (row_dy / row_dz), each the cells cx - R .. cx + R.  kClip: the
  // own row (r = OWN) as three pieces cut at multiples of four points from its start -- left of the own cell, the own cell, right
  // of it -- so pieces OWN, OWN + 1, OWN + 2, and the rows behind it shifted by two. ----
  const int xl = max(c[0] - R, 0), xh = min(c[0] + R, g.dim[0] - 1);
  // distance from the query to the walls of its cell; a row / piece at offset d cells is at least (|d| - 1) cells + that away
  double wlo[3] = {0, 0, 0}, whi[3] = {0, 0, 0};
  if (kClip || kRowSkip || kXCut) {
#pragma unroll
    for (int a = 0; a < 3; a++) {
      const double wall = cell_wall(g, a, c[a]);
      wlo[a] = fmax(q[a] - wall, 0.0);
      whi[a] = fmax(wall + g.res - q[a], 0.0);
    }
  }
  auto axis_gap = [&](int a, int d) { return d == 0 ? 0.0 : (d < 0 ? wlo[a] + (double)(-d - 1) * g.res : whi[a] + (double)(d - 1) * g.res); };
  auto bound2 = [&](double gy, double gz) { return (float)((gy * gy + gz * gz) * (1.0 - 1.0e-6)); };  // rounded down a little: never too high
#pragma unroll
  for (int r = 0; r < NROW; r++) {
    const int dy = row_dy(r, D, R), dz = row_dz(r, D, R);
    const int y = c[1] + dy, z = c[2] + dz;
    const bool in = y >= 0 && y < g.dim[1] && z >= 0 && z < g.dim[2];
    const int yy = in ? y : c[1], zz = in ? z : c[2];
    int a, b;
    if constexpr (kXCut) {
      const int own = cell_index(g, c[0], yy, zz);
      // start[own - 1 .. own + 2] in ONE 16-byte load (the vector-memory pipe works per lane and instruction, not per byte: four dword
      // gathers cost four times this); at the grid's x borders the window is shifted and the missing neighbour's range is empty
      const bool has_l = c[0] > 0, has_r = c[0] < g.dim[0] - 1;
      typedef int i32x4 __attribute__((ext_vector_type(4), aligned(4)));
      const i32x4 s4 = *reinterpret_cast<const i32x4*>(start + (own - (has_l ? 1 : 0)));   // (start[] has ncell + 1 entries and slack behind them)
      const int m1 = has_l ? s4.y : s4.x, m2 = has_l ? s4.z : s4.y, a0 = has_l ? s4.x : m1, b0 = has_r ? (has_l ? s4.w : s4.z) : m2;
      const double g2 = axis_gap(1, dy) * axis_gap(1, dy) + axis_gap(2, dz) * axis_gap(2, dz);
      const bool left = (float)((g2 + wlo[0] * wlo[0]) * (1.0 - 1.0e-6)) < tauf, right = (float)((g2 + whi[0] * whi[0]) * (1.0 - 1.0e-6)) < tauf;
      a = left ? a0 : m1;
      b = right ? b0 : m2;
    } else {
      a = start[cell_index(g, xl, yy, zz)];
      b = start[cell_index(g, xh, yy, zz) + 1];
    }
    if (kClip && r == OWN) {
      const int own = cell_index(g, c[0], c[1], c[2]);
      const int o0 = start[own], o1 = start[own + 1];
      const int a1 = a + ((o0 - a) & ~3), a2 = min(a + ((o1 - a + 3) & ~3), b);
      lo[OWN] = a;      hi[OWN] = a1;     min2[OWN] = bound2(wlo[0], 0.0);
      lo[OWN + 1] = a1; hi[OWN + 1] = a2; min2[OWN + 1] = 0.f;
      lo[OWN + 2] = a2; hi[OWN + 2] = b;  min2[OWN + 2] = bound2(whi[0], 0.0);
    } else {
      const int p = (!kClip || r < OWN) ? r : r + 2;
      lo[p] = in ? a : 0;
      hi[p] = in ? b : 0;
      min2[p] = (kClip || kRowSkip) ? bound2(axis_gap(1, dy), axis_gap(2, dz)) : 0.f;
    }
  }
  // A piece's last quad may read up to 3 points past the piece.  Where the next piece of this block (in memory order) starts closer
  // than that -- sparse layers: a wall's cells are all a grid row holds -- the piece keeps whole quads only and its tail, together
  // with the points in between (real points: harmless candidates), is handed to the next piece (which inherits its distance bound),
  // so no point is ever seen twice.  Past the end of the array sit four sentinel points (k_rank_gather).
  {
    int next_lo[NP];
    int nl = INT_MAX;
#pragma unroll
    for (int p = NP - 1; p >= 0; p--) {
      next_lo[p] = nl;
      if (hi[p] > lo[p]) nl = lo[p];
    }
    int carry = -1;
    float carry_min2 = 0.f;
#pragma unroll
    for (int p = 0; p < NP; p++) {
      if (hi[p] > lo[p]) {
        const int a = carry >= 0 ? carry : lo[p];
        if ((kClip || kRowSkip) && carry >= 0 && carry < lo[p]) min2[p] = fminf(min2[p], carry_min2);
        const int len = hi[p] - a;
        const bool tight = next_lo[p] - hi[p] < 3;
        const int keep = tight ? (len & ~3) : len;
        lo[p] = a;
        hi[p] = a + keep;
        carry = tight ? a + keep : -1;
        carry_min2 = min2[p];
      }
    }
  }
}

// The piece table of query (c, q) in the lane's LDS columns tmix / tlo (stride T), nearest first; returns the number of pieces.
// heavy_piece: some piece is too long for its table entry (or, without clipping, for the row-relative ordinal).
template <bool kClip, int R, int T>
__device__ __forceinline__ int sp_piece_table(const int* __restrict__ start, const Grid& g, const int (&c)[3], const double (&q)[3],
                                              lds_int* const tmix, lds_int* const tlo, bool& heavy_piece) {
  using Shape = SpShape<R, kClip>;
  constexpr int kRowRel = 127;
  constexpr int NP = Shape::NP;
  int lo[NP], hi[NP];
  float min2[NP];
  sp_pieces<kClip, R>(start, g, c, q, lo, hi, min2);
  // ---- piece table, nearest first (SpOrder).  Entry: kClip {distance bound | quads} until the piece is entered, then its first
  // ordinal; otherwise the first ordinal at once. ----
#pragma unroll
  for (int j = 0; j < Shape::CUM; j++) tmix[j * T] = INT_MAX;
  int nr = 0;
  heavy_piece = false;
  constexpr SpOrder<R, kClip> kOrder{};
#pragma unroll
  for (int it = 0; it < NP; it++) {
    const int p = kOrder.p[it];
    const int len = hi[p] - lo[p];
    if (len > 0) {
      const int quads = (len + 3) >> 2;
      heavy_piece |= quads > kPieceQuads;
      if (!kClip) heavy_piece |= quads > (kRowRel + 1) / 4;  // the row's candidates must fit the ordinal's row-relative part
      tlo[nr * T] = lo[p];
      // (whole rows: at most 32 quads, six bits; the row's distance bound, rounded down, in the rest)
      tmix[nr * T] = kClip ? ((__float_as_int(min2[p]) & ~kPieceQuads) | quads) : (kRowSkip ? ((__float_as_int(min2[p]) & ~63) | quads) : quads);
      nr++;
    }
  }
  return nr;
}

// sums -> mean / covariance (fast_gicp_impl.hpp:256-262) -> unit normal of the smallest eigenvalue, stored: the tail every route shares
__device__ __forceinline__ void normal_from_moments(double (&S)[6], double mx, double my, double mz, int k, int i, double* __restrict__ nx,
                                                    double* __restrict__ ny, double* __restrict__ nz) {
  const double inv_k = 1.0 / (double)k;
  mx *= inv_k; my *= inv_k; mz *= inv_k;
  S[0] = S[0] * inv_k - mx * mx; S[1] = S[1] * inv_k - mx * my; S[2] = S[2] * inv_k - mx * mz;
  S[3] = S[3] * inv_k - my * my; S[4] = S[4] * inv_k - my * mz; S[5] = S[5] * inv_k - mz * mz;
  double nrm[3];
  if (!min_eigenvector_direct(S, nrm)) {
    LAB_COUNT(4);
    min_eigenvector(S, nrm);
  }
  nx[i] = nrm[0];
  ny[i] = nrm[1];
  nz[i] = nrm[2];
}

// Mean / covariance (fast_gicp_impl.hpp:256-262) / normal of query (px, py, pz) from its neighbours' positions idx[0 .. k) in the sorted
// array.  One pass: with u_j = p_j - q (exact in fp64: both are fp32 values), cov = sum u u^T / k - ubar ubar^T -- the reference's
// centred sum up to rounding (|u| <= a few cells, so nothing cancels badly), half the gathers of the two-pass form.
// kFull (k == KC, the reference's k = 20) runs without per-neighbour guards: a load under a branch is waited for where the branch
// ends, which would serialise the gathers.
// nbr_out (nullable, kFull only): the neighbours' positions are stored there, 16 bytes at a time (the cache's list of this query: in the
// frame that builds the lists a point's position is its RANK, KnnCache), nbr_flag (0 or kListCertified) in the first one's top bit
template <int KC, bool kFull>
__device__ __forceinline__ void sp_normal_of(const float4* __restrict__ P, const int (&idx)[KC], float px, float py, float pz, int k, int i,
                                             double* __restrict__ nx, double* __restrict__ ny, double* __restrict__ nz, int* __restrict__ nbr_out = nullptr,
                                             int nbr_flag = 0) {
  double S[6] = {0, 0, 0, 0, 0, 0};
  if constexpr (kFull && KC % 4 == 0) {
    if (nbr_out) {
#pragma unroll
      for (int j = 0; j < KC; j += 4) *reinterpret_cast<int4*>(nbr_out + j) = make_int4(idx[j] | (j == 0 ? nbr_flag : 0), idx[j + 1], idx[j + 2], idx[j + 3]);
    }
  }
  const double qx = (double)px, qy = (double)py, qz = (double)pz;
  double mx = 0, my = 0, mz = 0;
#pragma unroll
  for (int j = 0; j < KC; j++) {
    if (kFull || j < k) {
      const float4 cp = point_at(P, (unsigned)idx[j] << 4);
      const double dx = (double)cp.x - qx, dy = (double)cp.y - qy, dz = (double)cp.z - qz;
      mx += dx; my += dy; mz += dz;
      // (explicit fma: the file is compiled without contraction for the sake of dist2(); these sums have no such constraint)
      S[0] = fma(dx, dx, S[0]); S[1] = fma(dx, dy, S[1]); S[2] = fma(dx, dz, S[2]);
      S[3] = fma(dy, dy, S[3]); S[4] = fma(dy, dz, S[4]); S[5] = fma(dz, dz, S[5]);
    }
  }
  normal_from_moments(S, mx, my, mz, k, i, nx, ny, nz);
}


// The certificate of a query's neighbour list (KnnCache): a_up = upper bound of the k-th squared distance as computed in this frame,
// b_lo = lower bound of the (k+1)-th candidate's, bound = distance to the nearest face of the block that is not a grid border (what the
// search proved: every point outside the block is at least that far).  With D the true (frame-independent) distances and e the largest
// error of a computed distance against D in any frame (two points rounded to fp32 in that frame: <= sqrt(3) ulp of the largest coordinate,
// plus the 4e-7 relative of the fp32 expression): members have D <= sqrt(a_up) + e, everything else D >= sqrt(min(b_lo, bound^2)) - e;
// in another frame the computed distances keep their order when the two are more than 2 e apart, i.e. when
//     sqrt(min(b_lo, bound^2)) - sqrt(a_up) > 4 e         (cert_slack = 4 sqrt(3) ulp, times 1.1; the relative part added here).
__device__ __forceinline__ bool list_certified(float a_up, float b_lo, double bound, float cert_slack) {
  const float lim = bound == 1.0e300 ? b_lo : fminf(b_lo, (float)(bound * bound * (1.0 - 1e-5)));
  const float ra = __builtin_sqrtf(a_up), rb = __builtin_sqrtf(lim);
  return rb - ra > cert_slack + 4.0e-6f * rb;
}

// A query that ends without a certificate -- no gap, or handed to the cooperative kernel -- while the lists are attached (a frame that
// searches everything): its list says so, and it goes onto the todo list the later frames search (by rank = its position in this frame:
// the lists outlive the frame's order).  ~1 % of the queries, an atomic each on one of kTodoLists words.
constexpr int kListCertified = (int)0x80000000;
#ifndef RGC_CACHE_XCD_EIGHTHS
#define RGC_CACHE_XCD_EIGHTHS 0  // 1: the list look-ups of each XCD cover one contiguous eighth of the map (measured: 2 % slower than the searches' runs)
#endif
#ifndef RGC_KNN_CACHE
#define RGC_KNN_CACHE 1  // 0: the neighbour-list workgroups are not compiled into k_knn_sp (rgc_api.hip's flag of the same name keeps the host from asking for them)
#endif
template <int KC>
__device__ __forceinline__ void cache_uncertified(const Deferred& df, int rank) {
  if (!df.cache.nbr) return;
  df.cache.nbr[(size_t)rank * KC] = 0;
  const int l = (int)(blockIdx.x % kTodoLists);
  const int e = atomicAdd(&df.cache.todo_cnt[l], 1);
  if (e < df.cache.todo_cap) df.cache.todo[(size_t)l * df.cache.todo_cap + e] = rank;
  else df.cache.overflow[df.cache.frame & 1] = df.cache.frame;  // no room: the next frame searches everything again (cache_redo)
}

// The MAP's search (a leaf-filtered cloud: nothing crowded, the block is nine whole rows).  KB: low key bits that hold the candidate's
// ordinal {table row (high bits) | position in the row (low 7 bits)}: the winning keys give their neighbours' positions with ONE table
// read, and a skipped row renumbers nothing.  (The raw scan's search -- pieces with lazily numbered ordinals, four lanes per query -- is
// knn_point_split below.)
// kExact: the launch is for k == KC (the reference's k = 20): the general-k branches are not even compiled into that instance
template <int KC, int KB, int R, int T, bool kExact>  // T: threads per workgroup = stride of the per-lane LDS columns
__device__ __forceinline__ void knn_point_sp(const float4* __restrict__ P, const int* __restrict__ start, const Grid& g, int n, int k,
                                             int i, int* lds, const Deferred& df, double* __restrict__ nx, double* __restrict__ ny,
                                             double* __restrict__ nz) {
  constexpr bool kClip = false;
  using Shape = SpShape<R, kClip>;
  constexpr int L = KC + 2;
  constexpr int kKeyOrd = (1 << KB) - 1;
  constexpr int kKeyBits = KB;
  // without clipping the ordinal is {table row (high bits) | position in the row (low 7 bits)}: the winning keys give their
  // neighbours' positions with ONE table read; with clipping ordinals are handed out as pieces are entered and searched for
  constexpr int kRowRel = 127;
  static_assert((SpShape<R, kClip>::NP << 7) <= (1 << KB), "ordinal bits: rows x 128");
  lds_int* const buf = (lds_int*)lds;                   // [kSpBuf][T]
  lds_int* const tmix = buf + kSpBuf * T;               // [CUM][T]: before a piece is reached {min distance^2 (fp32, low bits cut) | quads}, after: its first ordinal
  lds_int* const tlo = buf + (kSpBuf + Shape::CUM) * T; // [NP][T]
  const float4 pq = P[i];
  const float px = pq.x, py = pq.y, pz = pq.z;
  const int c[3] = {cell_coord(px, g) - g.minc[0], cell_coord(py, g) - g.minc[1], cell_coord(pz, g) - g.minc[2]};
  const double q[3] = {(double)px, (double)py, (double)pz};
  auto defer = [&](int enc, float thr) {
    const int e = atomicAdd(df.cnt, 1);
    df.idx[e] = enc;
    df.thr[e] = thr;
    if constexpr (kExact && KC == 20) cache_uncertified<KC>(df, i);
  };
  bool heavy_piece = false;
  const int nr = sp_piece_table<kClip, R, T>(start, g, c, q, tmix, tlo, heavy_piece);
  if (heavy_piece) {  // a row of more than 128 candidates does not fit the ordinal's row-relative part
    defer(i, INFINITY);
    return;
  }
  LAB_COUNT(0);
  // ---- one pass over the candidate stream ----
  Chain<L> top;
  top.init();
  int tau = INT_MAX, ri = 0, ordn = 0;
  unsigned off = 0, end = 0;
  lds_int* bp = buf;  // one past the newest buffered key of this lane's column (stride T)
  lds_int* const bp_full = buf + (kSpBuf - 4) * T;
  auto pop = [&]() {
    int key = INT_MAX;
    if (bp != buf) {
      bp -= T;
      key = *bp;
    }
    return key;
  };
  // An insert round costs the wave the same whether one lane or all 64 hold a key.  Buffers are therefore drained only DOWN TO
  // kSpLow keys when one runs full (the tail of a full drain is rounds in which the fullest lane works alone, and at any time some
  // lane of the 64 is in its loose-tail phase), and to the bottom once, at the end of the stream: 65 -> 47 rounds per wave on the
  // c-main map for 1 % more appended keys (scripts/sim_drain.py; counted on the device: profiles/r03_knn_isa_mix.json).
  auto drain = [&](lds_int* floor) {  // the next key is on its way from LDS while the current one walks down the chain
    int key = pop();
    for (;;) {
      const bool more = __any(bp > floor);
      const int nkey = more ? pop() : INT_MAX;
      LAB_COUNT(2);
      top.insert(key);
      if (!more) break;
      key = nkey;
    }
    tau = top.a[L - 1];
  };
  lds_int* const bp_low = buf + kSpLow * T;
  // Software pipeline: the loads of quad q + 1 are issued before quad q is processed, so a wave waits for memory once, not once
  // per quad; two register sets take turns (copying one into the other would wait for the loads just issued).
  // (off, end, ri, ordn) always describe the NEXT quad to fetch.
  struct Quad { float4 p0, p1, p2, p3; int ord; bool on, live; };  // live: the lane's stream is not exhausted (a skipped row gives on = false)
  auto fetch = [&](Quad& q) {
    if (off >= end && ri < nr) {
      // next row -- as an EMPTY one if nothing in it can be among the k + 2 nearest any more (its distance bound is not below the
      // chain's tail): the lane idles for this quad and moves on with the next; ordinals are {row, position}, so skipping renumbers nothing
      const int mix = tmix[ri * T];
      const int quads = kRowSkip ? ((__int_as_float(mix & ~63) >= __int_as_float(tau | kKeyOrd)) ? 0 : (mix & 63)) : mix;
      off = (unsigned)tlo[ri * T] << 4;
      end = off + ((unsigned)quads << 6);
      ordn = ri << 7;
      ri++;
    }
    q.on = off < end;
    q.live = q.on || ri < nr;
    q.ord = ordn;
    // unconditional loads (a lane that has run out of rows reads the sentinels): a load under a branch is waited for at the
    // end of its block, which would put the memory latency back into every iteration
    const unsigned a = q.on ? off : (unsigned)n << 4;
    q.p0 = point_at(P, a); q.p1 = point_at(P, a + 16); q.p2 = point_at(P, a + 32); q.p3 = point_at(P, a + 48);
    off += q.on ? 64u : 0u;
    ordn += q.on ? 4 : 0;
  };
  auto process = [&](const Quad& q) {
    LAB_COUNT(1);
    if (q.on) {
      const int k0 = (__float_as_int(dist2_fma(px, py, pz, q.p0.x, q.p0.y, q.p0.z)) & ~kKeyOrd) | q.ord;
      const int k1 = (__float_as_int(dist2_fma(px, py, pz, q.p1.x, q.p1.y, q.p1.z)) & ~kKeyOrd) | (q.ord + 1);
      const int k2 = (__float_as_int(dist2_fma(px, py, pz, q.p2.x, q.p2.y, q.p2.z)) & ~kKeyOrd) | (q.ord + 2);
      const int k3 = (__float_as_int(dist2_fma(px, py, pz, q.p3.x, q.p3.y, q.p3.z)) & ~kKeyOrd) | (q.ord + 3);
      if (k0 < tau) { *bp = k0; bp += T; }
      if (k1 < tau) { *bp = k1; bp += T; }
      if (k2 < tau) { *bp = k2; bp += T; }
      if (k3 < tau) { *bp = k3; bp += T; }
    }
    if (__any(bp > bp_full)) drain(bp_low);
  };
  Quad qa, qb;
  if (L <= 24) {
    // The first 24 candidates fill the chain whatever they are: instead of 24 appends and as many inserts (the tail is +inf) they go
    // straight into registers -- six quads' loads in flight at once -- and through a fixed sorting network (132 compare-exchanges
    // against ~24 x 22 med3); the 22 smallest become the chain.
    int w[24];
#pragma unroll
    for (int t = 0; t < 6; t++) {
      fetch(qa);
      w[4 * t + 0] = qa.on ? ((__float_as_int(dist2_fma(px, py, pz, qa.p0.x, qa.p0.y, qa.p0.z)) & ~kKeyOrd) | qa.ord) : INT_MAX;
      w[4 * t + 1] = qa.on ? ((__float_as_int(dist2_fma(px, py, pz, qa.p1.x, qa.p1.y, qa.p1.z)) & ~kKeyOrd) | (qa.ord + 1)) : INT_MAX;
      w[4 * t + 2] = qa.on ? ((__float_as_int(dist2_fma(px, py, pz, qa.p2.x, qa.p2.y, qa.p2.z)) & ~kKeyOrd) | (qa.ord + 2)) : INT_MAX;
      w[4 * t + 3] = qa.on ? ((__float_as_int(dist2_fma(px, py, pz, qa.p3.x, qa.p3.y, qa.p3.z)) & ~kKeyOrd) | (qa.ord + 3)) : INT_MAX;
    }
#pragma unroll
    for (int e = 0; e < kSort24N; e++) {
      const int a = kSort24[e] >> 5, b = kSort24[e] & 31;
      const int lo_ = min(w[a], w[b]);
      w[b] = max(w[a], w[b]);
      w[a] = lo_;
    }
#pragma unroll
    for (int j = 0; j < L && j < 24; j++) top.a[j] = w[j];
    tau = top.a[L - 1];
  }
  // (two exits.  The compiler then waits for ALL loads at the head of the loop -- see knn_point_seeded, whose loop has one exit -- but this
  // kernel, five waves per SIMD and bound by instruction issue, is 10 % FASTER that way than with the one-exit loop: measured, not understood)
  fetch(qa);
  for (;;) {
    if (!__any(qa.live)) break;
    fetch(qb);
    process(qa);
#if !RGC_SP_ONE_EXIT
    if (!__any(qb.live)) break;
#endif
    fetch(qa);
    process(qb);
  }
  if (__any(bp != buf)) drain(buf);
  // ---- the k-th neighbour: is it decided by the keys, and is it provably inside the block? ----
  int a_km2, a_km1, a_k, a_kp1;
  if (kExact || k == KC) {
    a_km2 = top.a[KC - 2]; a_km1 = top.a[KC - 1]; a_k = top.a[KC]; a_kp1 = top.a[KC + 1];
  } else {
    a_km2 = k >= 2 ? top.at(k - 2) : -(4 << kKeyBits);
    a_km1 = top.at(k - 1); a_k = top.at(k); a_kp1 = top.at(k + 1);
  }
  if (a_km1 >= 0x7f800000) {  // fewer than k candidates in the block (INT_MAX: empty slot; infinite distance: a sentinel point)
    defer(~i, INFINITY);
    return;
  }
  auto index_of = [&](int key) {  // ordinal -> position in the sorted array
    const int o = key & kKeyOrd;
    return tlo[(o >> 7) * T] + (o & kRowRel);
  };
  bool decided = true, swap = false;
  int kth_key = a_km1;
  if ((a_k >> kKeyBits) - (a_km1 >> kKeyBits) < 2) {
    // the k-th and (k+1)-th candidates are less than two key buckets apart: the truncated (and FMA-rounded) keys cannot order them
    decided = false;
    if ((a_kp1 >> kKeyBits) - (a_k >> kKeyBits) >= 2 && (a_km1 >> kKeyBits) - (a_km2 >> kKeyBits) >= 2) {  // exactly two contenders
      LAB_COUNT(5);
      const float4 p1 = P[index_of(a_km1)], p2 = P[index_of(a_k)];
      const float d1 = dist2(px, py, pz, p1), d2 = dist2(px, py, pz, p2);  // the reference's expression, uncontracted
      decided = d1 != d2;  // an exact tie is decided by the original index: cooperative kernel
      swap = d2 < d1;
      if (swap) kth_key = a_k;
    }
  }
  const float thr_up = __int_as_float(kth_key | kKeyOrd);  // upper bound of the k-th squared distance
  const double bound = cube_bound(g, c, q, R);
  const bool proven = (bound == 1.0e300) || (bound > 0.0 && (double)thr_up < bound * bound * (1.0 - 1e-5));
  if (!proven) {
    // a k-th "neighbour" farther than the block reaches is one of the stray points behind a row: it says nothing about where to look
    defer(~i, (double)thr_up < 3.0 * (R + 1) * (R + 1) * g.res * g.res ? thr_up : INFINITY);
    return;
  }
  if (!decided) {
    defer(i, thr_up);
    return;
  }
  // ---- neighbour positions replace the keys, then mean / covariance (fast_gicp_impl.hpp:256-262) / normal ----
  // k == KC (the reference's k = 20): the neighbours are summed in ascending POSITION in the sorted array -- a property of the neighbour
  // set alone, and the order a seeded search (knn_point_seeded: no chain, rows walked in memory order) admits its keys in: both routes
  // give the same bits.
  int idx[KC];
  if constexpr (kExact) {
    static_assert(KC == 20, "the position sort is a 20-input network");
#pragma unroll
    for (int j = 0; j < KC; j++) idx[j] = index_of(top.a[j]);
    if (swap) idx[KC - 1] = index_of(a_k);
    sort_positions<KC>(idx);
  } else {
    const int idx_k = swap ? index_of(a_k) : 0;
#pragma unroll
    for (int j = 0; j < KC; j++) idx[j] = j < k ? ((swap && j == k - 1) ? idx_k : index_of(top.a[j])) : INT_MAX;
    sort_positions<KC>(idx);  // (any k: the same order as the cooperative kernel's and the four-lane search's)
  }
  const int orig = __float_as_int(pq.w);
  if constexpr (kExact && KC == 20) {
    // (the (k+1)-th candidate is a_k; when the exact distances swapped the two the gap is a key bucket or two: no certificate)
    int* const nbr_out = df.cache.nbr ? df.cache.nbr + (size_t)i * KC : nullptr;
    const bool cert = nbr_out && !swap && list_certified(thr_up, __int_as_float(a_k & ~kKeyOrd), bound, df.cache.cert_slack);
    sp_normal_of<KC, true>(P, idx, px, py, pz, k, i, nx, ny, nz, cert ? nbr_out : nullptr, kListCertified);
    if (!cert) cache_uncertified<KC>(df, i);
  } else {
    if (kExact || k == KC) sp_normal_of<KC, true>(P, idx, px, py, pz, k, i, nx, ny, nz);
    else sp_normal_of<KC, false>(P, idx, px, py, pz, k, i, nx, ny, nz);
  }
  if (df.seed) df.seed[orig] = thr_up;
}

// ------------------------------------------------------------------------------------------------
// The map's search SEEDED with a bound (round 5).  A map handed over by rgc_set_target_reframed is the same point set as last time, rigidly
// moved: a point's k-th neighbour distance is what the last search found, up to the fp32 rounding of the coordinates in the two frames
// (df.seed_slack).  With the pruning bound known BEFORE the first candidate is fetched
//  * a grid row whose distance bound is not below it never enters the table (knn_point_sp learns its bound from the candidates: it
//    visits the nearest rows in full and sorts its first 24 candidates whatever they are),
//  * a candidate is admitted iff its key is below the bound -- about k of them: they are APPENDED to the lane's LDS column and that is
//    all; no sorted chain, no insert rounds, no sorting network.  The covariance is a sum over the neighbour SET
//    (fast_gicp_impl.hpp:256-262) and is taken in ascending position in the sorted array -- the order the rows, walked in memory
//    order, deliver the keys in, and the order knn_point_sp sorts its winners into: the two routes give the same bits.
// Exactness does not rest on the seed.  The bound admits keys up to three buckets above the seeded distance; exactly k admitted keys
// whose largest lies two buckets below the bound ARE the k nearest (everything not admitted is two buckets farther: the keys decide,
// as in knn_point_sp); k + 1 admitted keys: the largest goes, or the exact distances decide between the two largest as there; fewer
// than k (a stale seed: the caller overwrote the map in place), more than k + 1, a third contender or an exact tie: the lane returns
// false and runs knn_point_sp, which also makes every deferral decision but "the block cannot prove the k-th distance" (made here from
// the same k-th key, hence the same way).
// LDS per lane: k + 8 key slots (up to k + 3 keys are sorted out here; a quad of candidates appends up to four before the count is looked at) and nine table entries
// {first position << 6 | quads} of the rows to visit.
// ------------------------------------------------------------------------------------------------
constexpr int kSeedExtra = 3;  // keys beyond k the seeded search sorts out itself (more: the full search)
template <int KC>
struct SeedShape {
  static constexpr int BUF = KC + kSeedExtra + 5;  // k + kSeedExtra + 1 = "too many", and a quad appends up to four before the count is clamped
  static constexpr int LDS = BUF + SpShape<1, false>::NP;  // ints per lane
};
#ifndef RGC_SEED_ROW_SKIP
#define RGC_SEED_ROW_SKIP 1
#endif
constexpr bool kSeedRowSkip = RGC_SEED_ROW_SKIP != 0;
// a candidate's coordinates: the first 12 bytes of its 16-byte record (global_load_dwordx3: three quarters of the vector-memory
// pipe's cycles of a 16-byte load; the original index is only needed of the query itself)
#ifndef RGC_SEED_X3
#define RGC_SEED_X3 1
#endif
#if RGC_SEED_X3
struct Cand { float x, y, z; };
__device__ __forceinline__ Cand cand_at(const float4* __restrict__ P, unsigned byte_off) {
  typedef float f32x3 __attribute__((ext_vector_type(3)));
  const f32x3 v = *reinterpret_cast<const f32x3*>(reinterpret_cast<const char*>(P) + byte_off);
  return Cand{v.x, v.y, v.z};
}
#else
typedef float4 Cand;
__device__ __forceinline__ Cand cand_at(const float4* __restrict__ P, unsigned byte_off) { return point_at(P, byte_off); }
#endif
#ifndef RGC_SEED_XCUT
#define RGC_SEED_XCUT 1
#endif
constexpr bool kSeedXCut = RGC_SEED_XCUT != 0;  // rows also cut in x (sp_pieces)
#ifndef RGC_SEED_DEPTH
#define RGC_SEED_DEPTH 2
#endif
constexpr int kSeedDepth = RGC_SEED_DEPTH;  // quads of candidates in flight (2: one being processed, one loading; 3)
constexpr int kSeedMaxPoints = (1 << 26) - 8;  // a table entry holds a position in 26 bits

template <int KC, int KB, int T>
__device__ __forceinline__ bool knn_point_seeded(const float4* __restrict__ P, const int* __restrict__ start, const Grid& g, int n, int i, int* lds,
                                                 const Deferred& df, double* __restrict__ nx, double* __restrict__ ny, double* __restrict__ nz) {
  constexpr int R = 1;
  using Shape = SpShape<R, false>;
  constexpr int NP = Shape::NP;
  constexpr int kKeyOrd = (1 << KB) - 1;
  constexpr int kRowRel = 127;
  static_assert((NP << 7) <= (1 << KB), "ordinal bits: rows x 128");
  lds_int* const buf = (lds_int*)lds;              // [KC + 6][T] admitted keys, in stream order
  lds_int* const tab = buf + SeedShape<KC>::BUF * T;  // [NP][T]
  const float4 pq = P[i];
  const float px = pq.x, py = pq.y, pz = pq.z;
  const int orig = __float_as_int(pq.w);
  const float seed = df.seed[orig];
  if (!(seed < 1.0e30f)) { LAB_DECLINE(1); return false; }  // never searched (or deferred every time)
  // keys below tkey are admitted: the seeded k-th distance, grown by the slack, rounded up, plus three key buckets
  // (a search that also issues the certificates -- a frame that rebuilds the neighbour lists -- admits a little more: exactly k keys
  // under the bound then PROVES the gap a certificate needs, instead of leaving it undecided for every other query)
  const float rs = __builtin_sqrtf(seed) + df.seed_slack + (df.cache.nbr ? 1.5f * df.cache.cert_slack + 8.0e-6f * __builtin_sqrtf(seed) : 0.f);
  const int tkey = ((__float_as_int(rs * rs * 1.000001f) >> KB) + 3) << KB;
  const float tauf = __int_as_float(tkey);
  const int c[3] = {cell_coord(px, g) - g.minc[0], cell_coord(py, g) - g.minc[1], cell_coord(pz, g) - g.minc[2]};
  const double q[3] = {(double)px, (double)py, (double)pz};
  int nv = 0;
  {
    int lo[NP], hi[NP];
    float min2[NP];
    sp_pieces<false, R, kSeedXCut>(start, g, c, q, lo, hi, min2, tauf);
    bool heavy = false;
    // rows in MEMORY order (the bound is known: nothing is gained by visiting the nearest first): the admitted keys then come out in
    // ascending position in the sorted array -- the order the moments are summed in on every route
#pragma unroll
    for (int p = 0; p < NP; p++) {
      const int len = hi[p] - lo[p];
      const int quads = (len + 3) >> 2;
      heavy |= quads > (kRowRel + 1) / 4;
      if (len > 0 && (!kSeedRowSkip || min2[p] < tauf)) {  // (a row handed a neighbour's tail inherited its bound: sp_pieces)
        tab[nv * T] = (lo[p] << 6) | quads;
        nv++;
      }
    }
    if (heavy) { LAB_DECLINE(2); return false; }  // a row of more than 128 candidates: knn_point_sp defers the query
  }
  // ---- one pass over the candidate stream: the loads of quad q + 1 in flight while quad q is processed, as in knn_point_sp ----
  int ri = 0, ordn = 0;
  unsigned off = 0, end = 0;
  lds_int* bp = buf;
  lds_int* const bp_cap = buf + (KC + kSeedExtra + 1) * T;  // that many keys: "too many" whatever follows
  struct Quad { Cand p0, p1, p2, p3; int ord; bool on, live; };
  auto fetch = [&](Quad& qd) {
    if (off >= end && ri < nv) {
      const int e = tab[ri * T];
      off = ((unsigned)e >> 6) << 4;
      end = off + ((unsigned)(e & 63) << 6);
      ordn = ri << 7;
      ri++;
    }
    qd.on = off < end;
    qd.live = qd.on || ri < nv;
    qd.ord = ordn;
    const unsigned a = qd.on ? off : (unsigned)n << 4;  // (unconditional loads: a lane that has run out of rows reads the sentinels)
    qd.p0 = cand_at(P, a); qd.p1 = cand_at(P, a + 16); qd.p2 = cand_at(P, a + 32); qd.p3 = cand_at(P, a + 48);
    off += qd.on ? 64u : 0u;
    ordn += qd.on ? 4 : 0;
  };
  auto process = [&](const Quad& qd) {
    LAB_COUNT(7);
    if (qd.on) {
      const int k0 = (__float_as_int(dist2_fma(px, py, pz, qd.p0.x, qd.p0.y, qd.p0.z)) & ~kKeyOrd) | qd.ord;
      const int k1 = (__float_as_int(dist2_fma(px, py, pz, qd.p1.x, qd.p1.y, qd.p1.z)) & ~kKeyOrd) | (qd.ord + 1);
      const int k2 = (__float_as_int(dist2_fma(px, py, pz, qd.p2.x, qd.p2.y, qd.p2.z)) & ~kKeyOrd) | (qd.ord + 2);
      const int k3 = (__float_as_int(dist2_fma(px, py, pz, qd.p3.x, qd.p3.y, qd.p3.z)) & ~kKeyOrd) | (qd.ord + 3);
      // (the pointer is bumped in place: written as bp += T the compiler adds into a fresh register and moves it back, a VALU instruction per candidate)
      if (k0 < tkey) { *bp = k0; asm("v_add_u32 %0, %1, %0" : "+v"(bp) : "n"(T * 4)); }
      if (k1 < tkey) { *bp = k1; asm("v_add_u32 %0, %1, %0" : "+v"(bp) : "n"(T * 4)); }
      if (k2 < tkey) { *bp = k2; asm("v_add_u32 %0, %1, %0" : "+v"(bp) : "n"(T * 4)); }
      if (k3 < tkey) { *bp = k3; asm("v_add_u32 %0, %1, %0" : "+v"(bp) : "n"(T * 4)); }
      bp = bp < bp_cap ? bp : bp_cap;
    }
  };
  // ONE exit, at the head of the loop: a second one between the two halves makes the structurizer route both through a common latch
  // block, where the compiler's scoreboard merges "the first set's loads are pending" with "the second set's are" and puts an
  // s_waitcnt vmcnt(0) at the loop's head -- every other trip waited for the loads it had just issued (the loop of rounds 2-4 did).
  // A wave may run one trip more than its longest lane needs (sentinel loads, nothing appended).
  if constexpr (kSeedDepth == 3) {  // two quads' loads in flight behind the one being processed
    Quad qa, qb, qc;
    fetch(qa);
    fetch(qb);
    for (;;) {
      if (!__any(qa.live)) break;
      fetch(qc);
      process(qa);
      fetch(qa);
      process(qb);
      fetch(qb);
      process(qc);
    }
  } else {
    Quad qa, qb;
    fetch(qa);
    for (;;) {
      if (!__any(qa.live)) break;
      fetch(qb);
      process(qa);
      fetch(qa);
      process(qb);
    }
  }
  int m = (int)(bp - buf) / T;
  if (m < KC) {
    // Fewer than k candidates under the bound.  If the bound is not below what the block can prove (a query at the map's sparse border,
    // deferred last time too: its seed is the k-th distance the cooperative search found, beyond the block), the full search would find a
    // k-th key >= tkey, hence a k-th distance it cannot prove either, and defer the query as "block scanned, not enough": so does this.
    // (Its hint -- the k-th candidate of the block -- is not known here; the seeded distance serves: the cooperative search re-derives
    // its bound from what it finds, and its result does not depend on the hint.)
    const double bound = cube_bound(g, c, q, R);
    if (bound != 1.0e300 && !((double)tauf < bound * bound * (1.0 - 1e-5))) {
      const int e = atomicAdd(df.cnt, 1);
      df.idx[e] = ~i;
      df.thr[e] = rs * rs * 1.000001f;
      cache_uncertified<KC>(df, i);
      return true;
    }
    LAB_DECLINE(3);
    return false;
  }
  if (m > KC + kSeedExtra) { LAB_DECLINE(4); return false; }
  // two or three keys too many (a lane in ten thousand): the largest go, in LDS, until k + 1 are left; the last one taken out is the (k+2)-th
  int kp2 = tkey;  // the (k+2)-th smallest key, or (nothing taken out) a lower bound of it
  bool kp2_known = false;
  while (m > KC + 1) {
    int mxv = -1, mxj = 0;
    for (int j = 0; j < m; j++) {
      const int x = buf[j * T];
      if (x > mxv) { mxv = x; mxj = j; }
    }
    for (int j = mxj; j + 1 < m; j++) buf[j * T] = buf[(j + 1) * T];
    m--;
    kp2 = mxv;
    kp2_known = true;
  }
  const bool extra = m > KC;
  // ---- the admitted keys, their largest two and where the largest sits ----
  int w[KC + 1];
#pragma unroll
  for (int j = 0; j <= KC; j++) w[j] = buf[j * T];
  if (!extra) w[KC] = -1;  // (keys are non-negative)
  int mx = w[0], mx2 = -1, pos = 0;
#pragma unroll
  for (int j = 1; j <= KC; j++) {
    const bool gt = w[j] > mx;
    const int t2 = max(mx2, w[j]);
    mx2 = gt ? mx : t2;
    pos = gt ? j : pos;
    mx = max(mx, w[j]);
  }
  int kth_key = extra ? mx2 : mx;
  int drop = extra ? pos : KC;  // the slot that is not a neighbour
  const int next_bucket = extra ? (mx >> KB) : (tkey >> KB);  // of the (k+1)-th key, or below it
  auto index_of = [&](int key) {
    const int o = key & kKeyOrd;
    return (int)((unsigned)tab[(o >> 7) * T] >> 6) + (o & kRowRel);
  };
  // undecided: knn_point_sp's verdict on the same four keys around the k-th (a third contender, an exact tie) -- the cooperative kernel's
  // tie rule decides, from the same entry (query, its k-th key as the bound) the full search would have made
  bool undecided = false;
  if (next_bucket - (kth_key >> KB) < 2) {  // the keys cannot order the k-th and the (k+1)-th candidate
    if (!extra) { LAB_DECLINE(5); return false; }  // (the seed was too tight to tell)
    int mx3 = -1, pos2 = 0;
#pragma unroll
    for (int j = 0; j <= KC; j++) {
      pos2 = w[j] == mx2 ? j : pos2;
      mx3 = (w[j] != mx && w[j] != mx2) ? max(mx3, w[j]) : mx3;
    }
    if ((kp2 >> KB) - (mx >> KB) < 2) {  // the (k+2)-th contends too -- if that is what kp2 is
      if (!kp2_known) { LAB_DECLINE(6); return false; }
      undecided = true;
    } else if ((mx2 >> KB) - (mx3 >> KB) < 2) {  // ... or the (k-1)-th
      undecided = true;
    } else {
      LAB_COUNT(5);
      const float4 p1 = P[index_of(mx2)], p2 = P[index_of(mx)];
      const float d1 = dist2(px, py, pz, p1), d2 = dist2(px, py, pz, p2);  // the reference's expression, uncontracted
      undecided = d1 == d2;  // an exact tie: the original index decides
      if (d2 < d1) { kth_key = mx; drop = pos2; }
    }
  }
  const float thr_up = __int_as_float(kth_key | kKeyOrd);  // upper bound of the k-th squared distance
  const double bound = cube_bound(g, c, q, R);
  const bool proven = (bound == 1.0e300) || (bound > 0.0 && (double)thr_up < bound * bound * (1.0 - 1e-5));
  if (!proven) {  // (knn_point_sp's decision and its entry)
    const int e = atomicAdd(df.cnt, 1);
    df.idx[e] = ~i;
    df.thr[e] = (double)thr_up < 3.0 * (R + 1) * (R + 1) * g.res * g.res ? thr_up : INFINITY;
    cache_uncertified<KC>(df, i);
    return true;
  }
  if (undecided) {
    const int e = atomicAdd(df.cnt, 1);
    df.idx[e] = i;
    df.thr[e] = thr_up;
    cache_uncertified<KC>(df, i);
    return true;
  }
  int idx[KC];
#pragma unroll
  for (int j = 0; j < KC; j++) idx[j] = index_of(j >= drop ? w[j + 1] : w[j]);
  // the (k+1)-th candidate: exactly k admitted -- anything not admitted, >= tkey; k + 1 admitted -- the dropped key, mx (when the exact
  // distances made mx the k-th instead, the two are a key bucket or two apart: no certificate)
  int* const nbr_out = df.cache.nbr ? df.cache.nbr + (size_t)i * KC : nullptr;
  const bool cert = nbr_out && (!extra || kth_key == mx2) &&
                    list_certified(thr_up, __int_as_float((extra ? mx : tkey) & ~kKeyOrd), bound, df.cache.cert_slack);
  sp_normal_of<KC, true>(P, idx, px, py, pz, KC, i, nx, ny, nz, cert ? nbr_out : nullptr, kListCertified);
  if (!cert) cache_uncertified<KC>(df, i);
  df.seed[orig] = thr_up;
  return true;
}

// ------------------------------------------------------------------------------------------------
// A certified query of an unchanged map (KnnCache): its k neighbours are the ones its last exact search found -- their ORIGINAL indices
// -- by rank -- are in the list, their positions in this frame's sorted array in pos_of[].  No search: twenty look-ups, the positions put into ascending
// order (the order every route sums the moments in: the same bits as a search would give), the moments.  Returns false -- nothing done --
// for a query without a certificate (it is on a todo list: searched by the launch's first workgroups, k_knn_sp).
// ------------------------------------------------------------------------------------------------
template <int KC>
__device__ __forceinline__ bool knn_point_cached(const float4* __restrict__ P, int i, const Deferred& df, double* __restrict__ nx,
                                                 double* __restrict__ ny, double* __restrict__ nz) {
  static_assert(KC == 20, "the position sort is a 20-input network");
  const float4 pq = P[i];
  const int4* const L = reinterpret_cast<const int4*>(df.cache.nbr + (size_t)df.cache.qrank[i] * KC);
  int idx[KC];
#pragma unroll
  for (int j = 0; j < KC; j += 4) {
    const int4 o = L[j >> 2];
    idx[j] = o.x; idx[j + 1] = o.y; idx[j + 2] = o.z; idx[j + 3] = o.w;
  }
  if (idx[0] >= 0) return false;  // no certificate (kListCertified): on the todo list
  idx[0] &= ~kListCertified;
#pragma unroll
  for (int j = 0; j < KC; j++) idx[j] = df.cache.pos_of[idx[j]];
#pragma unroll
  for (int e = 0; e < kSort20N; e++) {
    const int a = kSort20[e] >> 5, b = kSort20[e] & 31;
    const int lo_ = min(idx[a], idx[b]);
    idx[b] = max(idx[a], idx[b]);
    idx[a] = lo_;
  }
  sp_normal_of<KC, true>(P, idx, pq.x, pq.y, pq.z, KC, i, nx, ny, nz);
  return true;
}

// ------------------------------------------------------------------------------------------------
// The same search with FOUR LANES PER QUERY, for clouds too small to fill the chip with one lane per query (a raw sweep: ~470
// waves for 1024 SIMDs, and the launch lasts as long as its slowest wave -- the 64 queries of a crowded cell next to the sensor,
// 185 us against a median of 34, scripts/lab_wave.py).  The four lanes of a quad walk the SAME pieces in lock-step and take every
// fourth quad of candidates each (a piece is padded to a multiple of four quads, so they reach its end together):
//  * ordinals number the query's whole candidate stream, as if one lane had walked it: keys of the four lanes merge directly;
//  * each lane keeps the Ls = (k + 2) / 2 + 1 smallest keys of ITS candidates.  If every lane holds b = ceil((k + 2) / 4) keys
//    <= X then k + 2 keys of the union are <= X: the largest of the four lanes' b-th keys (two DPP quad permutes, after every
//    drain) bounds what can still matter -- appends and piece skipping use it, so pruning is as sharp as with one chain;
//    the skip decision uses ONLY this shared bound: the four lanes take it alike, their tables stay identical;
//  * at the end every lane merges the four chains (quad broadcasts) into the k + 2 smallest of the union.  A lane whose chain was
//    full and whose tail is below the (k + 2)-th merged key may have dropped a needed key: the query is deferred (rare with the
//    quads dealt round-robin and the deal rotated from piece to piece; the cooperative kernel is exact for anything);
//  * the neighbours' moments are gathered five per lane and summed over the quad; the eigenvector is solved in all four.
// ------------------------------------------------------------------------------------------------
#ifdef RGC_LAB
__device__ int g_lab_why[8];  // developer build: why the scan's bulk kernel deferred (1 piece too long, 2 ordinals, 3 dropped key, 4 < k, 5 unproven, 6 tie)
void lab_why(int* out8) {
  (void)hipDeviceSynchronize();
  (void)hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_lab_why), sizeof(g_lab_why));
  int z[8] = {0};
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_lab_why), z, sizeof(z));
}
#endif
// kFinal = false (the scan's bulk launch, block radius R): a query whose block holds fewer than k points, or whose k-th neighbour is not
// provably inside the block, is NOT deferred: the function returns true and the caller runs the search again on the next larger block
// (k_knn_sp: R = 1, then R = 2 -- the same four lanes, at once) before anything goes to the cooperative kernel.
template <int KC, int KB, int R, int T, bool kExact, bool kFinal = true>
__device__ __forceinline__ bool knn_point_split(const float4* __restrict__ P, const int* __restrict__ start, const Grid& g, int n, int k,
                                                int i, int sub, int* lds, const Deferred& df, double* __restrict__ nx,
                                                double* __restrict__ ny, double* __restrict__ nz) {
  constexpr bool kClip = true;
  constexpr int S = 4;
  using Shape = SpShape<R, kClip>;
  constexpr int L = KC + 2, Ls = (KC + 2) / 2 + 1;
  static_assert(Ls <= 24 && KC % S == 0, "initial fill by the 24-input network; neighbours dealt evenly to the quad");
  constexpr int kKeyOrd = (1 << KB) - 1;
  constexpr int kKeyBits = KB;
  lds_int* const buf = (lds_int*)lds;                   // [kSpBuf][T]
  lds_int* const tmix = buf + kSpBuf * T;               // [CUM][T]
  lds_int* const tlo = buf + (kSpBuf + Shape::CUM) * T; // [NP][T]
  const float4 pq = P[i];
  const float px = pq.x, py = pq.y, pz = pq.z;
  const int c[3] = {cell_coord(px, g) - g.minc[0], cell_coord(py, g) - g.minc[1], cell_coord(pz, g) - g.minc[2]};
  const double q[3] = {(double)px, (double)py, (double)pz};
  auto defer = [&](int enc, float thr, int why = 0) {  // (the four lanes of a query take every branch alike: one of them reports)
    if (sub == 0) {
#ifdef RGC_LAB
      atomicAdd(&g_lab_why[why], 1);
#endif
      const int e = atomicAdd(df.cnt, 1);
      if (df.coop_blocks > 0) {  // published to the waves that resolve the deferred queries inside this launch: one 64-bit word, nothing to order
        __hip_atomic_store(&df.slots[e], (unsigned long long)(unsigned)enc | ((unsigned long long)(unsigned)__float_as_int(thr) << 32),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the word has arrived before this wave goes on (and, at the launch's end, counts itself out)
      } else {
        df.idx[e] = enc;
        df.thr[e] = thr;
      }
    }
  };
  bool heavy_piece = false;
  const int nr = sp_piece_table<kClip, R, T>(start, g, c, q, tmix, tlo, heavy_piece);
  if (heavy_piece) {
    defer(i, INFINITY, 1);
    return false;
  }
  // ---- one pass over this lane's quarter of the candidate stream ----
  Chain<Ls> top;
  top.init();
  const int bi = (k + 2 + S - 1) / S - 1;
  int shared = INT_MAX;  // upper bound of the (k + 2)-th smallest key of the union, the same in the four lanes
  int tau = INT_MAX;     // min(own tail, shared): what an append must beat
  auto refresh = [&]() {
    shared = quad_max_i((kExact || k == KC) ? top.a[(KC + 2 + S - 1) / S - 1] : top.at(bi));
    tau = min(top.a[Ls - 1], shared);
  };
  int ri = 0, ordn = 0, ord = 0, jleft = 0;  // next piece, its first ordinal; this lane's next quad: ordinal, slots left in the piece
  unsigned off = 0, end = 0;
  bool overflow = false;
  lds_int* bp = buf;
  lds_int* const bp_full = buf + (kSpBuf - 4) * T;
  auto pop = [&]() {
    int key = INT_MAX;
    if (bp != buf) {
      bp -= T;
      key = *bp;
    }
    return key;
  };
  auto drain = [&]() {
    int key = pop();
    for (;;) {
      const int nkey = pop();
      top.insert(key);
      if (!__any(nkey != INT_MAX)) break;
      key = nkey;
    }
    refresh();
  };
  struct Quad { float4 p0, p1, p2, p3; int ord; bool on, live; };
  auto fetch = [&](Quad& qd) {
    while (jleft == 0 && ri < nr) {
      const int mix = tmix[ri * T];
      tmix[ri * T] = ordn;
      const int quads = mix & kPieceQuads;
      const bool reach = !(__int_as_float(mix & ~kPieceQuads) >= __int_as_float(shared | kKeyOrd));  // INT_MAX: NaN, no skip
      if (reach) {
        if (ordn + 4 * quads > kKeyOrd) {
          overflow = true;
          ri = nr;
          break;
        }
        // quad j of the piece goes to lane (j + piece number) & 3: the far field is many pieces of one or two quads, which would
        // otherwise all land on lane 0 (and overflow its chain)
        const int first = (sub - ri) & (S - 1);
        const unsigned base = (unsigned)tlo[ri * T] << 4;
        off = base + 64u * (unsigned)first;
        end = base + ((unsigned)quads << 6);
        ord = ordn + 4 * first;
        jleft = (quads + S - 1) / S;
        ordn += 4 * quads;
      }
      ri++;
    }
    qd.live = jleft > 0;
    qd.on = qd.live && off < end;
    qd.ord = ord;
    const unsigned a = qd.on ? off : (unsigned)n << 4;  // unconditional loads (the sentinels), as in knn_point_sp
    qd.p0 = point_at(P, a); qd.p1 = point_at(P, a + 16); qd.p2 = point_at(P, a + 32); qd.p3 = point_at(P, a + 48);
    off += qd.live ? 64u * S : 0u;
    ord += qd.live ? 4 * S : 0;
    jleft -= qd.live ? 1 : 0;
  };
  auto key_of = [&](const float4& cp, int o) { return (__float_as_int(dist2_fma(px, py, pz, cp.x, cp.y, cp.z)) & ~kKeyOrd) | o; };
  auto process = [&](const Quad& qd) {
    if (qd.on) {
      const int k0 = key_of(qd.p0, qd.ord), k1 = key_of(qd.p1, qd.ord + 1), k2 = key_of(qd.p2, qd.ord + 2), k3 = key_of(qd.p3, qd.ord + 3);
      if (k0 < tau) { *bp = k0; bp += T; }
      if (k1 < tau) { *bp = k1; bp += T; }
      if (k2 < tau) { *bp = k2; bp += T; }
      if (k3 < tau) { *bp = k3; bp += T; }
    }
    if (__any(bp > bp_full)) drain();
  };
  Quad qa, qb;
  {  // this lane's first 24 candidates through the sorting network (knn_point_sp); the Ls smallest become its chain
    int w[24];
#pragma unroll
    for (int t = 0; t < 6; t++) {
      fetch(qa);
      w[4 * t + 0] = qa.on ? key_of(qa.p0, qa.ord) : INT_MAX;
      w[4 * t + 1] = qa.on ? key_of(qa.p1, qa.ord + 1) : INT_MAX;
      w[4 * t + 2] = qa.on ? key_of(qa.p2, qa.ord + 2) : INT_MAX;
      w[4 * t + 3] = qa.on ? key_of(qa.p3, qa.ord + 3) : INT_MAX;
    }
#pragma unroll
    for (int e = 0; e < kSort24N; e++) {
      const int a = kSort24[e] >> 5, b = kSort24[e] & 31;
      const int lo_ = min(w[a], w[b]);
      w[b] = max(w[a], w[b]);
      w[a] = lo_;
    }
#pragma unroll
    for (int j = 0; j < Ls; j++) top.a[j] = w[j];
    refresh();
  }
  fetch(qa);
  for (;;) {
    if (!__any(qa.live)) break;
    fetch(qb);
    process(qa);
    if (!__any(qb.live)) break;
    fetch(qa);
    process(qb);
  }
  if (__any(bp != buf)) drain();
  if (overflow) {
    defer(i, INFINITY, 2);
    return false;
  }
  // ---- the k + 2 smallest keys of the union, in every lane of the quad ----
  Chain<L> all;
#pragma unroll
  for (int j = 0; j < L; j++) all.a[j] = j < Ls ? quad_perm_i<0x00>(top.a[j]) : INT_MAX;
  auto merge_from = [&](auto ctrl_tag) {
    constexpr int kCtrl = decltype(ctrl_tag)::value;
    bool more = true;  // a chain is ascending: once one of its keys is not below any lane's merged tail, none of the rest is
#pragma unroll
    for (int j = 0; j < Ls; j++) {
      const int x = quad_perm_i<kCtrl>(top.a[j]);
      if (more) {
        more = __any(x < all.a[L - 1]);
        if (more) all.insert(x);
      }
    }
  };
  merge_from(std::integral_constant<int, 0x55>{});
  merge_from(std::integral_constant<int, 0xAA>{});
  merge_from(std::integral_constant<int, 0xFF>{});
  int a_km2, a_km1, a_k, a_kp1;
  if (kExact || k == KC) {
    a_km2 = all.a[KC - 2]; a_km1 = all.a[KC - 1]; a_k = all.a[KC]; a_kp1 = all.a[KC + 1];
  } else {
    a_km2 = k >= 2 ? all.at(k - 2) : -(4 << kKeyBits);
    a_km1 = all.at(k - 1); a_k = all.at(k); a_kp1 = all.at(k + 1);
  }
  if (quad_or_i(top.a[Ls - 1] < a_kp1 ? 1 : 0)) {  // a full chain whose tail ranks inside the merged k + 2: a needed key may have been dropped
    defer(i, INFINITY, 3);
    return false;
  }
  if (a_km1 >= 0x7f800000) {  // fewer than k candidates in the block
    if (!kFinal) return true;  // fewer than k candidates in this block: the next larger one
    defer(~i, INFINITY, 4);
    return false;
  }
  auto index_of = [&](int key) {  // ordinal -> position in the sorted array (the lanes' tables are identical)
    const int o = key & kKeyOrd;
    int r = 0;
#pragma unroll
    for (int st = Shape::CUM / 2; st > 0; st >>= 1)
      if (tmix[(r + st) * T] <= o) r += st;
    return tlo[r * T] + (o - tmix[r * T]);
  };
  bool decided = true, swap = false;
  int kth_key = a_km1;
  if ((a_k >> kKeyBits) - (a_km1 >> kKeyBits) < 2) {  // as in knn_point_sp: two contenders are compared exactly, more are deferred
    decided = false;
    if ((a_kp1 >> kKeyBits) - (a_k >> kKeyBits) >= 2 && (a_km1 >> kKeyBits) - (a_km2 >> kKeyBits) >= 2) {
      const float4 p1 = P[index_of(a_km1)], p2 = P[index_of(a_k)];
      const float d1 = dist2(px, py, pz, p1), d2 = dist2(px, py, pz, p2);
      decided = d1 != d2;
      swap = d2 < d1;
      if (swap) kth_key = a_k;
    }
  }
  const float thr_up = __int_as_float(kth_key | kKeyOrd);
  const double bound = cube_bound(g, c, q, R);
  const bool proven = (bound == 1.0e300) || (bound > 0.0 && (double)thr_up < bound * bound * (1.0 - 1e-5));
  if (!proven) {
    if (!kFinal) return true;  // the k-th neighbour found may not be the true one: the next larger block decides
    defer(~i, (double)thr_up < 3.0 * (R + 1) * (R + 1) * g.res * g.res ? thr_up : INFINITY, 5);
    return false;
  }
  if (!decided) {
    defer(i, thr_up, 6);
    return false;
  }
  // ---- neighbour positions: every lane looks up its quarter (neighbours sub, sub + 4, ...), the quad exchanges them, and every lane holds
  // all of them in ascending position -- then the bulk kernels' expression (sp_normal_of), so that a query's covariance is the same bits
  // whether this search, the map's or the cooperative kernel ends up computing it (which one does depends on the grid's extent) ----
  const int idx_k = swap ? index_of(a_k) : 0;
  // this lane's pick of four registers by bit masks (a select chain on `sub` is turned into an indexed stack array by the compiler)
  const int m_lo = -(sub & 1), m_hi = -(sub >> 1);
  auto pick4 = [](int mlo, int mhi, int a0, int a1, int a2, int a3) {
    const int x = a0 ^ ((a0 ^ a1) & mlo), y = a2 ^ ((a2 ^ a3) & mlo);
    return x ^ ((x ^ y) & mhi);
  };
  int idx[KC];
  auto positions = [&](auto full_tag) {
    constexpr bool kFull = decltype(full_tag)::value;
#pragma unroll
    for (int t = 0; t < KC / S; t++) {  // positions replace the keys (slots 0 .. KC / S - 1: the keys there have been read by then)
      const int key = pick4(m_lo, m_hi, all.a[S * t], all.a[S * t + 1], all.a[S * t + 2], all.a[S * t + 3]);
      const int j = S * t + sub;
      all.a[t] = (swap && j == k - 1) ? idx_k : ((kFull || j < k) ? index_of(key) : INT_MAX);
    }
    if (!df.split_sums) {
      // a MAP (the sparse-map launch): every lane holds all the positions, ascending, and evaluates the dense kernels' expression -- the
      // covariance is the same bits whether this search, the dense one or the cooperative kernel finds the neighbours (which one does
      // depends on the grid's box, i.e. on the route the map came by)
#pragma unroll
      for (int t = 0; t < KC / S; t++) {
        idx[S * t + 0] = quad_perm_i<0x00>(all.a[t]);
        idx[S * t + 1] = quad_perm_i<0x55>(all.a[t]);
        idx[S * t + 2] = quad_perm_i<0xAA>(all.a[t]);
        idx[S * t + 3] = quad_perm_i<0xFF>(all.a[t]);
      }
      sort_positions<KC>(idx);
      sp_normal_of<KC, kFull>(P, idx, px, py, pz, k, i, nx, ny, nz);  // (the four lanes store the same three values)
      return;
    }
    // a SCAN: moments of neighbours sub, sub + 4, ... (in key order) in this lane, summed over the quad
    double S6[6] = {0, 0, 0, 0, 0, 0}, m3[3] = {0, 0, 0};
    const double qx = (double)px, qy = (double)py, qz = (double)pz;
#pragma unroll
    for (int t = 0; t < KC / S; t++) {
      const bool use = kFull || S * t + sub < k;
      const float4 cp = P[use ? all.a[t] : 0];  // (unconditional: position 0 is a valid point; its terms are masked)
      const double dx = use ? (double)cp.x - qx : 0.0, dy = use ? (double)cp.y - qy : 0.0, dz = use ? (double)cp.z - qz : 0.0;
      m3[0] += dx; m3[1] += dy; m3[2] += dz;
      S6[0] = fma(dx, dx, S6[0]); S6[1] = fma(dx, dy, S6[1]); S6[2] = fma(dx, dz, S6[2]);
      S6[3] = fma(dy, dy, S6[3]); S6[4] = fma(dy, dz, S6[4]); S6[5] = fma(dz, dz, S6[5]);
    }
#pragma unroll
    for (int a = 0; a < 3; a++) m3[a] = quad_sum_f64(m3[a]);
#pragma unroll
    for (int a = 0; a < 6; a++) S6[a] = quad_sum_f64(S6[a]);
    normal_from_moments(S6, m3[0], m3[1], m3[2], k, i, nx, ny, nz);  // (the four lanes store the same three values)
  };
  if (kExact || k == KC) positions(std::true_type{});
  else positions(std::false_type{});
  return false;
}

// kTarget names the two instantiations (map vs scan) for the profiles and picks their shape:
//   map  : 3x3x3 block, nine whole rows of at most 128 candidates, 11 ordinal bits {row | position} (one query in ~50 needs the
//          exact tie-break of two contenders) -- a leaf-filtered cloud, nothing to clip;
//   scan : 3x3x3 block as 11 pieces with distance bounds, 12 ordinal bits (4095 candidates) -- a raw sweep: hundreds of points per
//          cell next to the sensor (done after the own piece), metres between neighbours on its far rings (cooperative kernel).
template <bool kTarget> struct SpConfig {
#ifndef RGC_SCAN_KB
#define RGC_SCAN_KB 12
#endif
#ifndef RGC_SCAN_R
#define RGC_SCAN_R 1
#endif
  static constexpr int KB = kTarget ? 11 : RGC_SCAN_KB, R = kTarget ? 1 : RGC_SCAN_R;  // (R = 2, 3 work; for a VLP-16 sweep beside the map's launch they lose to R = 1 at 1 m cells)
  static constexpr bool kClip = !kTarget;
  // The scan's launch runs beside the map's, which fills every CU's LDS with four 256-thread workgroups: one-wave workgroups
  // (18 KiB of LDS each) are admitted as soon as ONE of those retires, a 256-thread one (73 KiB) would wait for two.
  static constexpr int T = kTarget ? KNN_T : WAVE;
};

#if defined(RGC_LAB) || defined(RGC_LAB_BLK)
#define RGC_LAB_BLOCKS 1
__device__ long long g_lab_blk[4 * 16384];  // developer build: {start, end (100 MHz), XCC id, first query} of every workgroup of the MAP's bulk kNN launch
void lab_blocks(long long* out) {
  (void)hipDeviceSynchronize();
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_lab_blk), sizeof(g_lab_blk));
}
#endif
#ifdef RGC_LAB
__device__ long long g_lab_wave[2 * 8192];  // developer build: start / end (100 MHz) of every wave of the scan's bulk kNN launch
void lab_wave_ts(long long* out, hipStream_t s) {
  (void)hipStreamSynchronize(s);
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_lab_wave), sizeof(g_lab_wave));
}
#endif
// kSeeded (the map, k == KC): every query first tries the seeded search (knn_point_seeded: df.seed holds a bound per original point
// index) and runs the full one where that declines.
// The seeded launch runs in smaller workgroups (kSeedT threads): its per-lane LDS columns (37 ints) admit four 256-thread workgroups per CU,
// and a slot freed by a finished wave is refilled only when a whole workgroup's worth is free; smaller ones refill sooner.  128, not 64:
// one-wave workgroups refill so promptly that the SCAN's one-wave kernels, which run beside this launch on the other stream, no longer
// get in -- a frame one at a time 0.354 -> 0.394 ms (the launch alone 0.122 -> 0.120; 256: 0.123 / 0.361).
#ifndef RGC_SEED_T
#define RGC_SEED_T 128
#endif
constexpr int kSeedT = RGC_SEED_T;
#ifndef RGC_SEED_WAVES
#define RGC_SEED_WAVES 1
#endif
#ifndef RGC_SP_WAVES
#define RGC_SP_WAVES 1  // the map's full search: 86 VGPRs and 30 KB of LDS per 256 threads = five waves per SIMD either way
#endif
template <bool kTarget, bool kSeeded> struct SpLaunch : SpConfig<kTarget> {
  static constexpr int T = kSeeded ? kSeedT : SpConfig<kTarget>::T;
  static constexpr int W = kSeeded ? RGC_SEED_WAVES : (kTarget ? RGC_SP_WAVES : 1);  // waves per SIMD the register allocation must leave room for (1: whatever it takes)
};
struct CoopRows {  // per-wave LDS scratch of the cooperative search (coop_run)
  int pref[WAVE + 1];
  int rowa[WAVE];
  int nb[32];
};

// The bulk launch for a SPARSE map (a few keyframes of a 16-beam sensor after the leaf filter: 0.1 points per 1 m cell, where the 3x3x3
// block of the dense-map kernel holds fewer than k points for most queries and 85 % of them went to the cooperative kernel): the
// scan's four-lanes-per-query search on the (2R+1)^3 block.  Same neighbours, same tie rule; the grid stays the voxel grid.
template <int KC, int R, bool kExact>
__global__ void __launch_bounds__(WAVE)
k_knn_sp_wide(const float4* __restrict__ P, const int* __restrict__ start, Grid g, int n, int k, Deferred df, double* __restrict__ nx,
              double* __restrict__ ny, double* __restrict__ nz) {
  extern __shared__ int slist_wide[];
  if (df.guard && *df.guard) return;
  const int t = (int)blockIdx.x * WAVE + (int)threadIdx.x;
  const int i = t >> 2;
  if (i < n) knn_point_split<KC, SpConfig<false>::KB, R, WAVE, kExact>(P, start, g, n, k, i, t & 3, slist_wide + threadIdx.x, df, nx, ny, nz);
}

// ------------------------------------------------------------------------------------------------
// Cooperative exact search: ONE WAVE PER QUERY.  The (2r+1)^2 rows of the search cube are spread over the lanes
// (their start[] loads overlap instead of forming a dependent chain), the candidates of each batch of 64 rows are
// flattened and dealt round-robin to the lanes (a crowded row does not serialise on one lane), every lane keeps
// the KC smallest distances of ITS candidates in a register chain, and the k-th smallest over the wave is found by
// a bit-wise bisection on the fp32 pattern with per-lane counts + a wave sum.  Exactness and the jump to a larger
// cube are decided once per wave.  Ties on the k-th distance: ascending original index, like the CPU path.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ int wave_min_i(int v) { return wave_min(v); }



__device__ __forceinline__ int coop_locate(const CoopRows* sh, int j) {  // sorted-array position of flattened candidate j
  int rr = 0;
#pragma unroll
  for (int step = 32; step > 0; step >>= 1) {  // last row whose prefix is <= j
    const int probe = rr + step;
    if (sh->pref[probe] <= j) rr = probe;
  }
  return sh->rowa[rr] + (j - sh->pref[rr]);
}

// f(s0, valid0, s1, valid1): two candidates per lane per step so that two searches / loads overlap
template <typename F>
__device__ __forceinline__ void coop_for_each_candidate(const Grid& g, const int c[3], int r, const int* __restrict__ start,
                                                        CoopRows* sh, int lane, F&& f) {
  const int z0 = max(c[2] - r, 0), z1 = min(c[2] + r, g.dim[2] - 1);
  const int y0 = max(c[1] - r, 0), y1 = min(c[1] + r, g.dim[1] - 1);
  const int x0 = max(c[0] - r, 0), x1 = min(c[0] + r, g.dim[0] - 1);
  if (x0 > x1 || y0 > y1 || z0 > z1) return;
  const int ny = y1 - y0 + 1, nrows = ny * (z1 - z0 + 1);
  for (int rb = 0; rb < nrows; rb += WAVE) {
    const int t = rb + lane;
    int a = 0, b = 0;
    if (t < nrows) {
      const int y = y0 + t % ny, z = z0 + t / ny;
      a = start[cell_index(g, x0, y, z)];
      b = start[cell_index(g, x1, y, z) + 1];
    }
    const int len = b - a;
    int inc = len;
#pragma unroll
    for (int o = 1; o < WAVE; o <<= 1) {
      const int u = __shfl_up(inc, o);
      if (lane >= o) inc += u;
    }
    const int total = __shfl(inc, WAVE - 1);
    wave_lds_fence();
    sh->pref[lane] = inc - len;
    sh->rowa[lane] = a;
    if (lane == 0) sh->pref[WAVE] = INT_MAX;  // sentinel: probes never run past the last row
    wave_lds_fence();
    for (int jb = 0; jb < total; jb += 2 * WAVE) {
      const int j0 = jb + lane, j1 = jb + WAVE + lane;
      const bool v0 = j0 < total, v1 = j1 < total;
      const int s0 = v0 ? coop_locate(sh, j0) : 0;
      const int s1 = v1 ? coop_locate(sh, j1) : 0;
      f(s0, v0, s1, v1);
    }
  }
}

// k-th smallest squared distance (as an fp32 bit pattern) over the candidates of cube r, one wave.  Every lane keeps the KW
// smallest of ITS candidates; the k-th smallest over the wave is the largest pattern T with #(values < T) < k, found by
// bisection on the bits: the count is a sum of ballot popcounts (compares write SGPR masks, s_bcnt1 adds them, no
// cross-lane traffic).  KW < k is a gamble that pays: candidates are dealt round-robin, so a lane holding more than KW of
// the k nearest is rare (KW = 4, k = 20: ~1.5 % of queries); `overflow` reports that a lane's largest kept value lies
// below T -- it may have dropped something below T -- and the caller repeats the sweep with full-width chains.
template <int KW>
__device__ __forceinline__ unsigned coop_kth(const float4* __restrict__ P, const int* __restrict__ start, const Grid& g, const int c[3], int r,
                                             CoopRows* sh, int lane, float px, float py, float pz, int k, bool& overflow) {
  TopK<KW> top;
  top.init();
  coop_for_each_candidate(g, c, r, start, sh, lane, [&](int s0, bool v0, int s1, bool v1) {
    const float4 c0 = P[s0], c1 = P[s1];
    const float x0 = v0 ? dist2(px, py, pz, c0) : INFINITY;
    const float x1 = v1 ? dist2(px, py, pz, c1) : INFINITY;
    if (top.improves(x0)) top.insert(x0);
    if (top.improves(x1)) top.insert(x1);
  });
  unsigned T = 0;
  for (int bit = 30; bit >= 0; bit--) {
    const unsigned cand = T | (1u << bit);
    if (cand > 0x7F800000u) continue;
    int cnt_lt = 0;
#pragma unroll
    for (int j = 0; j < KW; j++) cnt_lt += __popcll(__ballot(top.a[j] < (int)cand));
    if (cnt_lt < k) T = cand;
  }
  overflow = __any(top.a[KW - 1] < (int)T);
  return T;
}

// The GENERAL covariance route (rgc_set_regularization_method other than PLANE, VoxelAccumulationMode::MULTIPLICATIVE): every point's
// neighbourhood through the cooperative search below, its regularised 3x3 (six doubles, SoA) instead of a unit normal.  Unoptimised by
// design: the odometer never selects it (src/RGC_odometer.cpp:998-1006), it exists so that the reference's setters mean what they say.
struct GenOut { double* c6; int method; int n; };
template <int KC>
__device__ __forceinline__ void cov6_of(const float4* __restrict__ P, const int (&idx)[KC], int k, int i, const GenOut& go);

// One-wave workgroups: the launch runs beside other kernels of the frame (the map's bulk launch fills every CU), and a single wave is
// admitted wherever one SIMD has a slot; the grid is sized by the caller from the previous cloud's deferred count (idle workgroups
// still have to be dispatched: 2048 four-wave workgroups cost 0.25 ms of the scan's critical path when 200 queries were waiting).
// (the body: wave `wave` of `nwaves` takes every nwaves-th entry of the deferred list; sh: this wave's LDS scratch)
// one deferred entry (enc: the query, or ~query when radius 1 is known to be insufficient; thr: the k-th distance seen so far), one wave
template <int KC, bool kTarget, bool kGeneral = false>
__device__ __forceinline__ void coop_one(const float4* __restrict__ P, const int* __restrict__ start, const Grid& g, int k, const Deferred& df,
                                         double* __restrict__ nx, double* __restrict__ ny, double* __restrict__ nz, CoopRows* sh, int lane,
                                         int e, int enc, float thr, const GenOut& go = GenOut{nullptr, 0, 0}) {
  {
#ifdef RGC_LAB
    const long long lab_t0 = wall_clock64();
    int lab_rounds = 0;
#endif
    const int i = enc < 0 ? ~enc : enc;
    int r = enc < 0 ? 1 : 0;  // radius already known to be insufficient
    const float4 pq = P[i];
    const float px = pq.x, py = pq.y, pz = pq.z;
    const int c[3] = {cell_coord(px, g) - g.minc[0], cell_coord(py, g) - g.minc[1], cell_coord(pz, g) - g.minc[2]};
    const double q[3] = {(double)px, (double)py, (double)pz};
    const int rmax = max(max(max(c[0], g.dim[0] - 1 - c[0]), max(c[1], g.dim[1] - 1 - c[1])), max(c[2], g.dim[2] - 1 - c[2]));
    for (;;) {
      // next cube: the smallest one that can prove the current k-th distance, or twice the size if none is known
      int rn;
      if (thr < INFINITY) {
        const double need = sqrt((double)thr) * (1.0 + 1e-5);
        rn = r + 1;
        while (rn < rmax) {
          const double b = cube_bound(g, c, q, rn);
          if (b == 1.0e300 || b > need) break;
          rn++;
        }
      } else {
        rn = r + max(1, (r + 1) / 2);  // fewer than k candidates so far: grow geometrically (x1.5), not x2
      }
      r = min(rn, rmax);
      bool overflow;
#ifdef RGC_LAB
      lab_rounds++;
#endif
      unsigned T = coop_kth<4>(P, start, g, c, r, sh, lane, px, py, pz, k, overflow);
      if (overflow) T = coop_kth<KC>(P, start, g, c, r, sh, lane, px, py, pz, k, overflow);  // full width: nothing relevant can be dropped
      thr = (T >= 0x7F800000u) ? INFINITY : __uint_as_float(T);
      if (r >= rmax) break;  // whole grid scanned
      if (thr < INFINITY) {
        const double bound = cube_bound(g, c, q, r);
        if (bound == 1.0e300) break;
        if (bound > 0.0 && (double)thr < bound * bound * (1.0 - 1e-5)) break;
      }
    }
    // collect: strictly closer neighbours in wave order, then ties in ascending original index
    int m = 0;
    int tie_o = INT_MAX, tie_s = -1;
    auto take = [&](int s, bool valid, const float4& cp) {
      const float x = valid ? dist2(px, py, pz, cp) : INFINITY;
      const unsigned long long mask = __ballot(x < thr);
      if (x < thr) {
        const int pos = m + __popcll(mask & ((1ull << lane) - 1ull));
        if (pos < k) sh->nb[pos] = s;
      }
      m += __popcll(mask);
      const int o = __float_as_int(cp.w);
      if (x == thr && o < tie_o) { tie_o = o; tie_s = s; }
    };
    coop_for_each_candidate(g, c, r, start, sh, lane, [&](int s0, bool v0, int s1, bool v1) {
      const float4 c0 = P[s0], c1 = P[s1];
      take(s0, v0, c0);
      take(s1, v1, c1);
    });
    int last_o = -1;
    while (m < k) {
      // smallest original index among the ties that is larger than the last one taken
      int best_o = wave_min_i(tie_o);
      if (best_o == INT_MAX) break;  // cannot happen for n >= k
      if (tie_o == best_o) sh->nb[m] = tie_s;
      m++;
      last_o = best_o;
      if (m >= k) break;
      tie_o = INT_MAX;
      tie_s = -1;
      coop_for_each_candidate(g, c, r, start, sh, lane, [&](int s0, bool v0, int s1, bool v1) {
        const float4 c0 = P[s0], c1 = P[s1];
        const int o0 = __float_as_int(c0.w), o1 = __float_as_int(c1.w);
        if (v0 && dist2(px, py, pz, c0) == thr && o0 > last_o && o0 < tie_o) { tie_o = o0; tie_s = s0; }
        if (v1 && dist2(px, py, pz, c1) == thr && o1 > last_o && o1 < tie_o) { tie_o = o1; tie_s = s1; }
      });
    }
    wave_lds_fence();
    // The neighbour SET is what the search decides; the order it was collected in depends on the cube the search ended on, hence on the
    // hint (thr) the bulk kernel passed.  The neighbours are put into ascending position in the sorted array: the sums below are then a
    // function of the set alone, whichever route deferred the query (knn_point_seeded passes other hints than knn_point_sp).
    {
      const int mine = lane < k ? sh->nb[lane] : INT_MAX;
      int rank = 0;
      for (int j = 0; j < k; j++) rank += sh->nb[j] < mine;
      wave_lds_fence();
      if (lane < k) sh->nb[rank] = mine;
      wave_lds_fence();
    }
    // neighbourhood mean / covariance / normal (fast_gicp_impl.hpp:256-262): lane 0 evaluates the bulk kernels' expression (sp_normal_of: one
    // pass, neighbours in ascending position) -- a query's covariance is then the same bits whichever kernel ends up computing it, so
    // WHICH queries a launch defers (that depends on the grid's extent, on stray candidates behind a row, on the seeds) cannot show in
    // the results.  (~600 instructions on one lane, 3 us of a deferred query's 15; the lanes' tree sums of rounds 1-4 were another order.)
    if (lane == 0) {
      int idx[KC];
#pragma unroll
      for (int j = 0; j < KC; j++) idx[j] = j < k ? sh->nb[j] : 0;
      if constexpr (kGeneral) {
        cov6_of<KC>(P, idx, k, i, go);
      } else {
        if (k == KC) sp_normal_of<KC, true>(P, idx, px, py, pz, k, i, nx, ny, nz);
        else sp_normal_of<KC, false>(P, idx, px, py, pz, k, i, nx, ny, nz);
      }
      if (kTarget && df.seed) df.seed[__float_as_int(pq.w)] = thr;  // the k-th squared distance itself: where this point's next search starts (knn_point_seeded)
#ifdef RGC_LAB
      if (!kTarget && e < 8192) { g_lab_wave[2 * e] = lab_t0 | ((long long)r << 56) | ((long long)lab_rounds << 48); g_lab_wave[2 * e + 1] = wall_clock64(); }
#endif
    }
    wave_lds_fence();
  }
}

template <int KC, bool kTarget>
__device__ __forceinline__ void coop_run(const float4* __restrict__ P, const int* __restrict__ start, const Grid& g, int k, const Deferred& df,
                                         double* __restrict__ nx, double* __restrict__ ny, double* __restrict__ nz, CoopRows* sh, int lane,
                                         int wave, int nwaves) {
  const int cnt = *df.cnt;
  for (int e = wave; e < cnt; e += nwaves) {
    const int enc = __builtin_amdgcn_readfirstlane(df.idx[e]);
    const float thr = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(df.thr[e])));
    coop_one<KC, kTarget>(P, start, g, k, df, nx, ny, nz, sh, lane, e, enc, thr);
  }
}
// The scan's deferred queries INSIDE its bulk launch (round 5): wave `wave` of the launch's last workgroups takes entries wave, wave +
// nwaves, ... as they appear.  An entry is one 64-bit atomic word {enc, thr}; empty slots hold kSlotEmpty and the reader puts that back,
// so the list is clean for the next cloud.  The last bulk workgroup to count itself out (*done) writes kSlotEnd into the first slot behind
// the list for every waiting wave: a wave that reads it leaves.  Workgroups are dispatched in order, so every bulk workgroup is running
// or finished when the first of these starts: the wait cannot deadlock.  (A launch of its own
// behind the bulk one started 45 us of latency-bound work only when the last bulk wave had left.)
template <int KC>
__device__ __forceinline__ void coop_stream(const float4* __restrict__ P, const int* __restrict__ start, const Grid& g, int k, const Deferred& df,
                            double* __restrict__ nx, double* __restrict__ ny, double* __restrict__ nz, CoopRows* sh, int lane, int wave, int nwaves,
                            int bulk_blocks, int nslots) {
  for (int e = wave;; e += nwaves) {
    if (e >= nslots) return;  // (more waves than the cloud has points)
    unsigned long long v;
    for (;;) {
      v = __hip_atomic_load(&df.slots[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (v != kSlotEmpty) break;
      __builtin_amdgcn_s_sleep(20);
    }
    if (lane == 0) __hip_atomic_store(&df.slots[e], kSlotEmpty, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (v == kSlotEnd) return;
    const int enc = __builtin_amdgcn_readfirstlane((int)(unsigned)v);
    const float thr = __int_as_float(__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32)));
    coop_one<KC, false>(P, start, g, k, df, nx, ny, nz, sh, lane, e, enc, thr);
  }
}
template <int KC, bool kTarget>
__global__ void __launch_bounds__(WAVE)
k_knn_coop(const float4* __restrict__ P, const int* __restrict__ start, Grid g, int k, Deferred df, double* __restrict__ nx,
           double* __restrict__ ny, double* __restrict__ nz) {
  __shared__ CoopRows shm[1];
  wave_prio(!kTarget);
  if (df.guard && *df.guard) return;
  coop_run<KC, kTarget>(P, start, g, k, df, nx, ny, nz, &shm[0], (int)threadIdx.x, (int)blockIdx.x, (int)gridDim.x);
}

template <int KC, bool kTarget, bool kExact, bool kSeeded = false>
__global__ void __launch_bounds__((SpLaunch<kTarget, kSeeded>::T), (SpLaunch<kTarget, kSeeded>::W))
k_knn_sp(const float4* __restrict__ P, const int* __restrict__ start, Grid g, int n, int k, Deferred df,
         double* __restrict__ nx, double* __restrict__ ny, double* __restrict__ nz) {
  extern __shared__ int slist_sp[];  // [SpShape::LDS][T]
  using Cfg = SpLaunch<kTarget, kSeeded>;
  static_assert(!kSeeded || (kTarget && kExact), "seeds: the map's search at k == KC");
  wave_prio(!kTarget);
  if (df.guard && *df.guard) return;
  // XCD-aware block order: workgroups are dealt round-robin over the 8 XCDs (b and b + 8 share one) and queries are in cell order.
  // Each XCD takes runs of kXcdRun CONSECUTIVE query blocks (neighbouring cells: their candidates are re-used out of that XCD's L2),
  // the runs themselves dealt round-robin (whole contiguous eighths of the map differ too much in work: 7 % slower).
  constexpr int kXcdRun = RGC_XCD_RUN * KNN_T / Cfg::T;  // (a run is RGC_XCD_RUN x 256 consecutive queries whatever the workgroup size)
  int b = (int)blockIdx.x, slot = b >> 3;
  const int x = b & 7;
  if constexpr (!kTarget) {  // four lanes per query, queries in cell order
    if (df.coop_blocks > 0 && b >= (int)gridDim.x - df.coop_blocks) {
      // the launch's last workgroups: a wave per deferred query, taken as the searches in front publish them (coop_stream below)
      const int bulk = (int)gridDim.x - df.coop_blocks;
      coop_stream<KC>(P, start, g, k, df, nx, ny, nz, reinterpret_cast<CoopRows*>(slist_sp) + (threadIdx.x / WAVE), (int)threadIdx.x & (WAVE - 1),
                      (b - bulk) * (Cfg::T / WAVE) + (int)threadIdx.x / WAVE, df.coop_blocks * (Cfg::T / WAVE), bulk, n);
      return;
    }
    const int t = b * Cfg::T + (int)threadIdx.x;
    const int i = t >> 2;
#ifdef RGC_LAB
    const long long lab_t0 = wall_clock64();
#endif
    // A query the 3x3x3 block does not settle -- too few points in it, or a k-th neighbour that a closer point outside it could
    // displace: the sparse far field of a sweep, 12 % of a VLP-16's queries -- is searched again on the 5x5x5 block by the same four lanes,
    // at once: 97 % of them settle there, and a wave of the far field has few candidates either way.  (They used to go to the
    // cooperative kernel, a wave per query and four dependent passes each: the longest launch of the scan's preparation.)
    if (i < n && knn_point_split<KC, Cfg::KB, Cfg::R, Cfg::T, kExact, Cfg::R >= 2>(P, start, g, n, k, i, t & 3, slist_sp + threadIdx.x, df, nx, ny, nz)) {
      if constexpr (Cfg::R < 2) knn_point_split<KC, Cfg::KB, 2, Cfg::T, kExact, true>(P, start, g, n, k, i, t & 3, slist_sp + threadIdx.x, df, nx, ny, nz);
    }
#ifdef RGC_LAB
    if (threadIdx.x == 0 && b < 8192) { g_lab_wave[2 * b] = lab_t0; g_lab_wave[2 * b + 1] = wall_clock64(); }
#endif
    if (df.coop_blocks > 0) {  // counted out: everything this workgroup deferred has arrived (each publishing wave waited for its word).
      // No fence: an agent-scope release writes this XCD's whole L2 back and invalidates it -- under the searches still running on it
      // The LAST workgroup to count itself out tells the waiting waves: an "end" word in the first slot behind the list for each of them
      // (they watch their own slots only -- hundreds of waves polling ONE word queue up at its memory channel, in front of the searches'
      // own atomics on the list's counter: the launch took 145 us instead of 55).
      __syncthreads();
      if (threadIdx.x == 0) slist_sp[0] = __hip_atomic_fetch_add(df.done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (int)gridDim.x - df.coop_blocks - 1;
      __syncthreads();
      if (slist_sp[0]) {
        const int cnt = __hip_atomic_load(df.cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int nwaves = df.coop_blocks * (Cfg::T / WAVE);
        for (int j = (int)threadIdx.x; j < nwaves; j += Cfg::T)
          if (cnt + j < n) __hip_atomic_store(&df.slots[cnt + j], kSlotEnd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    return;
  }
  bool cached = false;
  if constexpr (kSeeded && RGC_KNN_CACHE != 0) {
    // The neighbour-list cache (KnnCache).  An unchanged map: the launch's FIRST cache_nb workgroups search the queries on the todo lists
    // (the ones without a certificate: whole waves of them, started first because they live ten times longer), the others take the
    // certified queries' neighbours from their lists (knn_point_cached) and skip the rest.  A frame that rebuilds the lists (cache_redo:
    // the map's first frame, a buffer rewritten in place): everything is searched by the workgroups behind the first cache_nb, as
    // without the cache, and every search leaves its list and certificate -- or its query on a todo list.
    if (df.cache.nbr) {
      const bool redo = cache_redo(df.cache);
      if (b < df.cache_nb) {
        if (redo) return;
        Deferred dfb = df;
        dfb.cache.nbr = nullptr;  // (these searches leave the lists alone: who is on a todo list stays there until the next rebuild)
        // list l = b % kTodoLists, T entries at a time, dealt to the list's workgroups
        const int l = b % kTodoLists, per = df.cache_nb / kTodoLists;
        const int cnt = min(df.cache.todo_cnt[l], df.cache.todo_cap);
        for (int e = (b / kTodoLists) * Cfg::T + (int)threadIdx.x; e < cnt; e += per * Cfg::T) {
          const int q = df.cache.pos_of[df.cache.todo[(size_t)l * df.cache.todo_cap + e]];
          if (!knn_point_seeded<KC, Cfg::KB, Cfg::T>(P, start, g, n, q, slist_sp + threadIdx.x, dfb, nx, ny, nz))
            knn_point_sp<KC, Cfg::KB, Cfg::R, Cfg::T, kExact>(P, start, g, n, k, q, slist_sp + threadIdx.x, dfb, nx, ny, nz);
        }
        return;
      }
      b -= df.cache_nb;  // (a multiple of 8: the XCD of a workgroup is still b & 7)
      slot = b >> 3;
      cached = !redo;
    }
  }
  int i = (((slot / kXcdRun) * 8 + x) * kXcdRun + slot % kXcdRun) * Cfg::T + threadIdx.x;
#ifdef RGC_LAB_BLOCKS
  const long long lab_b0 = wall_clock64();
#endif
  if constexpr (kSeeded) {
    if (cached) {
#if RGC_CACHE_XCD_EIGHTHS
      // every list look-up costs the same: each XCD takes one contiguous eighth of the map (the searches' runs are dealt round-robin because
      // their work differs from region to region) -- neighbouring queries' look-ups then stay in ONE XCD's L2
      i = (x * (((int)gridDim.x - df.cache_nb) >> 3) + slot) * Cfg::T + (int)threadIdx.x;
#endif
      if (i < n) knn_point_cached<KC>(P, i, df, nx, ny, nz);
      return;
    }
    const bool done = i >= n || knn_point_seeded<KC, Cfg::KB, Cfg::T>(P, start, g, n, i, slist_sp + threadIdx.x, df, nx, ny, nz);
    if (!done) {
      LAB_COUNT(6);
      knn_point_sp<KC, Cfg::KB, Cfg::R, Cfg::T, kExact>(P, start, g, n, k, i, slist_sp + threadIdx.x, df, nx, ny, nz);
    }
  } else {
    if (i < n) knn_point_sp<KC, Cfg::KB, Cfg::R, Cfg::T, kExact>(P, start, g, n, k, i, slist_sp + threadIdx.x, df, nx, ny, nz);
  }
#ifdef RGC_LAB_BLOCKS
  if (Cfg::T > WAVE) __syncthreads();
  if (threadIdx.x == 0 && b < 16384) {
    g_lab_blk[4 * b] = lab_b0; g_lab_blk[4 * b + 1] = wall_clock64();
    g_lab_blk[4 * b + 2] = (long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11));  // HW_REG_XCC_ID, bits 0..3
    g_lab_blk[4 * b + 3] = i;
#ifdef RGC_LAB_BLK
    g_lab_blk[4 * b + 3] |= (long long)g_lab_blk_why[b] << 32;
    g_lab_blk_why[b] = 0;
#endif
  }
#endif
}

// Lazy target: the map's search for the LISTED queries only (df.qlist: whole cells, a cell's points are consecutive entries) instead of
// all of them in cell order.  The launch is sized from the previous frame's list and strides over this one's whatever its length.  (Its own
// kernel: the stride loop around the search costs registers -- 107 against 86 -- that the full launch, five waves per SIMD, cannot spare.)
template <int KC, bool kExact, bool kSeeded = false>
__global__ void __launch_bounds__(SpConfig<true>::T)
k_knn_sp_listed(const float4* __restrict__ P, const int* __restrict__ start, Grid g, int n, int k, Deferred df,
                double* __restrict__ nx, double* __restrict__ ny, double* __restrict__ nz) {
  extern __shared__ int slist_spl[];
  using Cfg = SpConfig<true>;
  static_assert(!kSeeded || kExact, "seeds: the map's search at k == KC");
  if (df.guard && *df.guard) return;
  const int nq = *df.nq;
  for (int t = (int)blockIdx.x * Cfg::T + (int)threadIdx.x; t < nq; t += (int)gridDim.x * Cfg::T) {
    const int i = df.qlist[t];
    if constexpr (kSeeded) {
      if (knn_point_seeded<KC, Cfg::KB, Cfg::T>(P, start, g, n, i, slist_spl + threadIdx.x, df, nx, ny, nz)) continue;
    }
    knn_point_sp<KC, Cfg::KB, Cfg::R, Cfg::T, kExact>(P, start, g, n, k, i, slist_spl + threadIdx.x, df, nx, ny, nz);
  }
}

// ------------------------------------------------------------------------------------------------
// C3  Gaussian voxel map (ADDITIVE), fast_vgicp_voxel.hpp:112-121,129-156.  record = { mean xyz, cov00 01 02 11 12 22,
// num } (10 doubles), C_i = I - 0.999 n_i n_i^T.
// (Round 4, measured and dropped: the nine sums of a cell dealt to the cell's lanes instead of all added by its first lane -- each term
// still in the cloud's order -- made the launch 29 -> 60 us with the term-major LDS rows (the lanes of a cell read nine rows at the same
// column: one bank) and 29 -> 53 us with point-major rows; the serial head loop is not what the launch waits for.)
// One lane per POINT (sorted order): every lane puts its point's nine fp64 terms into LDS; the lane holding the
// FIRST point of a cell then adds the cell's terms in ascending order -- the sorted order inside a cell is ascending
// original index, so the sums run in the reference's cloud order, bit for bit what a serial loop gives -- reading LDS
// for the part inside this block and global memory for the few points that spill into the next block.
// (A lane per grid cell, as before, spends 96 % of its lanes on empty cells and serialises one memory round trip per
// point.)  The voxel id of a cell comes from the cell scan (cell_voxel, -1 for empty cells).
// ------------------------------------------------------------------------------------------------
// Lazy target, two passes.  k_footprint: the cells of the map's grid the solve can look up -- every cell within `margin` cells (Chebyshev)
// of the cell a scan point falls into at the guess (the look-up's own arithmetic, linearize_point) -- get this frame's stamp: plain stores,
// nothing is read and nothing cleared between frames (stamps, not flags).  k_lazy_lists: every sorted map point whose cell carries the
// stamp puts itself on the query list of the bulk kNN launch, the first point of such a cell its cell on the list of the voxel pass (one
// atomicAdd per wave and list; counts: [0] listed queries, [1] listed cells, zeroed by k_rank_gather).  The lists' order is the order of
// arrival of the waves; every listed query and cell is computed independently of the others, so the results do not depend on it.
// (Listing from inside the stamping pass -- whoever stamps a cell first appends its points -- made that pass read every cell it visits and
// serialised the appends on a few thousand threads: 8 -> 220 us.)
__global__ void k_footprint(const float* __restrict__ in, int stride_f, int n, Pose T, Grid g, int* __restrict__ need, int stamp, int margin) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float* pp = in + (size_t)i * stride_f;
  const double p0 = (double)pp[0], p1 = (double)pp[1], p2 = (double)pp[2];
  const double q0 = T.R[0] * p0 + T.R[1] * p1 + T.R[2] * p2 + T.t[0];
  const double q1 = T.R[3] * p0 + T.R[4] * p1 + T.R[5] * p2 + T.t[1];
  const double q2 = T.R[6] * p0 + T.R[7] * p1 + T.R[8] * p2 + T.t[2];
  if (!(fabs(q0) <= 1.0e8 && fabs(q1) <= 1.0e8 && fabs(q2) <= 1.0e8)) return;
  const int cx = (int)floor(q0 / g.res - 0.5) - g.minc[0];
  const int cy = (int)floor(q1 / g.res - 0.5) - g.minc[1];
  const int cz = (int)floor(q2 / g.res - 0.5) - g.minc[2];
  const int x0 = max(cx - margin, 0), x1 = min(cx + margin, g.dim[0] - 1);
  const int y0 = max(cy - margin, 0), y1 = min(cy + margin, g.dim[1] - 1);
  const int z0 = max(cz - margin, 0), z1 = min(cz + margin, g.dim[2] - 1);
  for (int z = z0; z <= z1; z++)
    for (int y = y0; y <= y1; y++)
      for (int x = x0; x <= x1; x++) need[cell_index(g, x, y, z)] = stamp;
}
constexpr int kListT = 1024;  // sixteen waves per workgroup: ONE atomicAdd per workgroup and list (same-address atomics cost ~12 ns each across the XCDs --
                              // one per wave was 6 000 of them on one word: 63 us for a pass that moves 20 MB)
// guard (nullable, as in k_knn_sp): the grid was not derived from this cloud and some point did not fit it -- k_count parked that point in cell 0 while P
// keeps its coordinates, so a cell computed from them may lie outside need[] / start[]: nothing is listed, the cloud will be prepared again
__global__ void __launch_bounds__(kListT) k_lazy_lists(const float4* __restrict__ P, int n, Grid g, const int* __restrict__ start, const int* __restrict__ need,
                                                       int stamp, int* __restrict__ qlist, int* __restrict__ cell_list, int* __restrict__ counts,
                                                       const int* __restrict__ guard) {
  __shared__ int wq[kListT / WAVE], wc[kListT / WAVE], base_s[2];
  if (guard && *guard) return;
  const int s = blockIdx.x * kListT + threadIdx.x;
  const int lane = threadIdx.x & (WAVE - 1), w = threadIdx.x / WAVE;
  bool want = false, head = false;
  if (s < n) {
    const float4 cp = P[s];
    const int c = cell_index(g, cell_coord(cp.x, g) - g.minc[0], cell_coord(cp.y, g) - g.minc[1], cell_coord(cp.z, g) - g.minc[2]);
    want = need[c] == stamp;
    head = want && start[c] == s;
  }
  const unsigned long long below = (1ull << lane) - 1ull;
  const unsigned long long mw = __ballot(want), mh = __ballot(head);
  if (lane == 0) { wq[w] = __popcll(mw); wc[w] = __popcll(mh); }
  __syncthreads();
  if (threadIdx.x == 0) {
    int tq = 0, tc = 0;
#pragma unroll
    for (int j = 0; j < kListT / WAVE; j++) { const int a = wq[j], b = wc[j]; wq[j] = tq; wc[j] = tc; tq += a; tc += b; }  // counts -> offsets inside the workgroup
    base_s[0] = tq ? atomicAdd(&counts[0], tq) : 0;
    base_s[1] = tc ? atomicAdd(&counts[1], tc) : 0;
  }
  __syncthreads();
  if (want) qlist[base_s[0] + wq[w] + __popcll(mw & below)] = s;
  if (head) cell_list[base_s[1] + wc[w] + __popcll(mh & below)] = s;
}

#ifndef RGC_GRID_T
#define RGC_GRID_T 256  // workgroup size of the grid build's per-point passes (k_count, k_place, k_rank_gather)
#endif
#ifndef RGC_VOX_T
#define RGC_VOX_T 256
#endif
constexpr int VOX_T = RGC_VOX_T;
#ifndef RGC_VOX_WAVES
#define RGC_VOX_WAVES 1  // waves per SIMD k_voxel_build_coop's register allocation must leave room for.  6 (rounds 3-5) held the launch to 80
                         // VGPRs -- and its cooperative half, the deferred queries everything behind the launch waits for, to 180 bytes of
                         // scratch spills; without the cap (128 VGPRs, no scratch) a frame is 8 us shorter
#endif
// LDS of a voxel-building workgroup: the nine terms of its points (rows one element longer than the block: the nine rows of one point
// fall into nine different bank pairs) and the list of the cells that START in the block.
struct VoxLds {
  double t[9][VOX_T + 1];
  int head_t[VOX_T + 1], head_c[VOX_T];  // per listed cell: its first point (index in the block), its cell
  int wcnt[VOX_T / WAVE];
};
// term k of sorted point u straight from memory (a cell that runs past its block's end): the expressions of the staging pass below
__device__ __forceinline__ double voxel_term(const float4* __restrict__ P, const double* __restrict__ nx, const double* __restrict__ ny,
                                             const double* __restrict__ nz, int k, int u) {
  if (k < 3) { const float4 p = P[u]; return (double)(k == 0 ? p.x : (k == 1 ? p.y : p.z)); }
  const double a = nx[u], b = ny[u], d = nz[u];
  switch (k) {
    case 3: return 1.0 - 0.999 * a * a;
    case 4: return -0.999 * a * b;
    case 5: return -0.999 * a * d;
    case 6: return 1.0 - 0.999 * b * b;
    case 7: return -0.999 * b * d;
    default: return 1.0 - 0.999 * d * d;
  }
}
// One lane per POINT stages the point's nine terms in LDS; then one lane per (cell that starts in the block, term) adds that term over the
// cell's points in ascending sorted position -- the cloud's order, the very additions a serial loop makes, so the same bits -- divides
// and stores its entry of the record.  (Until round 4 the lane of a cell's FIRST point added all nine terms and every lane looked its
// cell's bounds up: three dependent round trips per workgroup, now two.  The pass is bound by memory LATENCY at the parallelism its
// waves offer -- 64 % of its wave cycles wait on memory counters with 22 waves per CU in flight, 2 TB/s: scripts/pmc_kernel.sh -- not by
// the additions: the per-term lanes alone changed nothing, dropping the bounds look-up gave 36 -> 33 us.  Measured and dropped: workgroups
// that walk several tiles with the next tile's loads in flight during the sums (36.7 us: the barrier at the end of a tile waits for
// them anyway); dealing the terms to the cell's own lanes (round 4, earlier: it indexes the accumulators dynamically).)
__device__ __forceinline__ void voxel_build_block(const float4* __restrict__ P, const double* __restrict__ nx, const double* __restrict__ ny,
                                                  const double* __restrict__ nz, const int* __restrict__ start, const Grid& g, int n,
                                                  const int* __restrict__ cell_voxel, double* __restrict__ vox, int* __restrict__ vox_cell,
                                                  VoxLds& sh, int block) {
  const int b0 = block * VOX_T, bend = min(b0 + VOX_T, n);
  const int s = b0 + threadIdx.x;
  const int lane = threadIdx.x & (WAVE - 1), w = threadIdx.x / WAVE;
  int c = 0;
  bool head = false;
  if (s < n) {
    // a cell STARTS at this point iff the point before it (sorted order) lies in another cell: the neighbour's coordinates come with the
    // point's own in one round trip -- no look-up of the cell's bounds (a second, dependent round trip for every lane of the pass)
    const float4 cp = P[s], pv = P[s > 0 ? s - 1 : 0];
    const double a = nx[s], b = ny[s], d = nz[s];
    c = cell_index(g, cell_coord(cp.x, g) - g.minc[0], cell_coord(cp.y, g) - g.minc[1], cell_coord(cp.z, g) - g.minc[2]);
    const int cprev = cell_index(g, cell_coord(pv.x, g) - g.minc[0], cell_coord(pv.y, g) - g.minc[1], cell_coord(pv.z, g) - g.minc[2]);
    head = s == 0 || cprev != c;
    sh.t[0][threadIdx.x] = (double)cp.x;
    sh.t[1][threadIdx.x] = (double)cp.y;
    sh.t[2][threadIdx.x] = (double)cp.z;
    sh.t[3][threadIdx.x] = 1.0 - 0.999 * a * a;
    sh.t[4][threadIdx.x] = -0.999 * a * b;
    sh.t[5][threadIdx.x] = -0.999 * a * d;
    sh.t[6][threadIdx.x] = 1.0 - 0.999 * b * b;
    sh.t[7][threadIdx.x] = -0.999 * b * d;
    sh.t[8][threadIdx.x] = 1.0 - 0.999 * d * d;
  }
  const unsigned long long mh = __ballot(head);
  if (lane == 0) sh.wcnt[w] = __popcll(mh);
  __syncthreads();
  int base = 0, H = 0;
#pragma unroll
  for (int j = 0; j < VOX_T / WAVE; j++) { const int q = sh.wcnt[j]; base += j < w ? q : 0; H += q; }
  if (head) {
    const int h = base + __popcll(mh & ((1ull << lane) - 1ull));
    sh.head_t[h] = threadIdx.x; sh.head_c[h] = c;
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < 9 * H; idx += VOX_T) {
    const int h = idx / 9, k = idx - 9 * h;
    const int t0 = sh.head_t[h], cc = sh.head_c[h];
    const int vv = cell_voxel[cc];  // dense id from the cell scan: asked for now, needed behind the loop
    // the cell ends where the next one starts; the block's last cell may run past the block: its end is the one look-up of the cell table
    const bool last = h == H - 1;
    const int e_in = last ? bend - b0 : sh.head_t[h + 1];
    const int e1 = last ? start[cc + 1] : b0 + e_in;
    const double* __restrict__ row = sh.t[k];
    double acc = 0.0;
    for (int u = t0; u < e_in; u++) acc += row[u];
    for (int u = b0 + e_in; u < e1; u++) acc += voxel_term(P, nx, ny, nz, k, u);  // the cell runs past this block
    const double num = (double)(e1 - (b0 + t0));
    vox[(size_t)vv * kVoxRec + k] = acc / num;
    if (k == 0) {
      vox[(size_t)vv * kVoxRec + 9] = num;
      vox_cell[vv] = cc;
    }
  }
}
__global__ void __launch_bounds__(VOX_T)
k_voxel_build(const float4* __restrict__ P, const double* __restrict__ nx, const double* __restrict__ ny, const double* __restrict__ nz,
              const int* __restrict__ start, Grid g, int n, const int* __restrict__ cell_voxel, double* __restrict__ vox,
              int* __restrict__ vox_cell) {
  __shared__ VoxLds sh;
  voxel_build_block(P, nx, ny, nz, start, g, n, cell_voxel, vox, vox_cell, sh, (int)blockIdx.x);
}
// k_voxel_build and the map's cooperative kNN kernel in ONE launch: the FIRST nb_coop workgroups resolve the deferred queries -- four waves
// each, a wave per query; first, because workgroups are dispatched in order and a query is 20 us of latency -- the workgroups behind them
// build the voxel map (from the normals the bulk kernel wrote) at the same time.  The voxels that hold a deferred query are recomputed
// afterwards (k_voxel_patch).  Two streams and events did the same 20 us SLOWER than the serial chain (a cross-stream dependency costs
// ~10 us here); one launch has no such hop.
template <int KC>
__global__ void __launch_bounds__(VOX_T, RGC_VOX_WAVES)
k_voxel_build_coop(const float4* __restrict__ P, double* __restrict__ nx, double* __restrict__ ny, double* __restrict__ nz,
                   const int* __restrict__ start, Grid g, int n, const int* __restrict__ cell_voxel, double* __restrict__ vox,
                   int* __restrict__ vox_cell, int nb_coop, int k, Deferred df) {
  __shared__ VoxLds sh;
  __shared__ CoopRows shm[VOX_T / WAVE];
  if (df.guard && *df.guard) return;  // (a cloud parked on a grid it does not fit: its points' cells are not the sorted order's)
  if ((int)blockIdx.x >= nb_coop) {
    voxel_build_block(P, nx, ny, nz, start, g, n, cell_voxel, vox, vox_cell, sh, (int)blockIdx.x - nb_coop);
    return;
  }
  const int w = (int)threadIdx.x / WAVE;
  coop_run<KC, true>(P, start, g, k, df, nx, ny, nz, &shm[w], (int)threadIdx.x & (WAVE - 1), (int)blockIdx.x * (VOX_T / WAVE) + w,
                     nb_coop * (VOX_T / WAVE));
}

// The voxels of the DEFERRED queries once more.  The cooperative search (one wave per deferred query: ~100 of a million, 20 us of
// latency) runs beside the voxel map's build in the same launch (k_voxel_build_coop) instead of in front of it; the voxel records
// computed meanwhile from those points' stale normals are recomputed here, behind it -- the same sums in the same order (ascending
// sorted position = the cloud's order), so the table is bit for bit what the serial chain gave.  Several deferred queries of one
// voxel write the same values.
// One WAVE per deferred query's voxel: the lanes fetch the cell's points and normals in ONE round trip (a lane per point, 64 at a time)
// and lane 0 adds the nine terms in ascending position from LDS.  (A lane per voxel walking its ~11 points was eleven dependent round
// trips: 10 us of pure latency for ~100 voxels, on every frame's critical path.)
// the voxel record of cell c by ONE wave (term: its nine LDS rows): the cell's points and normals fetched a lane per point, the nine sums
// added by lane 0 in ascending sorted position = the cloud's order, like k_voxel_build -- the same bits
__device__ __forceinline__ void voxel_of_cell_wave(const float4* __restrict__ P, const double* __restrict__ nx, const double* __restrict__ ny,
                                                   const double* __restrict__ nz, const int* __restrict__ start, int c, const int* __restrict__ cell_voxel,
                                                   double* __restrict__ vox, int* __restrict__ vox_cell, double (*term)[WAVE], int lane) {
  const int s0 = start[c], s1 = start[c + 1];
  double m[3] = {0, 0, 0}, C[6] = {0, 0, 0, 0, 0, 0};
  for (int b0 = s0; b0 < s1; b0 += WAVE) {
    const int u = b0 + lane;
    if (u < s1) {
      const float4 p0 = P[u];
      const double a = nx[u], b = ny[u], d = nz[u];
      term[0][lane] = (double)p0.x; term[1][lane] = (double)p0.y; term[2][lane] = (double)p0.z;
      term[3][lane] = 1.0 - 0.999 * a * a; term[4][lane] = -0.999 * a * b; term[5][lane] = -0.999 * a * d;
      term[6][lane] = 1.0 - 0.999 * b * b; term[7][lane] = -0.999 * b * d; term[8][lane] = 1.0 - 0.999 * d * d;
    }
    wave_lds_fence();
    if (lane == 0) {
      const int nb = min(WAVE, s1 - b0);
      for (int t = 0; t < nb; t++) {
        m[0] += term[0][t]; m[1] += term[1][t]; m[2] += term[2][t];
        C[0] += term[3][t]; C[1] += term[4][t]; C[2] += term[5][t];
        C[3] += term[6][t]; C[4] += term[7][t]; C[5] += term[8][t];
      }
    }
    wave_lds_fence();
  }
  if (lane == 0) {
    const double num = (double)(s1 - s0);
    const int v = cell_voxel[c];
    if (vox_cell) vox_cell[v] = c;
    double* rec = vox + (size_t)v * kVoxRec;
    rec[0] = m[0] / num; rec[1] = m[1] / num; rec[2] = m[2] / num;
#pragma unroll
    for (int a = 0; a < 6; a++) rec[3 + a] = C[a] / num;
    rec[9] = num;
  }
}

__global__ void __launch_bounds__(WAVE)
k_voxel_patch(const float4* __restrict__ P, const double* __restrict__ nx, const double* __restrict__ ny, const double* __restrict__ nz,
              const int* __restrict__ start, Grid g, const int* __restrict__ deferred, const int* __restrict__ cell_voxel, double* __restrict__ vox) {
  __shared__ double term[9][WAVE];
  const int cnt = deferred[0];
  const int* idx = deferred + 16;
  const int lane = (int)threadIdx.x;
  for (int e = blockIdx.x; e < cnt; e += gridDim.x) {
    const int enc = idx[e];
    const int i = enc < 0 ? ~enc : enc;
    const float4 cp = P[i];
    const int c = cell_index(g, cell_coord(cp.x, g) - g.minc[0], cell_coord(cp.y, g) - g.minc[1], cell_coord(cp.z, g) - g.minc[2]);
    voxel_of_cell_wave(P, nx, ny, nz, start, c, cell_voxel, vox, nullptr, term, lane);
  }
}

// Lazy target: the voxel pass over the LISTED cells (k_footprint's cell_list: the first sorted point of each occupied cell the solve can
// look up), a wave per cell -- and, as in k_voxel_build_coop, the map's deferred queries resolved by the first nb_coop workgroups of the
// same launch (k_voxel_patch behind it redoes the voxels that hold one).
template <int KC>
__global__ void __launch_bounds__(VOX_T)
k_voxel_cells_coop(const float4* __restrict__ P, double* __restrict__ nx, double* __restrict__ ny, double* __restrict__ nz,
                   const int* __restrict__ start, Grid g, const int* __restrict__ cell_voxel, double* __restrict__ vox, int* __restrict__ vox_cell,
                   int nb_coop, int k, Deferred df, const int* __restrict__ cell_list, const int* __restrict__ ncells) {
  __shared__ double term[VOX_T / WAVE][9][WAVE];
  __shared__ CoopRows shm[VOX_T / WAVE];
  const int w = (int)threadIdx.x / WAVE, lane = (int)threadIdx.x & (WAVE - 1);
  if ((int)blockIdx.x < nb_coop) {
    if (df.guard && *df.guard) return;
    coop_run<KC, true>(P, start, g, k, df, nx, ny, nz, &shm[w], lane, (int)blockIdx.x * (VOX_T / WAVE) + w, nb_coop * (VOX_T / WAVE));
    return;
  }
  if (df.guard && *df.guard) return;  // (a point outside the speculative grid: nothing was listed, the cloud will be prepared again)
  const int nc = *ncells, nwaves = ((int)gridDim.x - nb_coop) * (VOX_T / WAVE);
  for (int e = ((int)blockIdx.x - nb_coop) * (VOX_T / WAVE) + w; e < nc; e += nwaves) {
    const float4 cp = P[cell_list[e]];
    const int c = cell_index(g, cell_coord(cp.x, g) - g.minc[0], cell_coord(cp.y, g) - g.minc[1], cell_coord(cp.z, g) - g.minc[2]);
    voxel_of_cell_wave(P, nx, ny, nz, start, c, cell_voxel, vox, vox_cell, term[w], lane);
  }
}

// ------------------------------------------------------------------------------------------------
// C4 + C5  correspondences, Mahalanobis, residual, Jacobian, normal equations.
// fast_vgicp_impl.hpp:73-116 (update_correspondences) + :119-180 (linearize); SURVEY A.4.
// One lane per source point; 28 fp64 partials per lane -> wave __shfl reduction -> LDS across the block's waves
// -> one partial row per block; a second single-block kernel folds the rows in a fixed order (deterministic).
// ------------------------------------------------------------------------------------------------
constexpr int LIN_T = 256;
int linearize_blocks(int n) { return (n + LIN_T - 1) / LIN_T; }

__device__ __forceinline__ bool inv_sym3(const double S[6], double M[6]) {
  const double a = S[0], b = S[1], c = S[2], d = S[3], e = S[4], f = S[5];
  const double c00 = d * f - e * e, c01 = c * e - b * f, c02 = b * e - c * d;
  const double det = a * c00 + b * c01 + c * c02;
  if (det == 0.0) return false;
  const double id = 1.0 / det;
  M[0] = c00 * id; M[1] = c01 * id; M[2] = c02 * id;
  M[3] = (a * f - c * c) * id; M[4] = (b * c - a * e) * id; M[5] = (a * d - b * b) * id;
  return true;
}

__device__ __forceinline__ void neighbor_offset(int noff, int o, int& ox, int& oy, int& oz) {
  // fast_vgicp_voxel.hpp:10-44
  if (noff == 1) { ox = oy = oz = 0; return; }
  if (noff == 7) {
    ox = (o == 1) - (o == 2);
    oy = (o == 3) - (o == 4);
    oz = (o == 5) - (o == 6);
    return;
  }
  ox = o / 9 - 1; oy = (o / 3) % 3 - 1; oz = o % 3 - 1;
}

// kWriteThrough: the row is handed to another workgroup of the SAME launch (last-block fold): agent-scope relaxed
// atomic stores lower to write-through (sc1) stores, so no release fence (L2 write-back) is needed before the ticket.
//
// Sum of NACC fp64 accumulators over the workgroup in a FIXED order: ((t0 + t1) + (t2 + t3)) inside every quad of lanes with
// two DPP quad permutes (registers only), the quad sums parked in LDS as [quad][accumulator], then thread (accumulator, eighth)
// adds its eight quads in ascending order and thread `accumulator` the eight eighths.  The former version reduced every
// accumulator across the wave with six dependent 64-bit shuffles (two ds_bpermute each): 7.6 us of the LM step's 20
// (scripts/lab_lm.py); this one moves 4 x fewer values through LDS once.
template <int NACC, bool kWriteThrough = false>
__device__ __forceinline__ void block_reduce_store(double (&acc)[NACC], double* __restrict__ row) {
  static_assert(NACC <= 32 && LIN_T == 256, "thread -> (accumulator, eighth) mapping");
  if constexpr (NACC <= 2) {
    __shared__ double red[LIN_T / WAVE][NACC];
    const int lane = threadIdx.x & (WAVE - 1), w = threadIdx.x / WAVE;
#pragma unroll
    for (int a = 0; a < NACC; a++) {
      double v = wave_sum(acc[a]);
      if (lane == 0) red[w][a] = v;
    }
    __syncthreads();
    if (threadIdx.x < NACC) {
      double s = 0;
#pragma unroll
      for (int j = 0; j < LIN_T / WAVE; j++) s += red[j][threadIdx.x];
      if (kWriteThrough) __hip_atomic_store(&row[threadIdx.x], s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else row[threadIdx.x] = s;
    }
  } else {
    constexpr int PITCH = NACC | 1;  // odd pitch: the quads of a wave spread over the banks
    __shared__ double quads[LIN_T / 4][PITCH];
    __shared__ double eighths[8][32];
    const int q = threadIdx.x >> 2;
#pragma unroll
    for (int a = 0; a < NACC; a++) {
      double v = acc[a];
      v += quad_perm_f64<0xB1>(v);  // lanes 0<->1, 2<->3
      v += quad_perm_f64<0x4E>(v);  // lanes 0,1<->2,3: lane 0 of the quad holds (t0 + t1) + (t2 + t3)
      if ((threadIdx.x & 3) == 0) quads[q][a] = v;
    }
    __syncthreads();
    const int a = threadIdx.x & 31, e = threadIdx.x >> 5;
    if (a < NACC) {
      double t = quads[e * 8][a];
#pragma unroll
      for (int j = 1; j < 8; j++) t += quads[e * 8 + j][a];
      eighths[e][a] = t;
    }
    __syncthreads();
    if (threadIdx.x < NACC) {
      double t = eighths[0][threadIdx.x];
#pragma unroll
      for (int j = 1; j < 8; j++) t += eighths[j][threadIdx.x];
      if (kWriteThrough) __hip_atomic_store(&row[threadIdx.x], t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else row[threadIdx.x] = t;
    }
  }
}

// (the body, with the point and its normal already in registers: k_lm_step fetches them before it takes the previous launch's decision)
__device__ __forceinline__ void linearize_point_pre(const float4 pp, const double a0, const double a1, const double a2, int i, int n, const Pose& T,
                                                    const Grid& g, const int* __restrict__ cell_voxel, const double* __restrict__ vox, int noff,
                                                    int* __restrict__ corr_v, double* __restrict__ corr_M, int want_H, double (&acc)[kAccum],
                                                    int& ncorr, int* __restrict__ miss = nullptr, const int* __restrict__ need = nullptr, int stamp = 0) {
    const double p0 = (double)pp.x, p1 = (double)pp.y, p2 = (double)pp.z;
    const double q0 = T.R[0] * p0 + T.R[1] * p1 + T.R[2] * p2 + T.t[0];
    const double q1 = T.R[3] * p0 + T.R[4] * p1 + T.R[5] * p2 + T.t[1];
    const double q2 = T.R[6] * p0 + T.R[7] * p1 + T.R[8] * p2 + T.t[2];
    // R C_A R^T = I - 0.999 (R n)(R n)^T
    const double r0 = T.R[0] * a0 + T.R[1] * a1 + T.R[2] * a2;
    const double r1 = T.R[3] * a0 + T.R[4] * a1 + T.R[5] * a2;
    const double r2 = T.R[6] * a0 + T.R[7] * a1 + T.R[8] * a2;
    const double CA[6] = {1.0 - 0.999 * r0 * r0, -0.999 * r0 * r1, -0.999 * r0 * r2,
                          1.0 - 0.999 * r1 * r1, -0.999 * r1 * r2, 1.0 - 0.999 * r2 * r2};
    const int cx = (int)floor(q0 / g.res - 0.5) - g.minc[0];
    const int cy = (int)floor(q1 / g.res - 0.5) - g.minc[1];
    const int cz = (int)floor(q2 / g.res - 0.5) - g.minc[2];
    for (int o = 0; o < noff; o++) {
      int ox, oy, oz;
      neighbor_offset(noff, o, ox, oy, oz);
      const int x = cx + ox, y = cy + oy, z = cz + oz;
      int v = -1;
      if (x >= 0 && x < g.dim[0] && y >= 0 && y < g.dim[1] && z >= 0 && z < g.dim[2]) v = cell_voxel[cell_index(g, x, y, z)];
      const size_t slot = (size_t)o * n + i;
      if (v >= 0 && miss && need[cell_index(g, x, y, z)] != stamp) {
        // lazy target: an occupied voxel the footprint did not cover (its cell does not carry this frame's stamp: nothing was built for
        // it).  The solve's result will be thrown away -- the caller completes the map and solves again -- so the correspondence is
        // simply dropped here (write-through flag: another XCD's workgroup reads it)
        __hip_atomic_store(miss, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        v = -1;
      }
      corr_v[slot] = v;
      if (v < 0) continue;
      const double* rec = vox + (size_t)v * kVoxRec;
      double S[6], M[6];
#pragma unroll
      for (int a = 0; a < 6; a++) S[a] = rec[3 + a] + CA[a];
      if (!inv_sym3(S, M)) {
#pragma unroll
        for (int a = 0; a < 6; a++) M[a] = 0.0;
      }
#pragma unroll
      for (int a = 0; a < 6; a++) corr_M[((size_t)a * noff + o) * n + i] = M[a];
      ncorr++;
      const double e0 = rec[0] - q0, e1 = rec[1] - q1, e2 = rec[2] - q2;
      const double w = sqrt(rec[9]);
      const double Me0 = M[0] * e0 + M[1] * e1 + M[2] * e2;
      const double Me1 = M[1] * e0 + M[3] * e1 + M[4] * e2;
      const double Me2 = M[2] * e0 + M[4] * e1 + M[5] * e2;
      acc[27] += w * (e0 * Me0 + e1 * Me1 + e2 * Me2);
      if (!want_H) continue;
      // J = [skew(q) | -I]  (3x6), columns [rot, trans]; H += w J^T M J, b += w J^T M e
      const double J[3][6] = {{0.0, -q2, q1, -1.0, 0.0, 0.0}, {q2, 0.0, -q0, 0.0, -1.0, 0.0}, {-q1, q0, 0.0, 0.0, 0.0, -1.0}};
      double MJ[3][6];
#pragma unroll
      for (int c = 0; c < 6; c++) {
        MJ[0][c] = M[0] * J[0][c] + M[1] * J[1][c] + M[2] * J[2][c];
        MJ[1][c] = M[1] * J[0][c] + M[3] * J[1][c] + M[4] * J[2][c];
        MJ[2][c] = M[2] * J[0][c] + M[4] * J[1][c] + M[5] * J[2][c];
      }
      int u = 0;
#pragma unroll
      for (int a = 0; a < 6; a++) {
#pragma unroll
        for (int c = a; c < 6; c++) {
          acc[u] += w * (J[0][a] * MJ[0][c] + J[1][a] * MJ[1][c] + J[2][a] * MJ[2][c]);
          u++;
        }
      }
#pragma unroll
      for (int a = 0; a < 6; a++) acc[21 + a] += w * (J[0][a] * Me0 + J[1][a] * Me1 + J[2][a] * Me2);
    }
}

// per-point work of FastVGICP::update_correspondences + linearize (fast_vgicp_impl.hpp:73-180) for sorted source point i
__device__ __forceinline__ void linearize_point(const float4* __restrict__ P, const double* __restrict__ nx, const double* __restrict__ ny,
                                                const double* __restrict__ nz, int i, int n, const Pose& T, const Grid& g,
                                                const int* __restrict__ cell_voxel, const double* __restrict__ vox, int noff,
                                                int* __restrict__ corr_v, double* __restrict__ corr_M, int want_H, double (&acc)[kAccum],
                                                int& ncorr, int* __restrict__ miss = nullptr, const int* __restrict__ need = nullptr, int stamp = 0) {
  linearize_point_pre(P[i], nx[i], ny[i], nz[i], i, n, T, g, cell_voxel, vox, noff, corr_v, corr_M, want_H, acc, ncorr, miss, need, stamp);
}

__device__ __forceinline__ void block_count_store(int ncorr, int* __restrict__ dst) {  // exact integer block sum -> *dst
  int c = ncorr;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o);
  __shared__ int cred[LIN_T / WAVE];
  if ((threadIdx.x & (WAVE - 1)) == 0) cred[threadIdx.x / WAVE] = c;
  __syncthreads();
  if (threadIdx.x == 0) {
    int s = 0;
    for (int j = 0; j < LIN_T / WAVE; j++) s += cred[j];
    *dst = s;
  }
}

__global__ void __launch_bounds__(LIN_T)
k_linearize(const float4* __restrict__ P,
            const double* __restrict__ nx, const double* __restrict__ ny, const double* __restrict__ nz, int n, Pose T, Grid g,
            const int* __restrict__ cell_voxel, const double* __restrict__ vox, int noff, int* __restrict__ corr_v,
            double* __restrict__ corr_M, int want_H, double* __restrict__ partials, int* __restrict__ ncorr_partials) {
  const int i = blockIdx.x * LIN_T + threadIdx.x;
  double acc[kAccum];
#pragma unroll
  for (int a = 0; a < kAccum; a++) acc[a] = 0.0;
  int ncorr = 0;
  if (i < n) linearize_point(P, nx, ny, nz, i, n, T, g, cell_voxel, vox, noff, corr_v, corr_M, want_H, acc, ncorr);
  block_reduce_store<kAccum>(acc, partials + (size_t)blockIdx.x * kAccum);
  block_count_store(ncorr, ncorr_partials + blockIdx.x);
}

// cost of sorted source point i with the correspondences and Mahalanobis matrices frozen by the last linearisation
__device__ __forceinline__ double error_point_pre(const float4 pp, int i, int n, const double* __restrict__ T12 /* row-major 3x4 */,
                                                  const double* __restrict__ vox, int noff, const int* __restrict__ corr_v,
                                                  const double* __restrict__ corr_M) {
  const double p0 = (double)pp.x, p1 = (double)pp.y, p2 = (double)pp.z;
  const double q0 = T12[0] * p0 + T12[1] * p1 + T12[2] * p2 + T12[3];
  const double q1 = T12[4] * p0 + T12[5] * p1 + T12[6] * p2 + T12[7];
  const double q2 = T12[8] * p0 + T12[9] * p1 + T12[10] * p2 + T12[11];
  double s = 0.0;
  for (int o = 0; o < noff; o++) {
    const int v = corr_v[(size_t)o * n + i];
    if (v < 0) continue;
    const double* rec = vox + (size_t)v * kVoxRec;
    double M[6];
#pragma unroll
    for (int a = 0; a < 6; a++) M[a] = corr_M[((size_t)a * noff + o) * n + i];
    const double e0 = rec[0] - q0, e1 = rec[1] - q1, e2 = rec[2] - q2;
    const double w = sqrt(rec[9]);
    s += w * (e0 * (M[0] * e0 + M[1] * e1 + M[2] * e2) + e1 * (M[1] * e0 + M[3] * e1 + M[4] * e2) + e2 * (M[2] * e0 + M[4] * e1 + M[5] * e2));
  }
  return s;
}
__device__ __forceinline__ double error_point(const float4* __restrict__ P, int i, int n, const double* __restrict__ T12, const double* __restrict__ vox,
                                              int noff, const int* __restrict__ corr_v, const double* __restrict__ corr_M) {
  return error_point_pre(P[i], i, n, T12, vox, noff, corr_v, corr_M);
}

// ---- device-chained LM: the whole LsqRegistration::computeTransformation loop (lsq_registration_impl.hpp:53-172) as a
// state machine in device memory, advanced by STEP kernels (k_lm_step below): every workgroup of a launch folds the rows the
// previous launch left and takes its decision; the host enqueues launches blindly and spins on the posted result.
// The one hand-off between workgroups of a launch that is left is the score's fold (and the separate score kernels'), through
// last_block_arrive (cdna_hip_programming.md G16): write-through (sc1) row stores, every wave drains vmcnt, workgroup barrier,
// one lane takes a ticket; the last arriver does an agent-scope ACQUIRE before its workgroup reads the rows.
__device__ __forceinline__ bool last_block_arrive(int* ticket) {
  __shared__ int is_last_s;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    // the rows were written with write-through (sc1) stores and every storing wave has drained vmcnt: no release fence
    const int t = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int last = (t == (int)gridDim.x - 1);
    if (last) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    is_last_s = last;
  }
  __syncthreads();
  return is_last_s != 0;
}

// Fold of the per-workgroup rows, all LIN_T threads at once: thread t owns accumulator t % 32 and the rows t / 32, t / 32 + 8, ... (summed
// in ascending order); the eight strided sums of an accumulator are then added in ascending order from LDS.  Fixed order -> deterministic
// for a given row count, and the same in every workgroup that folds the same rows.  The thread's first sixteen rows (t / 32 + 8 u, u < 16)
// are fetched by block_fold_rows_load16 -- k_lm_step issues those loads together with its other loads at the top of the launch: one round
// trip instead of one for the state and two for the rows -- any further ones eight at a time (a missing row is +0.0: no effect on a sum
// that starts at +0.0).
template <int NACC>
__device__ __forceinline__ void block_fold_rows_load16(const double* __restrict__ partials, int nrows, double (&v)[16]) {
  const int a = threadIdx.x & 31, gq = threadIdx.x >> 5;
#pragma unroll
  for (int u = 0; u < 16; u++) v[u] = (a < NACC && gq + 8 * u < nrows) ? partials[(size_t)(gq + 8 * u) * NACC + a] : 0.0;
}
template <int NACC>
__device__ __forceinline__ void block_fold_rows_pre(const double* __restrict__ partials, int nrows, const double (&v)[16], double* sh_out) {
  static_assert(NACC <= 32 && LIN_T == 256, "thread -> (accumulator, row group) mapping");
  __shared__ double grp[LIN_T / 32][32];
  const int a = threadIdx.x & 31, gq = threadIdx.x >> 5;
  double s = 0;
  if (a < NACC) {
#pragma unroll
    for (int u = 0; u < 16; u++) s += v[u];
    for (int r0 = gq + 128; r0 < nrows; r0 += 64) {
      double w[8];
#pragma unroll
      for (int u = 0; u < 8; u++) w[u] = (r0 + 8 * u < nrows) ? partials[(size_t)(r0 + 8 * u) * NACC + a] : 0.0;
#pragma unroll
      for (int u = 0; u < 8; u++) s += w[u];
    }
  }
  grp[gq][a] = s;
  __syncthreads();
  if (threadIdx.x < NACC) {
    double t = grp[0][threadIdx.x];
#pragma unroll
    for (int j = 1; j < LIN_T / 32; j++) t += grp[j][threadIdx.x];
    sh_out[threadIdx.x] = t;
  }
  __syncthreads();
}

__device__ __forceinline__ bool lm_is_converged(const double* d, double rot_eps, double trans_eps) {  // :82-91
  double m = 0;
#pragma unroll
  for (int a = 0; a < 3; a++) {
#pragma unroll
    for (int e = 0; e < 3; e++) m = fmax(m, fabs(d[a * 4 + e] - (a == e ? 1.0 : 0.0)) / rot_eps);
    m = fmax(m, fabs(d[a * 4 + 3]) / trans_eps);
  }
  return m < 1;
}

// The modes of a launch of the device-chained LM (k_lm_step below):
constexpr int LM_MODE_LIN = 0, LM_MODE_BA = 1, LM_MODE_B = 2;
constexpr int kStepAcc = kAccum + 2;  // 28 linearisation sums, the correspondence count, the trial cost

__device__ __forceinline__ void lm_load_pose(const double* m16, Pose& T) {
#pragma unroll
  for (int a = 0; a < 3; a++) {
#pragma unroll
    for (int e = 0; e < 3; e++) T.R[a * 3 + e] = m16[a * 4 + e];
    T.t[a] = m16[a * 4 + 3];
  }
}

#ifdef RGC_LAB
__device__ unsigned long long g_lab_ts[16];  // developer build: phase timestamps (100 MHz) of the last active k_lm_step
#define LAB_TS(k) do { if (threadIdx.x == 0) g_lab_ts[(first ? 0 : 8) + k] = wall_clock64(); } while (0)
#define LAB_TS_MIN(k) do { if (threadIdx.x == 0) atomicMin(&g_lab_ts[(first ? 0 : 8) + k], (unsigned long long)wall_clock64()); } while (0)
void lab_lm_ts(unsigned long long* out, hipStream_t s) {
  (void)hipStreamSynchronize(s);
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_lab_ts), sizeof(g_lab_ts));
  unsigned long long init[16];
  for (int i = 0; i < 16; i++) init[i] = ~0ull;
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_lab_ts), init, sizeof(init));
}
#else
#define LAB_TS(k)
#define LAB_TS_MIN(k)
#endif

// The frame's small counters have been captured into the LM state: put both clouds' blocks of d_small back to their initial image
// (bounding-box accumulators, guard, the scan's sum of count^2), so that the NEXT preparation of either cloud needs no 32-byte
// H2D copy in front of its first kernel (3.5 us of copy kernel + 4 us of gap at the head of each preparation chain).  The map's voxel
// count (nvox[0]) and the correspondence count behind it stay: a resident target is solved against again.
__device__ __forceinline__ void reinit_small_blocks(const int* nvox) {
  int* t = const_cast<int*>(nvox) - 7;  // d_small: [0..5] map bbox, [6] map guard, [7] nvox; [16..21] scan bbox, [22] scan guard, [23] scan sum count^2
#pragma unroll
  for (int b = 0; b < 2; b++) {
    int* m = t + 16 * b;
    m[0] = m[1] = m[2] = INT_MAX;
    m[3] = m[4] = m[5] = INT_MIN;
    m[6] = 0;
  }
  t[23] = 0;
}

// The fresh state of a solve (:53-63) plus the frame's counters, on a zeroed image (k_lm_step's opening launch, lane 0 of workgroup 0).
__device__ __forceinline__ void lm_state_open(LmState& ls, const LmInit& in, const int* __restrict__ nvox, const int* __restrict__ def_t,
                                              const int* __restrict__ def_s) {
#pragma unroll
  for (int a = 0; a < 16; a++) ls.x0[a] = in.x0[a];
  ls.lambda = -1.0;  // :56
  ls.nu = 2.0;
#pragma unroll
  for (int a = 0; a < 6; a++) ls.Hfin[a * 7] = 1.0;  // final_hessian_.setIdentity(), :21
  ls.rot_eps = in.rot_eps;
  ls.trans_eps = in.trans_eps;
  ls.init_factor = in.init_factor;
  ls.max_outer = in.max_outer;
  ls.max_inner = in.max_inner;
  ls.nvox = nvox ? *nvox : 0;
  ls.pad = nvox ? (nvox[-1] | (nvox[15] << 8)) : 0;  // grid guards of map and scan (d_small[6], [22])
  ls.def_t = def_t ? *def_t : 0;
  ls.def_s = def_s ? *def_s : 0;
  ls.src_sq = nvox ? __int_as_float(nvox[16]) : 0.f;  // d_small[23]
  if (nvox) reinit_small_blocks(nvox);
}

// The decision the sums of one launch call for, taken by lane 0 of a workgroup on its LDS copy `ls` of the state the launch ran from:
// what the folded sums mean in that launch's mode, accept / reject / terminate, and the next LM try (lsq_registration_impl.hpp:125-172).
__device__ __forceinline__ void lm_step_decide(LmState& ls, const double* folded, int mode, int cur, bool* took_xi) {
  const int first = 0;  // (the developer build's timestamps: slot of a launch that decides)
  (void)first;
  double H[36], b[6], x0[16], d[6], delta[16], xi[16];
  double lambda = ls.lambda;
  bool have_lin = false;  // H, b (registers) hold a linearisation at the pose the next try starts from
  auto adopt = [&]() {    // H, b, y0, ncorr of the linearisation just folded
    int u = 0;
#pragma unroll
    for (int a = 0; a < 6; a++)
#pragma unroll
      for (int e = a; e < 6; e++) { H[a * 6 + e] = folded[u]; H[e * 6 + a] = folded[u]; u++; }
#pragma unroll
    for (int a = 0; a < 6; a++) b[a] = folded[21 + a];
#pragma unroll
    for (int a = 0; a < 36; a++) ls.H[a] = H[a];
#pragma unroll
    for (int a = 0; a < 6; a++) ls.b[a] = b[a];
    ls.y0 = folded[27];
    ls.ncorr = (int)folded[kAccum];
    ls.n_lin++;
    have_lin = true;
  };
  if (mode == LM_MODE_LIN) {
    adopt();
    if (lambda < 0.0) {  // :130-132
      double m = 0;
#pragma unroll
      for (int a = 0; a < 6; a++) m = fmax(m, fabs(H[a * 7]));
      lambda = ls.init_factor * m;
    }
#pragma unroll
    for (int a = 0; a < 16; a++) x0[a] = ls.x0[a];
    ls.mode = LM_MODE_BA;
  } else {
    const double yi = folded[kAccum + 1];
    ls.yi = yi;
    ls.n_err++;
    double den = 0;
#pragma unroll
    for (int a = 0; a < 6; a++) den += ls.d[a] * (lambda * ls.d[a] - ls.b[a]);
    const double rho = (ls.y0 - yi) / den;  // :145
    const bool conv_now = lm_is_converged(ls.delta, ls.rot_eps, ls.trans_eps);
    bool outer_done = false;
    if (rho < 0) {  // :155-163
      if (conv_now) {
        outer_done = true;  // step_lm returns true with x unchanged
      } else {
        lambda = ls.nu * lambda;
        ls.nu = 2 * ls.nu;
        ls.inner++;
        if (ls.inner >= ls.max_inner) { ls.failed = 1; ls.done = 1; ls.lambda = lambda; return; }  // "lm not converged!!", :69-72
#pragma unroll
        for (int a = 0; a < 36; a++) H[a] = ls.H[a];
#pragma unroll
        for (int a = 0; a < 6; a++) b[a] = ls.b[a];
#pragma unroll
        for (int a = 0; a < 16; a++) x0[a] = ls.x0[a];
        have_lin = true;  // the same linearisation, a larger lambda
        ls.mode = LM_MODE_B;
      }
    } else {  // :165-168
      *took_xi = true;
#pragma unroll
      for (int a = 0; a < 16; a++) { x0[a] = ls.xi[a]; ls.x0[a] = x0[a]; }
      const double r21 = 2 * rho - 1;
      lambda = lambda * fmax(1.0 / 3.0, 1 - r21 * r21 * r21);
#pragma unroll
      for (int a = 0; a < 36; a++) ls.Hfin[a] = ls.H[a];
      outer_done = true;
    }
    if (outer_done) {
      ls.conv = conv_now ? 1 : 0;  // :74
      ls.outer++;
      ls.nu = 2.0;
      ls.inner = 0;
      if (conv_now || ls.outer >= ls.max_outer) {  // :65
        ls.done = 1;
        ls.lambda = lambda;
        return;
      }
      // accepted and not finished (rho < 0 never gets here: it only ends an outer iteration when converged)
      if (mode == LM_MODE_BA) {
        adopt();          // the speculative linearisation was taken at xi = the new x0
        ls.cur = cur ^ 1;
      } else {
        ls.mode = LM_MODE_LIN;  // retry path: the next launch linearises at the new x0
      }
    }
  }
  ls.lambda = lambda;
  if (!have_lin) return;
  LAB_TS(5);
  rgclm::lm_try(H, b, lambda, x0, d, delta, xi);  // :136-143
  LAB_TS(6);
#pragma unroll
  for (int a = 0; a < 6; a++) ls.d[a] = d[a];
#pragma unroll
  for (int a = 0; a < 16; a++) { ls.delta[a] = delta[a]; ls.xi[a] = xi[a]; }
  LAB_TS(7);
}

// The finished solve's state goes straight into MAPPED HOST memory, followed by the solve's sequence number in `gen`: the host thread
// spins on that word instead of waiting for the stream to drain (the blind launches enqueued behind the deciding one, the copy of
// the state, the wake-up of a blocked hipStreamSynchronize: ~20 us at the end of every frame of a dependent sequence).
// src: the state (LDS or device memory); every thread of the workgroup takes part; skip_fit: the fitness words are written by the caller.
__device__ __forceinline__ void post_state_to_host(const LmState* src, LmState* __restrict__ h_post, int seq, int nthreads, bool skip_fit) {
  constexpr int kWords = (int)(sizeof(LmState) / sizeof(int));
  constexpr int kGen = (int)(offsetof(LmState, gen) / sizeof(int));
  constexpr int kFit0 = (int)(offsetof(LmState, fit_sum) / sizeof(int)), kHas = (int)(offsetof(LmState, has_fit) / sizeof(int));
  const int* sw = reinterpret_cast<const int*>(src);
  int* hw = reinterpret_cast<int*>(h_post);
  for (int u = threadIdx.x; u < kWords; u += nthreads) {
    if (u == kGen || (skip_fit && (u == kFit0 || u == kFit0 + 1 || u == kHas))) continue;
    __hip_atomic_store(&hw[u], sw[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");  // system scope: this thread's words are on their way to the host before the barrier
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(&hw[kGen], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// (defined with the score's kernels below)
constexpr int FIT_T = 64;
int fitness_blocks(int n) { return (n + FIT_T - 1) / FIT_T; }
__device__ __forceinline__ int fitness_blocks_dev(int n) { return (n + FIT_T - 1) / FIT_T; }
__device__ __forceinline__ double fitness_wave(const float4* __restrict__ SP, int ns, const PoseF& T, const float4* __restrict__ TP,
                                               const int* __restrict__ tstart, const Grid& g, int n_all, int w, int W);
__device__ __forceinline__ double fitness_fold(const double* __restrict__ partials, int W);

// What the score needs beside the solve's own arguments (k_lm_step): the map's sorted points and cell starts, the rows of the per-wave
// sums, and whether a small map is scanned whole (fitness_wave).  on == 0: no score is chained to this solve.
struct FitArgs {
  const float4* TP; const int* tstart; double* partials; int n_all; int on;
  const int* need; int stamp;  // lazy target: the target is built only for the cells stamped `stamp` in need[]: every look-up is checked
  const int* counts;           // ... and its list sizes ([0] queries, [1] cells) ride home with the state
  LmEarly* early;              // mapped host memory (nullable): the final pose goes there before the score is computed
};

// this wave's share of getFitnessScore at pose m16 (cast to float like final_transformation_, :77), as a write-through row
__device__ __forceinline__ void step_fitness_rows(const float4* __restrict__ SP, int n, const double* m16, const Grid& g, const FitArgs& fa) {
  const int W = (n + FIT_T - 1) / FIT_T, w = (int)blockIdx.x * (LIN_T / WAVE) + ((int)threadIdx.x >> 6);
  if (w >= W) return;
  PoseF T;
#pragma unroll
  for (int a = 0; a < 12; a++) T.m[a] = (float)m16[a];
  const double v = fitness_wave(SP, n, T, fa.TP, fa.tstart, g, fa.n_all, w, W);
  if (((int)threadIdx.x & (WAVE - 1)) == 0) __hip_atomic_store(&fa.partials[w], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// The words of the LM area that are NOT part of either state image (rgc_api.hip allocates 4096 bytes and zeroes them once):
//   +3072  the lazy target's miss flag ("a look-up hit an occupied voxel outside the part that was built"), set by any workgroup of any
//          launch of a solve, taken into the finished state (pad2) and cleared by whoever finishes it
//   +3076  the ticket of the score's fold (the one launch of a solve whose workgroups hand rows to each other)
static_assert(2 * sizeof(LmState) <= 3072, "two state images in front of the area's loose words");
__device__ __forceinline__ int* lm_area_miss(LmState* st) { return reinterpret_cast<int*>(reinterpret_cast<char*>(st) + 3072); }
__device__ __forceinline__ int* lm_area_ticket(LmState* st) { return reinterpret_cast<int*>(reinterpret_cast<char*>(st) + 3076); }

// One launch of the device-chained LM, launch number j of its solve (the host counts; 0 opens the solve).
//
// EVERY workgroup takes the decision the previous launch's sums call for, by itself: it reads the state image the previous launch
// left (st[(j-1) & 1]) and that launch's rows, folds them in the fixed order, and its lane 0 runs accept / reject / terminate and the
// next LM try (lm_step_decide) on an LDS copy -- the same inputs and the same code in every workgroup, hence the same state.  Then it
// does this launch's per-point work at the pose that decision produced and stores its row; workgroup 0 also stores the state image
// (st[j & 1]: never the one the launch reads).  No workgroup waits for another inside a step: the kernel boundary is the only
// synchronisation (the former step had both -- a ticket, a last arriver that fetched everyone's rows, decided and tried alone while
// the launch's other 117 workgroups had left: ~2.8 us of a 14 us step).
//
// What a launch does at the pose it arrives at (ls.mode, after the decision):
//   LM_MODE_LIN  linearise at x0 (correspondences -> buffer cur);                                                          next: BA
//   LM_MODE_BA   cost at the trial pose xi over the FROZEN correspondences of buffer cur (compute_error, :144) AND, speculatively, the
//                next linearisation AT xi into buffer cur^1.  The next launch's decision: rho >= 0 (the usual case) accepts x0 = xi
//                and -- unless the solve is over -- adopts the speculative H, b, y0 and correspondences (cur ^= 1) and tries again at
//                once: ONE launch per outer iteration.  rho < 0 discards the speculation, raises lambda, retries with the old H, b;   next: B
//   LM_MODE_B    cost only (a retry of the same linearisation); accept -> next: LIN, reject -> B again.
// A try whose delta is already below the convergence thresholds ends the solve whatever its cost turns out to be: its launch skips the
// speculative linearisation.  The arithmetic of each adopted linearisation / cost evaluation is exactly that of the two-kernel slots; the
// speculative linearisation at the final pose is never adopted, so buffer cur holds what the reference's last linearize() left.
//
// The launch whose decision ENDS the solve computes getFitnessScore at the final pose (fa.on; the same grid covers the scan: four
// waves per workgroup, one row per wave), its last-arriving workgroup folds the rows into the state and posts it to the host.  A
// launch on a finished solve only hands the state image on (workgroup 0: st[(j-1) & 1] -> st[j & 1]) so that the host finds the
// latest image behind its last launch whatever their number.
__global__ void __launch_bounds__(LIN_T)
k_lm_step(const float4* __restrict__ P, const double* __restrict__ nx, const double* __restrict__ ny, const double* __restrict__ nz, int n, Grid g,
          const int* __restrict__ cell_voxel, const double* __restrict__ vox, int noff, int* __restrict__ corr_v0, double* __restrict__ corr_M0,
          int* __restrict__ corr_v1, double* __restrict__ corr_M1, double* __restrict__ partials, LmState* __restrict__ st, int j,
          LmInit in, const int* __restrict__ nvox, const int* __restrict__ def_t, const int* __restrict__ def_s, LmState* __restrict__ h_post,
          int seq, FitArgs fa) {
  wave_prio(2);  // a latency chain: issue ahead of whatever shares the CU (the other context's kNN, the next scan's preparation)
#ifdef RGC_LAB_TURN
  const unsigned long long lab_turn_t0 = wall_clock64();
#endif
  constexpr int kStateWords = (int)(sizeof(LmState) / sizeof(int));
  __shared__ LmState ls;
  __shared__ double folded[kStepAcc];
  const int first = j == 0;
  (void)first;
  LmState* const sn = st + (j & 1);                                  // the image this launch leaves
  double* const rows_out = partials + (size_t)(j & 1) * gridDim.x * kStepAcc;
  int* const miss = lm_area_miss(st);
  auto store_image = [&](LmState* dst) {
    const int* lw = reinterpret_cast<const int*>(&ls);
    int* gw = reinterpret_cast<int*>(dst);
    for (int u = threadIdx.x; u < kStateWords; u += LIN_T) gw[u] = lw[u];
  };
  LAB_TS_MIN(0);
  // Everything this launch will need from memory that does not depend on the decision is asked for NOW, in one round trip: the state image,
  // this thread's rows of the previous launch, its scan point and normal.
  const int i = blockIdx.x * LIN_T + threadIdx.x;
  float4 pp = make_float4(0.f, 0.f, 0.f, 0.f);
  double pn0 = 0.0, pn1 = 0.0, pn2 = 0.0;
  if (i < n) { pp = P[i]; pn0 = nx[i]; pn1 = ny[i]; pn2 = nz[i]; }
  if (j == 0) {
    // The opening launch: nobody reads the (stale) images; pose and thresholds come from the kernel arguments.  Workgroup 0 builds the
    // fresh state (:53-63) with the frame's counters, so that ONE read-back at the end carries every statistic.
    if (blockIdx.x == 0) {
      int* lw = reinterpret_cast<int*>(&ls);
      for (int u = threadIdx.x; u < kStateWords; u += LIN_T) lw[u] = 0;
      __syncthreads();
      if (threadIdx.x == 0) {
        lm_state_open(ls, in, nvox, def_t, def_s);
        ls.lazy_nq = fa.counts ? fa.counts[0] : 0;
        ls.lazy_ncell = fa.counts ? fa.counts[1] : 0;
        ls.mode = LM_MODE_LIN;
        if (in.max_outer <= 0) ls.done = 1;  // max_iterations <= 0: the guess is the answer (a later launch scores it if asked to)
      }
      __syncthreads();
      store_image(sn);
      if (in.max_outer <= 0 && h_post && seq > 0) post_state_to_host(&ls, h_post, seq, LIN_T, false);
    }
    if (in.max_outer <= 0) return;
  } else {
    const double* const rows_in = partials + (size_t)((j - 1) & 1) * gridDim.x * kStepAcc;
    double rv[16];
    {
      static_assert(kStateWords <= 2 * LIN_T, "two words of the image per thread");
      int* lw = reinterpret_cast<int*>(&ls);
      const int* gw = reinterpret_cast<const int*>(st + ((j - 1) & 1));
      const int u0 = threadIdx.x, u1 = threadIdx.x + LIN_T;
      const int w0 = gw[u0], w1 = u1 < kStateWords ? gw[u1] : 0;
      block_fold_rows_load16<kStepAcc>(rows_in, gridDim.x, rv);  // (rows of a finished solve are stale but valid memory; they are not used then)
      lw[u0] = w0;
      if (u1 < kStateWords) lw[u1] = w1;
    }
    __syncthreads();
    const bool was_done = ls.done != 0;
    if (!was_done) {
      block_fold_rows_pre<kStepAcc>(rows_in, gridDim.x, rv, folded);
      LAB_TS(4);
      if (threadIdx.x == 0) {
        bool took_xi = false;
        lm_step_decide(ls, folded, ls.mode, ls.cur, &took_xi);
      }
      __syncthreads();
    }
    if (ls.done) {
      const bool score = fa.on && !ls.has_fit;
      if (score) {
        // the solve ended with this launch's decision (or earlier, unscored: max_iterations <= 0): the score at the final pose, every
        // wave its row; the last arriver folds them
        if (fa.early && blockIdx.x == 0 && !was_done) {
          // ... the final POSE first (every linearisation of this solve ran in an earlier launch: the guards and the lazy target's miss flag
          // are final too): whoever needs only the pose to go on does not wait for the score
          int* hw = reinterpret_cast<int*>(fa.early);
          const int* sw = reinterpret_cast<const int*>(ls.x0);
          if (threadIdx.x < 32) __hip_atomic_store(&hw[threadIdx.x], sw[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          if (threadIdx.x == 32) __hip_atomic_store(&fa.early->pad, ls.pad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          if (threadIdx.x == 33) __hip_atomic_store(&fa.early->pad2, __hip_atomic_load(miss, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          if (threadIdx.x == 34) __hip_atomic_store(&fa.early->outer, ls.outer, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
          __syncthreads();
          if (threadIdx.x == 0) __hip_atomic_store(&fa.early->gen, seq < 0 ? -seq : seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
#ifdef RGC_LAB_TURN
          if (threadIdx.x == 0) { const unsigned long long now = wall_clock64(); atomicExch(&g_lab_turn[3], now); atomicAdd(&g_lab_turn[5], now - lab_turn_t0); }
#endif
        }
        step_fitness_rows(P, n, ls.x0, g, fa);
        if (!last_block_arrive(lm_area_ticket(st))) return;
        if (threadIdx.x < WAVE) {
          const double t = fitness_fold(fa.partials, fitness_blocks_dev(n));
          if (threadIdx.x == 0) { ls.fit_sum = t; ls.has_fit = 1; }
        }
      } else if (blockIdx.x != 0) {
        return;
      }
      if (threadIdx.x == 0 && !was_done) {  // the lazy target's miss flag rides home with the state and is left cleared for the next solve
        ls.pad2 = __hip_atomic_load(miss, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(miss, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      __syncthreads();
      store_image(sn);
      // seq > 0: no score is chained to this solve -- a finished state is the frame's result; seq < 0: a finished state WITH its score is
      // (from here, or from k_fitness_lm when the caller keeps the score apart)
      if (h_post && (!was_done || score) && (seq > 0 || ls.has_fit)) post_state_to_host(&ls, h_post, seq < 0 ? -seq : seq, LIN_T, false);
      return;
    }
  }
  // ---- this launch's per-point work, at the pose the decision above arrived at ----
  const int mode = j == 0 ? LM_MODE_LIN : ls.mode, cur = j == 0 ? 0 : ls.cur;
  const bool final_try = mode != LM_MODE_LIN && lm_is_converged(ls.delta, ls.rot_eps, ls.trans_eps);
  int* cv_cur = cur ? corr_v1 : corr_v0;
  double* cm_cur = cur ? corr_M1 : corr_M0;
  int* cv_nxt = cur ? corr_v0 : corr_v1;
  double* cm_nxt = cur ? corr_M0 : corr_M1;
  double acc[kStepAcc];
#pragma unroll
  for (int a = 0; a < kStepAcc; a++) acc[a] = 0.0;
  if (mode != LM_MODE_LIN && i < n) acc[kAccum + 1] = error_point_pre(pp, i, n, ls.xi, vox, noff, cv_cur, cm_cur);
  if (mode != LM_MODE_B && !final_try) {
    Pose T;
    if (j == 0) lm_load_pose(in.x0, T);
    else lm_load_pose(mode == LM_MODE_LIN ? ls.x0 : ls.xi, T);
    double lin[kAccum];
#pragma unroll
    for (int a = 0; a < kAccum; a++) lin[a] = 0.0;
    int ncorr = 0;
    if (i < n) linearize_point_pre(pp, pn0, pn1, pn2, i, n, T, g, cell_voxel, vox, noff, mode == LM_MODE_LIN ? cv_cur : cv_nxt,
                                   mode == LM_MODE_LIN ? cm_cur : cm_nxt, 1, lin, ncorr, fa.need ? miss : nullptr, fa.need, fa.stamp);
#pragma unroll
    for (int a = 0; a < kAccum; a++) acc[a] = lin[a];
    acc[kAccum] = (double)ncorr;  // exact: counts are far below 2^53
  }
  LAB_TS_MIN(1);
  block_reduce_store<kStepAcc>(acc, rows_out + (size_t)blockIdx.x * kStepAcc);  // (read by the NEXT launch: plain stores)
  LAB_TS_MIN(2);
  if (j > 0 && blockIdx.x == 0) store_image(sn);
}

// fold per-block rows in a fixed order: block a (one wave) owns accumulator a; lane l sums rows l, l+64, ...
// then a fixed shuffle tree (deterministic for a given row count).  Block NACC folds the integer counts.
template <int NACC>
__global__ void __launch_bounds__(WAVE) k_fold(const double* __restrict__ partials, int nrows, double* __restrict__ out,
                                                const int* __restrict__ ipartials, int* __restrict__ iout) {
  const int a = blockIdx.x, lane = threadIdx.x;
  if (a < NACC) {
    double s = 0;
    for (int r = lane; r < nrows; r += WAVE) s += partials[(size_t)r * NACC + a];
    s = wave_sum(s);
    if (lane == 0) out[a] = s;
  } else if (ipartials) {
    int c = 0;
    for (int r = lane; r < nrows; r += WAVE) c += ipartials[r];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o);
    if (lane == 0) *iout = c;
  }
}

// C6  fast_vgicp_impl.hpp:183-204: frozen correspondences and Mahalanobis
__global__ void __launch_bounds__(LIN_T)
k_error(const float4* __restrict__ P, int n, Pose T, const double* __restrict__ vox, int noff, const int* __restrict__ corr_v,
        const double* __restrict__ corr_M, double* __restrict__ partials) {
  const int i = blockIdx.x * LIN_T + threadIdx.x;
  double acc[1] = {0.0};
  const double T12[12] = {T.R[0], T.R[1], T.R[2], T.t[0], T.R[3], T.R[4], T.R[5], T.t[1], T.R[6], T.R[7], T.R[8], T.t[2]};
  if (i < n) acc[0] = error_point(P, i, n, T12, vox, noff, corr_v, corr_M);
  block_reduce_store<1>(acc, partials + blockIdx.x);
}

// ------------------------------------------------------------------------------------------------
// C8  pcl::Registration::getFitnessScore: fp32 transform, exact 1-NN in the target grid, fp32 distances summed
// in fp64.  One lane per source point.
// ------------------------------------------------------------------------------------------------
// (FIT_T = 64, fitness_blocks: above k_lm_step -- one wave per row of the score's partial sums)

// Exact nearest neighbour of (px,py,pz) in the sorted target: own cell first (an aligned point's nearest map point is usually
// closer than its cell walls), else a cube of cells that grows until the best distance is provably inside it -- or until the
// unscanned region is farther than cap_r (then nothing within cap_r is missing).  Ties: smaller original index.
// best = squared distance (INFINITY if none found), bs = position in the sorted array (-1 if none; only tracked if kIndex:
// the fitness score needs the distance alone, and the index bookkeeping costs it a third of its time).
// unresolved (nullable): instead of growing the cube beyond its first size the search gives up and sets the flag -- the caller has a
// cheaper way for what is left (a small map: the whole wave scans all of it, fitness_wave)
template <bool kIndex>
__device__ __forceinline__ void nn_search(float px, float py, float pz, const float4* __restrict__ TP, const int* __restrict__ tstart,
                                          const Grid& g, double cap_r, float& best, int& bs, bool* unresolved = nullptr) {
  const int c[3] = {cell_coord(px, g) - g.minc[0], cell_coord(py, g) - g.minc[1], cell_coord(pz, g) - g.minc[2]};
  const double q[3] = {(double)px, (double)py, (double)pz};
  int rmax = 0, r = 1;
  bool inside = true;
#pragma unroll
  for (int a = 0; a < 3; a++) {
    rmax = max(rmax, max(c[a], g.dim[a] - 1 - c[a]));
    r = max(r, max(-c[a], c[a] - (g.dim[a] - 1)));  // first cube that touches the grid when the query lies outside
    inside = inside && c[a] >= 0 && c[a] < g.dim[a];
  }
  best = INFINITY;
  bs = -1;
  int bo = INT_MAX;
  auto take = [&](const float4& cp, int s) {
    const float d = dist2(px, py, pz, cp);
    if (kIndex) {
      const int o = __float_as_int(cp.w);
      if (d < best || (d == best && o < bo)) { best = d; bs = s; bo = o; }
    } else {
      best = fminf(best, d);
    }
  };
  auto scan = [&](int s0, int s1) {
    int s = s0;
    unsigned off = (unsigned)s0 << 4;
    for (; s + 8 <= s1; s += 8, off += 128) {  // eight loads in flight: this search is a chain of memory round trips
      float4 cc[8];
#pragma unroll
      for (int u = 0; u < 8; u++) cc[u] = point_at(TP, off + 16u * u);
#pragma unroll
      for (int u = 0; u < 8; u++) take(cc[u], s + u);
    }
    if (s < s1) {  // 1..7 left: clamped loads (a repeated candidate changes neither the minimum nor its index)
      const int last = s1 - 1;
      float4 cc[7];
#pragma unroll
      for (int u = 0; u < 7; u++) cc[u] = point_at(TP, (unsigned)min(s + u, last) << 4);
#pragma unroll
      for (int u = 0; u < 7; u++) take(cc[u], min(s + u, last));
    }
  };
  if (inside) {
    const int own = cell_index(g, c[0], c[1], c[2]);
    const int o0 = tstart[own], o1 = tstart[own + 1];
    scan(o0, o1);
    if (best < INFINITY) {
      const double bound = cube_bound(g, c, q, 0);
      if (bound == 1.0e300 || (bound > 0.0 && (double)best < bound * bound * (1.0 - 1e-5))) return;
      // The nearest point of the own cell bounds the search ball: of the 26 neighbouring cells only those the ball reaches can hold
      // anything nearer (or an equal-distance tie) -- usually one to three of them, not the whole 3x3x3 block.  All their row ranges
      // are fetched together (one round trip), then scanned.  If the ball pokes out of the block the general loop below takes over.
      const double b1 = cube_bound(g, c, q, 1);
      if (b1 == 1.0e300 || (b1 > 0.0 && (double)best < b1 * b1 * (1.0 - 1e-5))) {
        const double rad2 = (double)best * (1.0 + 1e-5);
        double wl[3], wh[3];
#pragma unroll
        for (int a = 0; a < 3; a++) {
          const double wall = cell_wall(g, a, c[a]);
          wl[a] = q[a] - wall;
          wh[a] = wall + g.res - q[a];
        }
        int ra[9], rb[9];
#pragma unroll
        for (int r = 0; r < 9; r++) {
          const int dy = r % 3 - 1, dz = r / 3 - 1;
          const int y = c[1] + dy, z = c[2] + dz;
          const double gy = dy < 0 ? wl[1] : (dy > 0 ? wh[1] : 0.0), gz = dz < 0 ? wl[2] : (dz > 0 ? wh[2] : 0.0);
          const double m = gy * gy + gz * gz;
          const bool need = y >= 0 && y < g.dim[1] && z >= 0 && z < g.dim[2] && m <= rad2;
          const int xa = (c[0] > 0 && wl[0] * wl[0] + m <= rad2) ? c[0] - 1 : c[0];
          const int xb = (c[0] < g.dim[0] - 1 && wh[0] * wh[0] + m <= rad2) ? c[0] + 1 : c[0];
          ra[r] = need ? tstart[cell_index(g, xa, y, z)] : 0;
          rb[r] = need ? tstart[cell_index(g, xb, y, z) + 1] : 0;
        }
#pragma unroll
        for (int r = 0; r < 9; r++) {
          if (r == 4) {  // the own row: its own cell has been scanned
            if (ra[4] < o0) scan(ra[4], o0);
            if (rb[4] > o1) scan(o1, rb[4]);
          } else if (rb[r] > ra[r]) {
            scan(ra[r], rb[r]);
          }
        }
        return;
      }
    }
  }
  for (;;) {
    // cube [c-r, c+r]^3 as contiguous row ranges of the sorted target (the minimum only improves on re-scans)
    for_each_cube_row(g, c, r, tstart, scan);
    if (r >= rmax) break;
    const double bound = cube_bound(g, c, q, r);
    if (bound == 1.0e300) break;
    if (bound > cap_r) break;  // everything unscanned is farther than the cap
    int rn;
    if (best < INFINITY) {
      if (bound > 0.0 && (double)best < bound * bound * (1.0 - 1e-5)) break;
    }
    if (unresolved) {
      *unresolved = true;
      return;
    }
    if (best < INFINITY) {
      const double need = sqrt((double)best) * (1.0 + 1e-5);
      rn = r + 1;
      while (rn < rmax) {
        const double b = cube_bound(g, c, q, rn);
        if (b == 1.0e300 || b > need) break;
        rn++;
      }
    } else {
      rn = r + max(1, (r + 1) / 2);
    }
    r = min(rn, rmax);
  }
}

// squared distance from the transformed source point i to its nearest target point
__device__ __forceinline__ float fitness_point(const float4* __restrict__ SP, int i, const PoseF& T, const float4* __restrict__ TP,
                                               const int* __restrict__ tstart, const Grid& g, bool* unresolved = nullptr, float* moved = nullptr) {
  const float4 sp = SP[i];
  const float x = sp.x, y = sp.y, z = sp.z;
  const float px = ((T.m[0] * x + T.m[1] * y) + T.m[2] * z) + T.m[3];
  const float py = ((T.m[4] * x + T.m[5] * y) + T.m[6] * z) + T.m[7];
  const float pz = ((T.m[8] * x + T.m[9] * y) + T.m[10] * z) + T.m[11];
  float best;
  int bs;
  nn_search<false>(px, py, pz, TP, tstart, g, 1.0e300, best, bs, unresolved);
  if (moved) { moved[0] = px; moved[1] = py; moved[2] = pz; }
  return best;
}

// f4  one ICP iteration's correspondences and sums (pcl::IterativeClosestPoint as used at RGC_mapping.cpp:2050-2069):
// nearest target point of every (already transformed) source point, kept if d^2 <= max_d2
// (CorrespondenceEstimation::determineCorrespondences), then the sums TransformationEstimationSVD needs --
// n, sum p, sum q, sum p q^T -- and sum d^2 for the convergence criteria: 17 of the 28 accumulators.
__global__ void __launch_bounds__(LIN_T)
k_icp_accumulate(const float4* __restrict__ SP, int ns, const float4* __restrict__ TP, const int* __restrict__ tstart, Grid g, double max_dist,
                 double max_d2, double* __restrict__ partials) {
  const int i = blockIdx.x * LIN_T + threadIdx.x;
  double acc[kAccum];
#pragma unroll
  for (int a = 0; a < kAccum; a++) acc[a] = 0.0;
  if (i < ns) {
    const float4 sp = SP[i];
    float best;
    int bs;
    nn_search<true>(sp.x, sp.y, sp.z, TP, tstart, g, max_dist * (1.0 + 1e-5), best, bs);
    if (bs >= 0 && (double)best <= max_d2) {
      const float4 tq = TP[bs];
      const double p[3] = {(double)sp.x, (double)sp.y, (double)sp.z}, q[3] = {(double)tq.x, (double)tq.y, (double)tq.z};
      acc[0] = 1.0;
#pragma unroll
      for (int a = 0; a < 3; a++) {
        acc[1 + a] = p[a];
        acc[4 + a] = q[a];
#pragma unroll
        for (int b = 0; b < 3; b++) acc[7 + a * 3 + b] = p[a] * q[b];
      }
      acc[16] = (double)best;
    }
  }
  block_reduce_store<kAccum>(acc, partials + (size_t)blockIdx.x * kAccum);
}

// The fold of the per-wave sums by ONE wave, in a fixed order (lane l: rows l, l + 64, ... ascending, eight fetched before any is added --
// one round trip, not eight; then the wave's shuffle tree): the score is the same bits wherever it is folded.
__device__ __forceinline__ double fitness_fold(const double* __restrict__ partials, int W) {
  const int lane = (int)threadIdx.x & (WAVE - 1);
  double t = 0;
  for (int r0 = lane; r0 < W; r0 += 8 * FIT_T) {
    double v8[8];
#pragma unroll
    for (int u = 0; u < 8; u++) v8[u] = (r0 + u * FIT_T < W) ? partials[r0 + u * FIT_T] : 0.0;
#pragma unroll
    for (int u = 0; u < 8; u++) t += v8[u];
  }
  return wave_sum(t);
}

// One wave's share of the scan against the map (FIT_T == WAVE); returns the wave's sum of squared nearest distances.
// n_all > 0 -- a SMALL map, the odometer's own three keyframes: ~11 k points, one per ten cells of its grid, and of a sweep's 11 k
// points several hundred lie 1-7 m from it (the keyframes hold the less-flat feature cloud, not whole sweeps).  Growing the cube for
// those is (2r+1)^2 mostly empty rows per level, and the launch lasted as long as its farthest lanes: 84 us.  Instead
//   * a query that its own cell and the first cube around it do not settle is handed to the WHOLE WAVE, which scans all n_all map points
//     for it (n_all / 64 coalesced loads per lane, the same dist2(), the minimum over the lanes: the exact nearest distance again), and
//   * the wave takes every (number of waves)-th point of the cell-sorted scan, not 64 consecutive ones: a sweep's far points are
//     neighbours, and a wave that holds 64 of them scans the map 64 times while the others idle.
// (w, W): this wave's number and the number of waves the scan is dealt to -- fitness_blocks(ns) everywhere, so that the per-wave sums and
// their fold (fitness_fold) are the same numbers whichever kernel computes them: k_fitness / k_fitness_lm (one-wave workgroups) or a step
// launch of the solve (k_lm_step: four waves per workgroup).
__device__ __forceinline__ double fitness_wave(const float4* __restrict__ SP, int ns, const PoseF& T, const float4* __restrict__ TP,
                                               const int* __restrict__ tstart, const Grid& g, int n_all, int w, int W) {
  static_assert(FIT_T == WAVE, "one wave per row");
  const int lane = (int)threadIdx.x & (WAVE - 1);
  const int i = n_all > 0 ? lane * W + w : w * FIT_T + lane;
  float best = 0.f, q[3] = {0.f, 0.f, 0.f};
  // (two calls, not one with a conditional pointer: a flag whose address is passed "maybe" lives in scratch memory -- one byte of it made
  // every launch of the solve a kernel with a private segment)
  bool unresolved = false;
  if (i < ns) {
    if (n_all > 0) best = fitness_point(SP, i, T, TP, tstart, g, &unresolved, q);
    else best = fitness_point(SP, i, T, TP, tstart, g, nullptr, q);
  }
  if (n_all > 0) {
    // (up to FOUR unsettled queries share one pass over the map: the pass is a chain of load round trips -- 21 of them for a 10 k-point
    // map, ~17 us -- and a sweep's far points come three or four to a wave: one pass instead of three or four.  A minimum does not depend
    // on the order it is taken in: the same bits.)
    unsigned long long todo = __ballot(unresolved);
    while (todo) {
      int l[4];
      float qx[4], qy[4], qz[4], m[4];
#pragma unroll
      for (int a = 0; a < 4; a++) {
        l[a] = todo ? __ffsll((long long)todo) - 1 : -1;
        todo &= todo - 1;  // (0 stays 0)
        const int src = l[a] < 0 ? 0 : l[a];
        qx[a] = __shfl(q[0], src); qy[a] = __shfl(q[1], src); qz[a] = __shfl(q[2], src);
        m[a] = INFINITY;
      }
      int s = lane;
      for (; s + 7 * WAVE < n_all; s += 8 * WAVE) {  // eight loads in flight per lane (sixteen: the same 46 us -- four queries per candidate is arithmetic, not round trips)
        float4 c[8];
#pragma unroll
        for (int u = 0; u < 8; u++) c[u] = TP[s + u * WAVE];
#pragma unroll
        for (int u = 0; u < 8; u++) {
#pragma unroll
          for (int a = 0; a < 4; a++) m[a] = fminf(m[a], dist2(qx[a], qy[a], qz[a], c[u]));
        }
      }
      for (; s < n_all; s += WAVE) {
        const float4 c = TP[s];
#pragma unroll
        for (int a = 0; a < 4; a++) m[a] = fminf(m[a], dist2(qx[a], qy[a], qz[a], c));
      }
#pragma unroll
      for (int a = 0; a < 4; a++) {
        const float all = __int_as_float(wave_min(__float_as_int(m[a])));  // (squared distances are >= +0: they order like their bit patterns)
        if (lane == l[a]) best = all;
      }
    }
  }
  return wave_sum(i < ns ? (double)best : 0.0);
}

__global__ void __launch_bounds__(FIT_T)
k_fitness(const float4* __restrict__ SP, int ns, PoseF T, const float4* __restrict__ TP,
          const int* __restrict__ tstart, Grid g, double* __restrict__ partials, int n_all) {
  const double v = fitness_wave(SP, ns, T, TP, tstart, g, n_all, (int)blockIdx.x, (int)gridDim.x);
  if (threadIdx.x == 0) partials[blockIdx.x] = v;
}

// the same with the final pose taken from the device-resident LM state; the last block folds the rows into the state
__global__ void __launch_bounds__(FIT_T)
k_fitness_lm(const float4* __restrict__ SP, int ns, LmState* __restrict__ st, const float4* __restrict__ TP,
             const int* __restrict__ tstart, Grid g, double* __restrict__ partials, LmState* __restrict__ h_post, int seq, int n_all) {
  if (!st->done || st->has_fit) return;  // enqueued blindly behind a batch of LM slots (and once more behind a later batch)
  wave_prio(2);
  PoseF T;
#pragma unroll
  for (int a = 0; a < 12; a++) T.m[a] = (float)st->x0[a];  // final_transformation_ = x0.cast<float>(), :77
  const double v = fitness_wave(SP, ns, T, TP, tstart, g, n_all, (int)blockIdx.x, (int)gridDim.x);
  if (threadIdx.x == 0) __hip_atomic_store(&partials[blockIdx.x], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // write-through row
  if (!last_block_arrive(&st->ticketB)) return;  // the LM is over: its ticket is free
  const double t = fitness_fold(partials, (int)gridDim.x);
  if (threadIdx.x == 0) {
    st->fit_sum = t;
    st->has_fit = 1;
    if (h_post) {
      __hip_atomic_store(&h_post->fit_sum, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(&h_post->has_fit, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
  if (h_post) post_state_to_host(st, h_post, seq, FIT_T, true);
}

__global__ void k_transform_f32(const float* __restrict__ in, int stride_f, int n, PoseF T, float* __restrict__ out, int ostride_f) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float* p = in + (size_t)i * stride_f;
  const float x = p[0], y = p[1], z = p[2];
  float* o = out + (size_t)i * ostride_f;
  o[0] = ((T.m[0] * x + T.m[1] * y) + T.m[2] * z) + T.m[3];
  o[1] = ((T.m[4] * x + T.m[5] * y) + T.m[6] * z) + T.m[7];
  o[2] = ((T.m[8] * x + T.m[9] * y) + T.m[10] * z) + T.m[11];
}

__global__ void k_unsort3(const double* __restrict__ a, const double* __restrict__ b, const double* __restrict__ c,
                          const float4* __restrict__ P, int n, double* __restrict__ out3) {
  int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n) return;
  const int i = __float_as_int(P[s].w);
  out3[(size_t)i * 3 + 0] = a[s];
  out3[(size_t)i * 3 + 1] = b[s];
  out3[(size_t)i * 3 + 2] = c[s];
}

// the inverse: three values per point given in the caller's order go to the sorted positions (setSource/TargetCovariances)
__global__ void k_sort3(const double* __restrict__ in3, const float4* __restrict__ P, int n, double* __restrict__ a, double* __restrict__ b,
                        double* __restrict__ c) {
  int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n) return;
  const int i = __float_as_int(P[s].w);
  a[s] = in3[(size_t)i * 3 + 0];
  b[s] = in3[(size_t)i * 3 + 1];
  c[s] = in3[(size_t)i * 3 + 2];
}

// First LM try of an outer iteration on the device (one lane): lsq_registration_impl.hpp:130-143.
// acc = the 28 folded doubles of linearize; out layout (doubles): [28] ncorr, [32..37] d, [38..53] xi (row-major 4x4),
// [54] lambda used, [55] 1 if the solve succeeded.
__global__ void k_lm_try(double* __restrict__ out, const int* __restrict__ ncorr, LmIn in) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double H[36], b[6];
  int u = 0;
#pragma unroll
  for (int a = 0; a < 6; a++)
#pragma unroll
    for (int e = a; e < 6; e++) { H[a * 6 + e] = out[u]; H[e * 6 + a] = out[u]; u++; }
#pragma unroll
  for (int a = 0; a < 6; a++) b[a] = out[21 + a];
  double lambda = in.lambda;
  if (lambda < 0.0) {
    double m = 0;
#pragma unroll
    for (int i = 0; i < 6; i++) m = fmax(m, fabs(H[i * 7]));
    lambda = in.init_factor * m;
  }
  double d[6], delta[16], xi[16];
  rgclm::lm_try(H, b, lambda, in.x0, d, delta, xi);
  out[28] = (double)*ncorr;
#pragma unroll
  for (int a = 0; a < 6; a++) out[32 + a] = d[a];
#pragma unroll
  for (int a = 0; a < 16; a++) out[38 + a] = xi[a];
  out[54] = lambda;
  out[55] = 1.0;
}

// C6 with the trial pose read from device memory (written by k_lm_try in the same stream)
__global__ void __launch_bounds__(LIN_T)
k_error_dev(const float4* __restrict__ P, int n, const double* __restrict__ Tdev, const double* __restrict__ vox, int noff,
            const int* __restrict__ corr_v, const double* __restrict__ corr_M, double* __restrict__ partials) {
  const int i = blockIdx.x * LIN_T + threadIdx.x;
  double acc[1] = {0.0};
  if (i < n) acc[0] = error_point(P, i, n, Tdev, vox, noff, corr_v, corr_M);
  block_reduce_store<1>(acc, partials + blockIdx.x);
}

// ------------------------------------------------------------------------------------------------
// launch wrappers
// ------------------------------------------------------------------------------------------------
static inline int nblk(long long n, int t) { return (int)((n + t - 1) / t); }

// ================================================================================================
// f1  Scan-to-map FEATURE registration of the mapping node (src/RGC_mapping.cpp:1069-1358): association kernels and the
// robustified normal equations of LidarEdgeFactor / LidarPlaneNormFactor (src/lidarFactor.hpp:9-51, 91-121).
// Factor record = 8 doubles: edge {a[3], b[3], var, valid}, plane {n[3], d, 0, 0, var, valid}.
// ================================================================================================
__device__ __forceinline__ void quat_rot_d(const Quat& q, const double p[3], double out[3]) {  // Eigen: quaternion * vector
  const double tx = 2 * (q.y * p[2] - q.z * p[1]), ty = 2 * (q.z * p[0] - q.x * p[2]), tz = 2 * (q.x * p[1] - q.y * p[0]);
  out[0] = p[0] + q.w * tx + (q.y * tz - q.z * ty);
  out[1] = p[1] + q.w * ty + (q.z * tx - q.x * tz);
  out[2] = p[2] + q.w * tz + (q.x * ty - q.y * tx);
}

// all three eigenpairs of a symmetric 3x3 (same cyclic Jacobi as min_eigenvector); ord[] = indices by DESCENDING eigenvalue
__device__ __forceinline__ void eig3_sym(const double S[6], double ev[3], double (&V)[3][3], int ord[3]) {
  double A[3][3] = {{S[0], S[1], S[2]}, {S[1], S[3], S[4]}, {S[2], S[4], S[5]}};
#pragma unroll
  for (int a = 0; a < 3; a++)
#pragma unroll
    for (int b = 0; b < 3; b++) V[a][b] = a == b ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 60; sweep++) {
    const double off = A[0][1] * A[0][1] + A[0][2] * A[0][2] + A[1][2] * A[1][2];
    const double diag = A[0][0] * A[0][0] + A[1][1] * A[1][1] + A[2][2] * A[2][2];
    if (off <= 1e-40 * diag || off == 0.0) break;
    jacobi_rot<0, 1, 2>(A, V);
    jacobi_rot<0, 2, 1>(A, V);
    jacobi_rot<1, 2, 0>(A, V);
  }
  ev[0] = A[0][0]; ev[1] = A[1][1]; ev[2] = A[2][2];
  ord[0] = 0; ord[1] = 1; ord[2] = 2;
  for (int i = 0; i < 2; i++)
    for (int j = i + 1; j < 3; j++)
      if (ev[ord[j]] > ev[ord[i]]) { const int t = ord[i]; ord[i] = ord[j]; ord[j] = t; }
}

// ---- the general covariance route (see GenOut above) ----
// fast_gicp_impl.hpp:256-293 for one point: mean and covariance of its k neighbours (fp64; neighbours in ascending position, the order every
// route of this file sums in), then the selected regularisation.  JacobiSVD of a symmetric positive semi-definite matrix is its
// eigen-decomposition (U = V), which is what is computed here; method = rgc_regularization_method (0 NONE, 1 MIN_EIG, 2 NORMALIZED_MIN_EIG,
// 3 PLANE, 4 FROBENIUS).  c6 = {xx, xy, xz, yy, yz, zz}, SoA: c6[a * n + i].
template <int KC>
__device__ __forceinline__ void cov6_of(const float4* __restrict__ P, const int (&idx)[KC], int k, int i, const GenOut& go) {
  double mx = 0.0, my = 0.0, mz = 0.0;
  for (int j = 0; j < k; j++) { const float4 p = P[idx[j]]; mx += (double)p.x; my += (double)p.y; mz += (double)p.z; }
  mx /= (double)k; my /= (double)k; mz /= (double)k;                         // neighbors.rowwise().mean(), :261
  double S[6] = {0, 0, 0, 0, 0, 0};
  for (int j = 0; j < k; j++) {
    const float4 p = P[idx[j]];
    const double x = (double)p.x - mx, y = (double)p.y - my, z = (double)p.z - mz;
    S[0] += x * x; S[1] += x * y; S[2] += x * z; S[3] += y * y; S[4] += y * z; S[5] += z * z;
  }
  for (int a = 0; a < 6; a++) S[a] /= (double)k;                             // :262
  double C[6];
  if (go.method == 0) {                                                       // NONE, :264-265
    for (int a = 0; a < 6; a++) C[a] = S[a];
  } else if (go.method == 4) {                                                // FROBENIUS, :266-271: (C_inv / |C_inv|_F)^-1 = |C_inv|_F (S + lambda I)
    const double R[6] = {S[0] + 1e-3, S[1], S[2], S[3] + 1e-3, S[4], S[5] + 1e-3};
    double Ci[6];
    if (!inv_sym3(R, Ci)) { for (int a = 0; a < 6; a++) Ci[a] = 0.0; }
    const double nrm = sqrt(Ci[0] * Ci[0] + Ci[3] * Ci[3] + Ci[5] * Ci[5] + 2.0 * (Ci[1] * Ci[1] + Ci[2] * Ci[2] + Ci[4] * Ci[4]));
    for (int a = 0; a < 6; a++) C[a] = nrm * R[a];
  } else {
    double ev[3], V[3][3];
    int ord[3];
    eig3_sym(S, ev, V, ord);                                                  // :273, singular values in descending order
    double val[3];
    const double smax = ev[ord[0]];
    for (int r = 0; r < 3; r++) {
      const double sv = ev[ord[r]];
      if (go.method == 3) val[r] = r < 2 ? 1.0 : 1e-3;                        // PLANE, :280-282
      else if (go.method == 1) val[r] = sv > 1e-3 ? sv : 1e-3;                // MIN_EIG, :283-285
      else { const double t = sv / smax; val[r] = t > 1e-3 ? t : 1e-3; }      // NORMALIZED_MIN_EIG, :286-289
    }
    for (int a = 0; a < 6; a++) C[a] = 0.0;
    for (int r = 0; r < 3; r++) {                                             // U diag(values) V^T, :293
      const int c = ord[r];
      const double v0 = V[0][c], v1 = V[1][c], v2 = V[2][c], w = val[r];
      C[0] += w * v0 * v0; C[1] += w * v0 * v1; C[2] += w * v0 * v2; C[3] += w * v1 * v1; C[4] += w * v1 * v2; C[5] += w * v2 * v2;
    }
  }
  for (int a = 0; a < 6; a++) go.c6[(size_t)a * go.n + i] = C[a];
}

// every point of a cloud through the cooperative search (a wave per query; wave w takes queries w, w + nwaves, ...)
template <int KC>
__global__ void __launch_bounds__(WAVE)
k_knn_cov6(const float4* __restrict__ P, const int* __restrict__ start, Grid g, int n, int k, GenOut go, const int* __restrict__ guard) {
  __shared__ CoopRows shm[1];
  if (guard && *guard) return;
  Deferred df{};
  for (int e = (int)blockIdx.x; e < n; e += (int)gridDim.x)
    coop_one<KC, false, true>(P, start, g, k, df, nullptr, nullptr, nullptr, &shm[0], (int)threadIdx.x, e, e, INFINITY, go);
}

// fast_vgicp_voxel.hpp:129-156 for the general route, one thread per grid cell, a voxel's points in ascending position (= the cloud's order
// inside a cell): ADDITIVE (:105-122) mean = sum p / n, cov = sum C / n; MULTIPLICATIVE (:76-99) cov = (sum C^-1)^-1, mean = cov * sum C^-1 p.
__global__ void __launch_bounds__(256)
k_voxel_build_general(const float4* __restrict__ P, const double* __restrict__ c6, const int* __restrict__ start, int ncell, int n,
                      const int* __restrict__ cell_voxel, double* __restrict__ vox, int* __restrict__ vox_cell, int multiplicative,
                      const int* __restrict__ guard) {
  if (guard && *guard) return;
  const int cell = blockIdx.x * 256 + threadIdx.x;
  if (cell >= ncell) return;
  const int a = start[cell], b = start[cell + 1];
  if (b <= a) return;
  const int v = cell_voxel[cell];
  double m[3] = {0, 0, 0}, A[6] = {0, 0, 0, 0, 0, 0};
  for (int u = a; u < b; u++) {
    const float4 p = P[u];
    double C[6];
    for (int t = 0; t < 6; t++) C[t] = c6[(size_t)t * n + u];
    if (multiplicative) {
      double Ci[6];
      if (!inv_sym3(C, Ci)) { for (int t = 0; t < 6; t++) Ci[t] = 0.0; }
      for (int t = 0; t < 6; t++) A[t] += Ci[t];
      m[0] += Ci[0] * (double)p.x + Ci[1] * (double)p.y + Ci[2] * (double)p.z;
      m[1] += Ci[1] * (double)p.x + Ci[3] * (double)p.y + Ci[4] * (double)p.z;
      m[2] += Ci[2] * (double)p.x + Ci[4] * (double)p.y + Ci[5] * (double)p.z;
    } else {
      for (int t = 0; t < 6; t++) A[t] += C[t];
      m[0] += (double)p.x; m[1] += (double)p.y; m[2] += (double)p.z;
    }
  }
  const double num = (double)(b - a);
  double* rec = vox + (size_t)v * kVoxRec;
  if (multiplicative) {
    double Cv[6];
    if (!inv_sym3(A, Cv)) { for (int t = 0; t < 6; t++) Cv[t] = 0.0; }
    rec[0] = Cv[0] * m[0] + Cv[1] * m[1] + Cv[2] * m[2];
    rec[1] = Cv[1] * m[0] + Cv[3] * m[1] + Cv[4] * m[2];
    rec[2] = Cv[2] * m[0] + Cv[4] * m[1] + Cv[5] * m[2];
    for (int t = 0; t < 6; t++) rec[3 + t] = Cv[t];
  } else {
    for (int t = 0; t < 3; t++) rec[t] = m[t] / num;
    for (int t = 0; t < 6; t++) rec[3 + t] = A[t] / num;
  }
  rec[9] = num;
  vox_cell[v] = cell;
}

// FastVGICP::update_correspondences + linearize (fast_vgicp_impl.hpp:73-180) for sorted source point i with a GENERAL source covariance:
// linearize_point_pre above with R C_A R^T computed from the six entries instead of from the normal
__device__ __forceinline__ void linearize_point_general(const float4 pp, const double* __restrict__ c6, int i, int n, const Pose& T, const Grid& g,
                                                        const int* __restrict__ cell_voxel, const double* __restrict__ vox, int noff,
                                                        int* __restrict__ corr_v, double* __restrict__ corr_M, int want_H, double (&acc)[kAccum],
                                                        int& ncorr) {
  const double p0 = (double)pp.x, p1 = (double)pp.y, p2 = (double)pp.z;
  const double q0 = T.R[0] * p0 + T.R[1] * p1 + T.R[2] * p2 + T.t[0];
  const double q1 = T.R[3] * p0 + T.R[4] * p1 + T.R[5] * p2 + T.t[1];
  const double q2 = T.R[6] * p0 + T.R[7] * p1 + T.R[8] * p2 + T.t[2];
  const double Cs[3][3] = {{c6[i], c6[(size_t)n + i], c6[2 * (size_t)n + i]},
                           {c6[(size_t)n + i], c6[3 * (size_t)n + i], c6[4 * (size_t)n + i]},
                           {c6[2 * (size_t)n + i], c6[4 * (size_t)n + i], c6[5 * (size_t)n + i]}};
  double RC[3][3], RCR[3][3];
  for (int a = 0; a < 3; a++)
    for (int b = 0; b < 3; b++) RC[a][b] = T.R[3 * a] * Cs[0][b] + T.R[3 * a + 1] * Cs[1][b] + T.R[3 * a + 2] * Cs[2][b];
  for (int a = 0; a < 3; a++)
    for (int b = 0; b < 3; b++) RCR[a][b] = RC[a][0] * T.R[3 * b] + RC[a][1] * T.R[3 * b + 1] + RC[a][2] * T.R[3 * b + 2];
  const double CA[6] = {RCR[0][0], RCR[0][1], RCR[0][2], RCR[1][1], RCR[1][2], RCR[2][2]};
  const int cx = (int)floor(q0 / g.res - 0.5) - g.minc[0];
  const int cy = (int)floor(q1 / g.res - 0.5) - g.minc[1];
  const int cz = (int)floor(q2 / g.res - 0.5) - g.minc[2];
  for (int o = 0; o < noff; o++) {
    int ox, oy, oz;
    neighbor_offset(noff, o, ox, oy, oz);
    const int x = cx + ox, y = cy + oy, z = cz + oz;
    int v = -1;
    if (x >= 0 && x < g.dim[0] && y >= 0 && y < g.dim[1] && z >= 0 && z < g.dim[2]) v = cell_voxel[cell_index(g, x, y, z)];
    const size_t slot = (size_t)o * n + i;
    corr_v[slot] = v;
    if (v < 0) continue;
    const double* rec = vox + (size_t)v * kVoxRec;
    double S[6], M[6];
    for (int a = 0; a < 6; a++) S[a] = rec[3 + a] + CA[a];
    if (!inv_sym3(S, M)) { for (int a = 0; a < 6; a++) M[a] = 0.0; }
    for (int a = 0; a < 6; a++) corr_M[((size_t)a * noff + o) * n + i] = M[a];
    ncorr++;
    const double e0 = rec[0] - q0, e1 = rec[1] - q1, e2 = rec[2] - q2;
    const double w = sqrt(rec[9]);
    const double Me0 = M[0] * e0 + M[1] * e1 + M[2] * e2;
    const double Me1 = M[1] * e0 + M[3] * e1 + M[4] * e2;
    const double Me2 = M[2] * e0 + M[4] * e1 + M[5] * e2;
    acc[27] += w * (e0 * Me0 + e1 * Me1 + e2 * Me2);
    if (!want_H) continue;
    const double J[3][6] = {{0.0, -q2, q1, -1.0, 0.0, 0.0}, {q2, 0.0, -q0, 0.0, -1.0, 0.0}, {-q1, q0, 0.0, 0.0, 0.0, -1.0}};
    double MJ[3][6];
    for (int c = 0; c < 6; c++) {
      MJ[0][c] = M[0] * J[0][c] + M[1] * J[1][c] + M[2] * J[2][c];
      MJ[1][c] = M[1] * J[0][c] + M[3] * J[1][c] + M[4] * J[2][c];
      MJ[2][c] = M[2] * J[0][c] + M[4] * J[1][c] + M[5] * J[2][c];
    }
    int u = 0;
    for (int a = 0; a < 6; a++)
      for (int c = a; c < 6; c++) { acc[u] += w * (J[0][a] * MJ[0][c] + J[1][a] * MJ[1][c] + J[2][a] * MJ[2][c]); u++; }
    for (int a = 0; a < 6; a++) acc[21 + a] += w * (J[0][a] * Me0 + J[1][a] * Me1 + J[2][a] * Me2);
  }
}
__global__ void __launch_bounds__(LIN_T)
k_linearize_general(const float4* __restrict__ P, const double* __restrict__ c6, int n, Pose T, Grid g, const int* __restrict__ cell_voxel,
                    const double* __restrict__ vox, int noff, int* __restrict__ corr_v, double* __restrict__ corr_M, int want_H,
                    double* __restrict__ partials, int* __restrict__ ncorr_partials) {
  const int i = blockIdx.x * LIN_T + threadIdx.x;
  double acc[kAccum];
#pragma unroll
  for (int a = 0; a < kAccum; a++) acc[a] = 0.0;
  int ncorr = 0;
  if (i < n) linearize_point_general(P[i], c6, i, n, T, g, cell_voxel, vox, noff, corr_v, corr_M, want_H, acc, ncorr);
  block_reduce_store<kAccum>(acc, partials + (size_t)blockIdx.x * kAccum);
  block_count_store(ncorr, ncorr_partials + blockIdx.x);
}
// caller order <-> sorted order for the six entries (getters / setters of the general route): out9 / in9 = n x 9 doubles, row-major 3x3
__global__ void k_unsort6(const double* __restrict__ c6, const float4* __restrict__ P, int n, double* __restrict__ out9) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int o = __float_as_int(P[i].w);
  double C[6];
  for (int a = 0; a < 6; a++) C[a] = c6[(size_t)a * n + i];
  double* d = out9 + (size_t)o * 9;
  d[0] = C[0]; d[1] = C[1]; d[2] = C[2]; d[3] = C[1]; d[4] = C[3]; d[5] = C[4]; d[6] = C[2]; d[7] = C[4]; d[8] = C[5];
}
__global__ void k_sort6(const double* __restrict__ in9, const float4* __restrict__ P, int n, double* __restrict__ c6) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double* s9 = in9 + (size_t)__float_as_int(P[i].w) * 9;
  c6[i] = s9[0]; c6[(size_t)n + i] = s9[1]; c6[2 * (size_t)n + i] = s9[2]; c6[3 * (size_t)n + i] = s9[4]; c6[4 * (size_t)n + i] = s9[5];
  c6[5 * (size_t)n + i] = s9[8];
}

// least-squares solution of the 5x3 system A x = b: Householder QR with column pivoting (Eigen::ColPivHouseholderQR)
__device__ void lstsq_5x3_colpiv(double A[5][3], double b[5], double x[3]) {
  int perm[3] = {0, 1, 2};
  int rank = 3;
  for (int k = 0; k < 3; k++) {
    int piv = k;
    double best = -1.0;
    for (int j = k; j < 3; j++) {
      double s = 0;
      for (int i = k; i < 5; i++) s += A[i][j] * A[i][j];
      if (s > best) { best = s; piv = j; }
    }
    if (best <= 0.0) { rank = k; break; }
    if (piv != k) {
      for (int i = 0; i < 5; i++) { const double t = A[i][k]; A[i][k] = A[i][piv]; A[i][piv] = t; }
      const int tp = perm[k]; perm[k] = perm[piv]; perm[piv] = tp;
    }
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
  x[0] = x[1] = x[2] = 0.0;
  for (int k = 0; k < 3; k++) x[perm[k]] = y[k];
}

// One lane per feature: pointAssociateToMap (:1811-1820, fp64 then stored as float), exact 5-NN in the map grid (the grid's
// cell is the threshold distance itself -- 1 m for edges :1098, sqrt(2) m for planes :1200 -- so the 3x3x3 block proves
// every 5th-neighbour distance below the threshold; a farther 5th neighbour means "no factor" either way), then the line
// test (:1100-1138) or the plane fit (:1202-1236).  *nvalid counts the factors created (corner_num, surf_num, ...).
template <bool kEdge>
__device__ __forceinline__ void mapreg_associate_one(const float* __restrict__ feat, int n, Quat q, double tx, double ty, double tz,
                                                     const float4* __restrict__ P, const int* __restrict__ start, const Grid& g,
                                                     double* __restrict__ fac, int* __restrict__ nvalid, int i) {
  if (i >= n) return;
  double* f = fac + (size_t)i * 8;
#pragma unroll
  for (int a = 0; a < 8; a++) f[a] = 0.0;
  const double p[3] = {(double)feat[4 * i], (double)feat[4 * i + 1], (double)feat[4 * i + 2]};
  double w[3];
  quat_rot_d(q, p, w);
  const float sx = (float)(w[0] + tx), sy = (float)(w[1] + ty), sz = (float)(w[2] + tz);
  const int c[3] = {cell_coord(sx, g) - g.minc[0], cell_coord(sy, g) - g.minc[1], cell_coord(sz, g) - g.minc[2]};
  float bd[5];
  int bs[5], bo[5];
#pragma unroll
  for (int j = 0; j < 5; j++) { bd[j] = INFINITY; bs[j] = -1; bo[j] = INT_MAX; }
  const int x0 = max(c[0] - 1, 0), x1 = min(c[0] + 1, g.dim[0] - 1);
  if (x0 <= x1) {
    for (int z = max(c[2] - 1, 0); z <= min(c[2] + 1, g.dim[2] - 1); z++)
      for (int y = max(c[1] - 1, 0); y <= min(c[1] + 1, g.dim[1] - 1); y++) {
        const int s0 = start[cell_index(g, x0, y, z)], s1 = start[cell_index(g, x1, y, z) + 1];
        for (int s = s0; s < s1; s++) {
          const float4 cp = P[s];
          float d = dist2(sx, sy, sz, cp);
          int o = __float_as_int(cp.w), ss = s;
          // insertion into the ascending (distance, original index) list
#pragma unroll
          for (int j = 0; j < 5; j++) {
            const bool before = d < bd[j] || (d == bd[j] && o < bo[j]);
            if (before) {
              const float td = bd[j]; bd[j] = d; d = td;
              const int ts = bs[j]; bs[j] = ss; ss = ts;
              const int to = bo[j]; bo[j] = o; o = to;
            }
          }
        }
      }
  }
  const float limit = kEdge ? 1.0f : 2.0f;
  if (!(bd[4] < limit)) return;  // :1098 / :1200 (also: fewer than five points in reach)
  double Q[5][3];
#pragma unroll
  for (int j = 0; j < 5; j++) {
    const float4 cp = P[bs[j]];
    Q[j][0] = (double)cp.x; Q[j][1] = (double)cp.y; Q[j][2] = (double)cp.z;
  }
  if (kEdge) {
    double ctr[3] = {0, 0, 0};
#pragma unroll
    for (int j = 0; j < 5; j++) { ctr[0] += Q[j][0]; ctr[1] += Q[j][1]; ctr[2] += Q[j][2]; }
    ctr[0] /= 5.0; ctr[1] /= 5.0; ctr[2] /= 5.0;
    double S[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < 5; j++) {
      const double zx = Q[j][0] - ctr[0], zy = Q[j][1] - ctr[1], zz = Q[j][2] - ctr[2];
      S[0] += zx * zx; S[1] += zx * zy; S[2] += zx * zz; S[3] += zy * zy; S[4] += zy * zz; S[5] += zz * zz;
    }
    double ev[3], V[3][3];
    int ord[3];
    eig3_sym(S, ev, V, ord);
    if (!(ev[ord[0]] > 3 * ev[ord[1]])) return;  // :1122 (Eigen sorts ascending: eigenvalues()[2] > 3 eigenvalues()[1])
    const int m = ord[0];
#pragma unroll
    for (int a = 0; a < 3; a++) {
      const double dir = m == 0 ? V[a][0] : (m == 1 ? V[a][1] : V[a][2]);
      f[a] = 0.1 * dir + ctr[a];       // point_a
      f[3 + a] = -0.1 * dir + ctr[a];  // point_b
    }
  } else {
    double A[5][3], b[5] = {-1, -1, -1, -1, -1};
#pragma unroll
    for (int j = 0; j < 5; j++) { A[j][0] = Q[j][0]; A[j][1] = Q[j][1]; A[j][2] = Q[j][2]; }
    double nrm[3];
    lstsq_5x3_colpiv(A, b, nrm);
    const double nn = sqrt(nrm[0] * nrm[0] + nrm[1] * nrm[1] + nrm[2] * nrm[2]);
    if (!(nn > 0)) return;
    const double d = 1.0 / nn;  // negative_OA_dot_norm
    nrm[0] /= nn; nrm[1] /= nn; nrm[2] /= nn;
#pragma unroll
    for (int j = 0; j < 5; j++)
      if (fabs(nrm[0] * Q[j][0] + nrm[1] * Q[j][1] + nrm[2] * Q[j][2] + d) > 0.2) return;  // :1218-1226
    f[0] = nrm[0]; f[1] = nrm[1]; f[2] = nrm[2]; f[3] = d;
  }
  f[6] = (double)feat[4 * i + 3];  // var = pointOri.normal_x
  f[7] = 1.0;
  if (nvalid) atomicAdd(nvalid, 1);
}

// up to four association loops in ONE launch (blockIdx.y selects the loop): they are independent, each is a few thousand
// lanes of latency-bound search, and side by side they cost the longest one instead of the sum
__global__ void k_mapreg_associate(MapregAssoc a0, MapregAssoc a1, MapregAssoc a2, MapregAssoc a3) {
  const MapregAssoc& a = blockIdx.y == 0 ? a0 : (blockIdx.y == 1 ? a1 : (blockIdx.y == 2 ? a2 : a3));
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (a.edge) mapreg_associate_one<true>(a.feat, a.n, a.q, a.t[0], a.t[1], a.t[2], a.P, a.start, a.g, a.fac, a.nvalid, i);
  else mapreg_associate_one<false>(a.feat, a.n, a.q, a.t[0], a.t[1], a.t[2], a.P, a.start, a.g, a.fac, a.nvalid, i);
}

// Robustified normal equations of ONE pose over its edge and plane factors (or the cost only): 21 + 6 + 1 sums per lane
// -> the same block reduction and fixed-order fold as the registration's linearisation.  Local parameterisation =
// EigenQuaternionParameterization (q' = dq (x) q, dq = (sin|d|/|d| d, cos|d|)): d(R p)/dd = -2 [R p]x.  HuberLoss(a) has
// rho'' <= 0, so Ceres' corrector scales residual and Jacobian by sqrt(rho'): the sums carry the weight rho'.
struct MapregPose {  // one pose's feature sets and estimate
  const float* cfeat; const double* efac; int ne;
  const float* sfeat; const double* pfac; int np;
  Quat q; double t[3];
};
__global__ void __launch_bounds__(LIN_T)
k_mapreg_terms(MapregPose pose0, MapregPose pose1, double huber_a, int want_H, double* __restrict__ partials) {
  // blockIdx.y selects the pose (current / last): one launch and one fold for both
  const MapregPose& ps = blockIdx.y ? pose1 : pose0;
  const float* __restrict__ cfeat = ps.cfeat; const double* __restrict__ efac = ps.efac; const int ne = ps.ne;
  const float* __restrict__ sfeat = ps.sfeat; const double* __restrict__ pfac = ps.pfac; const int np = ps.np;
  const Quat q = ps.q;
  const double tx = ps.t[0], ty = ps.t[1], tz = ps.t[2];
  const int i = blockIdx.x * LIN_T + threadIdx.x;
  double acc[kAccum];
#pragma unroll
  for (int a = 0; a < kAccum; a++) acc[a] = 0.0;
  const bool is_edge = i < ne;
  const int j = is_edge ? i : i - ne;
  if (i < ne + np) {
    const double* f = (is_edge ? efac : pfac) + (size_t)j * 8;
    if (f[7] != 0.0) {
      const float* fp = (is_edge ? cfeat : sfeat) + 4 * (size_t)j;
      const double p[3] = {(double)fp[0], (double)fp[1], (double)fp[2]};
      double Rp[3];
      quat_rot_d(q, p, Rp);
      const double lp[3] = {Rp[0] + tx, Rp[1] + ty, Rp[2] + tz};
      double r[3] = {0, 0, 0}, J[3][6];
      int dim;
      if (is_edge) {
        dim = 3;
        const double u[3] = {lp[0] - f[0], lp[1] - f[1], lp[2] - f[2]}, v[3] = {lp[0] - f[3], lp[1] - f[4], lp[2] - f[5]};
        const double nu[3] = {u[1] * v[2] - u[2] * v[1], u[2] * v[0] - u[0] * v[2], u[0] * v[1] - u[1] * v[0]};
        const double de[3] = {f[0] - f[3], f[1] - f[4], f[2] - f[5]};
        const double sc = f[6] / sqrt(de[0] * de[0] + de[1] * de[1] + de[2] * de[2]);
        r[0] = nu[0] * sc; r[1] = nu[1] * sc; r[2] = nu[2] * sc;
        // d nu / d lp = -[de]x ; columns 0..2: (-[de]x)(-2 [Rp]x) = 2 [de]x [Rp]x ; columns 3..5: -[de]x
        const double Sd[3][3] = {{0, -de[2], de[1]}, {de[2], 0, -de[0]}, {-de[1], de[0], 0}};
        const double Sr[3][3] = {{0, -Rp[2], Rp[1]}, {Rp[2], 0, -Rp[0]}, {-Rp[1], Rp[0], 0}};
#pragma unroll
        for (int a = 0; a < 3; a++)
#pragma unroll
          for (int c = 0; c < 3; c++) {
            J[a][c] = sc * 2.0 * (Sd[a][0] * Sr[0][c] + Sd[a][1] * Sr[1][c] + Sd[a][2] * Sr[2][c]);
            J[a][3 + c] = -sc * Sd[a][c];
          }
      } else {
        dim = 1;
        r[0] = (f[0] * lp[0] + f[1] * lp[1] + f[2] * lp[2] + f[3]) * f[6];
        const double nx_[3] = {f[1] * Rp[2] - f[2] * Rp[1], f[2] * Rp[0] - f[0] * Rp[2], f[0] * Rp[1] - f[1] * Rp[0]};  // n x Rp
#pragma unroll
        for (int c = 0; c < 3; c++) { J[0][c] = -2.0 * nx_[c] * f[6]; J[0][3 + c] = f[c] * f[6]; }
#pragma unroll
        for (int c = 0; c < 6; c++) { J[1][c] = 0.0; J[2][c] = 0.0; }
      }
      const double s = r[0] * r[0] + r[1] * r[1] + r[2] * r[2];
      const double b2 = huber_a * huber_a;
      double rho, rho1;
      if (s > b2) { const double sq = sqrt(s); rho = 2 * huber_a * sq - b2; rho1 = huber_a / sq; }
      else { rho = s; rho1 = 1.0; }
      acc[27] = 0.5 * rho;
      if (want_H) {
        int u = 0;
#pragma unroll
        for (int a = 0; a < 6; a++)
#pragma unroll
          for (int c = a; c < 6; c++) {
            double v = J[0][a] * J[0][c];
            if (dim == 3) v += J[1][a] * J[1][c] + J[2][a] * J[2][c];
            acc[u++] = rho1 * v;
          }
#pragma unroll
        for (int a = 0; a < 6; a++) {
          double v = J[0][a] * r[0];
          if (dim == 3) v += J[1][a] * r[1] + J[2][a] * r[2];
          acc[21 + a] = rho1 * v;
        }
      }
    }
  }
  block_reduce_store<kAccum>(acc, partials + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * kAccum);
}

// fold of the two poses' rows: blocks 0..27 -> pose 0, 28..55 -> pose 1
__global__ void __launch_bounds__(WAVE) k_mapreg_fold(const double* __restrict__ partials, int nrows, double* __restrict__ out56) {
  const int b = blockIdx.x / kAccum, a = blockIdx.x % kAccum, lane = threadIdx.x;
  const double* base = partials + (size_t)b * nrows * kAccum;
  double s = 0;
  for (int r = lane; r < nrows; r += WAVE) s += base[(size_t)r * kAccum + a];
  s = wave_sum(s);
  if (lane == 0) out56[b * kAccum + a] = s;
}

void mapreg_associate(hipStream_t s, const MapregAssoc* sets, int nsets) {
  MapregAssoc a[4] = {};
  int nmax = 0;
  for (int k = 0; k < 4; k++) {
    if (k < nsets) a[k] = sets[k];
    nmax = max(nmax, a[k].n);
  }
  if (nmax <= 0 || nsets <= 0) return;
  hipLaunchKernelGGL(k_mapreg_associate, dim3(nblk(nmax, 64), nsets), dim3(64), 0, s, a[0], a[1], a[2], a[3]);
}
int mapreg_blocks(int ne, int np) { return (ne + np + LIN_T - 1) / LIN_T; }
void mapreg_terms(hipStream_t s, const float* const feat[4], const double* const fac[4], const int nfeat[4], const double x14[14], double huber_a,
                  int want_H, double* partials, double* out56) {
  MapregPose ps[2];
  for (int b = 0; b < 2; b++) {
    const double* q = x14 + 7 * b;
    ps[b] = MapregPose{feat[2 * b], fac[2 * b], nfeat[2 * b], feat[2 * b + 1], fac[2 * b + 1], nfeat[2 * b + 1], Quat{q[0], q[1], q[2], q[3]},
                       {q[4], q[5], q[6]}};
  }
  const int nb = max(max(mapreg_blocks(nfeat[0], nfeat[1]), mapreg_blocks(nfeat[2], nfeat[3])), 1);
  hipLaunchKernelGGL(k_mapreg_terms, dim3(nb, 2), dim3(LIN_T), 0, s, ps[0], ps[1], huber_a, want_H, partials);
  hipLaunchKernelGGL(k_mapreg_fold, dim3(2 * kAccum), dim3(WAVE), 0, s, partials, nb, out56);
}

void bbox(hipStream_t s, const float* in, int stride_f, int n, double res, int* mm6, int* flags, int hi) {
  hipLaunchKernelGGL(k_bbox, dim3(min(nblk(n, 256), 1024)), dim3(256), 0, s, in, stride_f, n, res, mm6, flags, hi);
}
void count_cells(hipStream_t s, const float* in, int stride_f, int n, Grid g, int* cell_of, int* slot_of, int* cnt, int hi, int* guard,
                 const Reframe* rf) {
  if (rf) hipLaunchKernelGGL(k_count<true>, dim3(nblk(n, RGC_GRID_T)), dim3(RGC_GRID_T), 0, s, in, 4, n, g, cell_of, slot_of, cnt, guard, hi, *rf);
  else hipLaunchKernelGGL(k_count<false>, dim3(nblk(n, RGC_GRID_T)), dim3(RGC_GRID_T), 0, s, in, stride_f, n, g, cell_of, slot_of, cnt, guard, hi, Reframe{});
}
void scan_cells(hipStream_t s, int* cnt, int* start, int n, void* block_sums, int* cell_voxel, int* nvox, int hi, float* sum_sq) {
  const int nb = nblk(n, SCAN_B);
  unsigned long long* bs = (unsigned long long*)block_sums;
  if (nb > 1 || sum_sq) hipLaunchKernelGGL(k_cells_reduce, dim3(nb), dim3(SCAN_T), 0, s, cnt, n, bs, hi, sum_sq);
  if (nb <= 4096) {
    hipLaunchKernelGGL(k_cells_scan_write<true>, dim3(nb), dim3(SCAN_T), 0, s, cnt, start, n, bs, nb, cell_voxel, nvox, hi);
  } else {
    hipLaunchKernelGGL(k_cells_scan_sums, dim3(1), dim3(SCAN_T), 0, s, bs, nb, nvox, hi);
    hipLaunchKernelGGL(k_cells_scan_write<false>, dim3(nb), dim3(SCAN_T), 0, s, cnt, start, n, bs, nb, cell_voxel, nvox, hi);
  }
}
void place(hipStream_t s, int n, const int* cell_of, const int* slot_of, const int* start, unsigned long long* order_tmp, int hi, const KnnCache* cache) {
  hipLaunchKernelGGL(k_place, dim3(nblk(n, RGC_GRID_T)), dim3(RGC_GRID_T), 0, s, n, cell_of, slot_of, start, order_tmp, hi, cache ? *cache : KnnCache{});
}
void exclusive_scan(hipStream_t s, const int* in, int* out, int n, int* block_sums, int hi) {
  const int nb = nblk(n, SCAN_B);
  hipLaunchKernelGGL(k_scan_block, dim3(nb), dim3(SCAN_T), 0, s, in, out, n, block_sums, hi);
  if (nb > 1) {
    hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(SCAN_T), 0, s, block_sums, nb, hi);
    hipLaunchKernelGGL(k_scan_add, dim3(nb), dim3(SCAN_T), 0, s, out, n, block_sums, hi);
  }
}
void rank_gather(hipStream_t s, const float* in, int stride_f, int n, const int* cell_of, const int* start,
                 const unsigned long long* order_tmp, float4* P, int* zero_me, int hi, const KnnCache* cache) {
  hipLaunchKernelGGL(k_rank_gather, dim3(nblk(n, RGC_GRID_T)), dim3(RGC_GRID_T), 0, s, in, stride_f, n, cell_of, start, order_tmp, P, zero_me, hi,
                     cache ? *cache : KnnCache{});
}
size_t deferred_bytes(int n) { return sizeof(int) * (2 * (size_t)n + 16); }

// deferred list: [cnt, pad x15][idx n][thr n]
static Deferred deferred_of(const void* buf, int n) {
  int* base = (int*)const_cast<void*>(buf);
  return Deferred{base + 16, (float*)(base + 16 + (size_t)n), base, nullptr, nullptr, nullptr, nullptr, 0.f, KnnCache{}, 0,
                  0, base + 1, reinterpret_cast<unsigned long long*>(base + 16), 0};
}

template <int KC, bool kExact>
static void knn_bulk_kc(hipStream_t s, bool is_target, const float4* P, const int* start, Grid g, int n, int k, const void* deferred,
                        double* nx, double* ny, double* nz, const int* guard, int wide_r, hipEvent_t ev0, hipEvent_t ev1,
                        const int* qlist, const int* nq, int q_est, const KnnSeeds& seeds, int stream_coop_waves) {
  Deferred df = deferred_of(deferred, n);  // df.cnt was zeroed by k_rank_gather
  df.guard = guard;
  df.split_sums = is_target ? 0 : 1;
  df.qlist = qlist; df.nq = nq;
  // seeds: the map's dense search at k == KC only; `warm` = some search has written them (the seeded kernel is worth launching)
  const bool seeds_ok = is_target && kExact && wide_r != 2 && seeds.seed && n <= kSeedMaxPoints;
  df.seed = seeds_ok ? seeds.seed : nullptr;
  df.seed_slack = seeds.slack;
  const bool seeded = seeds_ok && seeds.warm;
  if (seeds_ok && seeds.cache.nbr) df.cache = seeds.cache;  // (the searches write the lists and certificates from the first launch on)
  if (wide_r == 2) {
    const size_t ldsw = (size_t)SpShape<2, true>::LDS * WAVE * sizeof(int);
    hipLaunchKernelGGL((k_knn_sp_wide<KC, 2, kExact>), dim3(nblk(n, WAVE / 4)), dim3(WAVE), ldsw, s, P, start, g, n, k, df, nx, ny, nz);
    return;
  }
  using CT = SpConfig<true>;
  using CS = SpConfig<false>;
  const int T = is_target ? (seeded && !qlist ? kSeedT : CT::T) : CS::T;
  // (the scan's launch lays its per-lane LDS columns out for the 3x3x3 block and again, for the queries that block does not settle, for the 5x5x5 one)
  const size_t lds = (size_t)(is_target ? (seeded ? std::max(SpShape<CT::R, CT::kClip>::LDS, SeedShape<KC>::LDS) : SpShape<CT::R, CT::kClip>::LDS)
                                        : std::max(SpShape<CS::R, CS::kClip>::LDS, SpShape<2, CS::kClip>::LDS)) * T * sizeof(int);
  // whole rounds of 8 XCDs x RGC_XCD_RUN blocks (excess blocks fall out at i >= n); lazy target: as many blocks as the listed queries are
  // expected to fill (the kernel strides over the list whatever its true length)
  const int n_launch = (is_target && qlist) ? (q_est < T ? T : (q_est > n ? n : q_est)) : n;
  const int xcd_run = RGC_XCD_RUN * KNN_T / T;
  int nb = (is_target && qlist) ? nblk(n_launch, T) : 8 * xcd_run * nblk(nblk(n, T), 8 * xcd_run);
  if (seeded && !qlist && df.cache.nbr) {
    // the workgroups in front of the bulk ones search the listed queries: sized for 4 % of the map (they stride over longer lists)
    df.cache_nb = kTodoLists * std::max(1, nblk((int)(0.04 * (double)n / kTodoLists) + 1, T));
    nb += df.cache_nb;
  }
  // (ev0 / ev1: the launch's own start / stop times go into the caller's events -- no separate record packets around it)
  if constexpr (kExact) {
    if (seeded) {
      if (qlist) hipLaunchKernelGGL((k_knn_sp_listed<KC, true, true>), dim3(nb), dim3(T), lds, s, P, start, g, n, k, df, nx, ny, nz);
      else if (ev0 && ev1) hipExtLaunchKernelGGL((k_knn_sp<KC, true, true, true>), dim3(nb), dim3(T), (std::uint32_t)lds, s, ev0, ev1, 0u, P, start, g, n, k, df, nx, ny, nz);
      else hipLaunchKernelGGL((k_knn_sp<KC, true, true, true>), dim3(nb), dim3(T), lds, s, P, start, g, n, k, df, nx, ny, nz);
      return;
    }
  }
  if (is_target && qlist) hipLaunchKernelGGL((k_knn_sp_listed<KC, kExact>), dim3(nb), dim3(T), lds, s, P, start, g, n, k, df, nx, ny, nz);
  else if (is_target && ev0 && ev1) hipExtLaunchKernelGGL((k_knn_sp<KC, true, kExact>), dim3(nb), dim3(T), (std::uint32_t)lds, s, ev0, ev1, 0u, P, start, g, n, k, df, nx, ny, nz);
  else if (is_target) hipLaunchKernelGGL((k_knn_sp<KC, true, kExact>), dim3(nb), dim3(T), lds, s, P, start, g, n, k, df, nx, ny, nz);
  else {  // four lanes per query; the deferred queries resolved by the launch's last workgroups (coop_stream) when the caller asks for it
    if (stream_coop_waves > 0) df.coop_blocks = nblk(stream_coop_waves < 32 ? 32 : (stream_coop_waves > 8192 ? 8192 : stream_coop_waves), T / WAVE);
    static_assert(sizeof(CoopRows) * (CS::T / WAVE) <= (size_t)SpShape<CS::R, CS::kClip>::LDS * CS::T * sizeof(int), "the cooperative waves' scratch fits the bulk launch's LDS");
    hipLaunchKernelGGL((k_knn_sp<KC, false, kExact>), dim3(nblk(n, T / 4) + df.coop_blocks), dim3(T), lds, s, P, start, g, n, k, df, nx, ny, nz);
  }
}
bool knn_seeds_apply(int n, int k) { return k == 20 && n <= kSeedMaxPoints; }
static void deferred_seeds(Deferred& df, const KnnSeeds& seeds, int n, int k) {  // the cooperative search leaves its k-th distance as the point's seed
  if (seeds.seed && knn_seeds_apply(n, k)) { df.seed = seeds.seed; df.seed_slack = seeds.slack; }
}
template <int KC>
static void knn_coop_kc(hipStream_t s, bool is_target, const float4* P, const int* start, Grid g, int n, int k, const void* segs,
                        double* nx, double* ny, double* nz, const int* guard, int waves, const KnnSeeds& seeds) {
  Deferred df = deferred_of(segs, n);
  df.guard = guard;
  if (is_target) deferred_seeds(df, seeds, n, k);
  // the number of deferred queries is only known on the device: `waves` one-wave workgroups share the list (each takes every
  // waves-th entry); the caller sizes it from the previous cloud of the sequence
  const int nbc = waves < 32 ? 32 : (waves > 8192 ? 8192 : waves);
  if (is_target)
    hipLaunchKernelGGL((k_knn_coop<KC, true>), dim3(nbc), dim3(WAVE), 0, s, P, start, g, k, df, nx, ny, nz);
  else
    hipLaunchKernelGGL((k_knn_coop<KC, false>), dim3(nbc), dim3(WAVE), 0, s, P, start, g, k, df, nx, ny, nz);
}
bool knn_bulk_times_itself(bool is_target, int wide_r) { return is_target && wide_r != 2; }
void knn_bulk(hipStream_t s, bool is_target, const float4* P, const int* start, Grid g, int n, int k, const void* deferred, double* nx,
              double* ny, double* nz, const int* guard, int wide_r, hipEvent_t ev0, hipEvent_t ev1, const int* qlist, const int* nq, int q_est,
              const KnnSeeds& seeds, int stream_coop_waves) {
  if (is_target || wide_r == 2) stream_coop_waves = 0;  // (the scan's four-lane search only)
  // (k == 20, the reference's setting, gets an instance without the general-k branches)
  if (k == 20) knn_bulk_kc<20, true>(s, is_target, P, start, g, n, k, deferred, nx, ny, nz, guard, wide_r, ev0, ev1, qlist, nq, q_est, seeds, stream_coop_waves);
  else if (k < 20) knn_bulk_kc<20, false>(s, is_target, P, start, g, n, k, deferred, nx, ny, nz, guard, wide_r, ev0, ev1, qlist, nq, q_est, seeds, stream_coop_waves);
  else knn_bulk_kc<32, false>(s, is_target, P, start, g, n, k, deferred, nx, ny, nz, guard, wide_r, ev0, ev1, qlist, nq, q_est, seeds, stream_coop_waves);
}

void knn_coop(hipStream_t s, bool is_target, const float4* P, const int* start, Grid g, int n, int k, const void* segs, double* nx,
              double* ny, double* nz, const int* guard, int waves, const KnnSeeds& seeds) {
  if (k <= 20) knn_coop_kc<20>(s, is_target, P, start, g, n, k, segs, nx, ny, nz, guard, waves, seeds);
  else knn_coop_kc<32>(s, is_target, P, start, g, n, k, segs, nx, ny, nz, guard, waves, seeds);
}
void footprint(hipStream_t s, const float* in, int stride_f, int n, Pose T, Grid g, int* need, int stamp, int margin, const float4* P, int n_map,
               const int* start, int* qlist, int* cell_list, int* counts, const int* guard) {
  if (n > 0) hipLaunchKernelGGL(k_footprint, dim3(nblk(n, 256)), dim3(256), 0, s, in, stride_f, n, T, g, need, stamp, margin);
  if (n_map > 0) hipLaunchKernelGGL(k_lazy_lists, dim3(nblk(n_map, kListT)), dim3(kListT), 0, s, P, n_map, g, start, need, stamp, qlist, cell_list, counts, guard);
}
void voxel_cells_coop(hipStream_t s, const float4* P, double* nx, double* ny, double* nz, const int* start, Grid g, int n, const int* cell_voxel,
                      double* vox, int* vox_cell, int k, const void* deferred, const int* guard, int waves, const int* cell_list, const int* ncells,
                      int cells_est, const KnnSeeds& seeds) {
  Deferred df = deferred_of(deferred, n);
  df.guard = guard;
  deferred_seeds(df, seeds, n, k);
  const int nbc = nblk(waves < 32 ? 32 : (waves > 8192 ? 8192 : waves), VOX_T / WAVE);
  const int nbv = nblk(cells_est < 256 ? 256 : (cells_est > (1 << 20) ? (1 << 20) : cells_est), VOX_T / WAVE);  // a wave per listed cell (grid-stride beyond the estimate)
  if (k <= 20) hipLaunchKernelGGL((k_voxel_cells_coop<20>), dim3(nbc + nbv), dim3(VOX_T), 0, s, P, nx, ny, nz, start, g, cell_voxel, vox, vox_cell, nbc, k, df, cell_list, ncells);
  else hipLaunchKernelGGL((k_voxel_cells_coop<32>), dim3(nbc + nbv), dim3(VOX_T), 0, s, P, nx, ny, nz, start, g, cell_voxel, vox, vox_cell, nbc, k, df, cell_list, ncells);
}
void voxel_build(hipStream_t s, const float4* P, const double* nx, const double* ny, const double* nz, const int* start, Grid g,
                 int n, const int* cell_voxel, double* vox, int* vox_cell) {
  hipLaunchKernelGGL(k_voxel_build, dim3(nblk(n, VOX_T)), dim3(VOX_T), 0, s, P, nx, ny, nz, start, g, n, cell_voxel, vox, vox_cell);
}
void voxel_build_coop(hipStream_t s, const float4* P, double* nx, double* ny, double* nz, const int* start, Grid g, int n, const int* cell_voxel,
                      double* vox, int* vox_cell, int k, const void* deferred, const int* guard, int waves, const KnnSeeds& seeds) {
  Deferred df = deferred_of(deferred, n);
  df.guard = guard;
  deferred_seeds(df, seeds, n, k);
  const int nbv = nblk(n, VOX_T);
  const int nbc = nblk(waves < 32 ? 32 : (waves > 8192 ? 8192 : waves), VOX_T / WAVE);
  if (k <= 20) hipLaunchKernelGGL((k_voxel_build_coop<20>), dim3(nbv + nbc), dim3(VOX_T), 0, s, P, nx, ny, nz, start, g, n, cell_voxel, vox, vox_cell, nbc, k, df);
  else hipLaunchKernelGGL((k_voxel_build_coop<32>), dim3(nbv + nbc), dim3(VOX_T), 0, s, P, nx, ny, nz, start, g, n, cell_voxel, vox, vox_cell, nbc, k, df);
}
void voxel_patch(hipStream_t s, const float4* P, const double* nx, const double* ny, const double* nz, const int* start, Grid g, const void* deferred,
                 const int* cell_voxel, double* vox, int lanes, hipEvent_t done) {
  const int nb = lanes < 64 ? 64 : (lanes > 4096 ? 4096 : lanes);  // one-wave workgroups, one deferred entry each (grid-stride beyond)
  // (done: the launch's own completion is the event -- no record packet between the map's last kernel and the solve's first step)
  if (done) hipExtLaunchKernelGGL(k_voxel_patch, dim3(nb), dim3(WAVE), 0u, s, nullptr, done, 0u, P, nx, ny, nz, start, g, (const int*)deferred, cell_voxel, vox);
  else hipLaunchKernelGGL(k_voxel_patch, dim3(nb), dim3(WAVE), 0, s, P, nx, ny, nz, start, g, (const int*)deferred, cell_voxel, vox);
}
void linearize(hipStream_t s, const float4* P, const double* nx, const double* ny, const double* nz, int n, Pose T, Grid g,
               const int* cell_voxel, const double* vox, int noff, int* corr_v, double* corr_M, int want_H, double* partials,
               int* ncorr_partials, double* out28, int* out_ncorr) {
  const int nb = linearize_blocks(n);
  hipLaunchKernelGGL(k_linearize, dim3(nb), dim3(LIN_T), 0, s, P, nx, ny, nz, n, T, g, cell_voxel, vox, noff, corr_v, corr_M, want_H, partials, ncorr_partials);
  hipLaunchKernelGGL(k_fold<kAccum>, dim3(kAccum + 1), dim3(WAVE), 0, s, partials, nb, out28, ncorr_partials, out_ncorr);
}
// ---- the general covariance route ----
void knn_cov6(hipStream_t s, const float4* P, const int* start, Grid g, int n, int k, int method, double* c6, const int* guard) {
  const GenOut go{c6, method, n};
  const int waves = n < 16384 ? n : 16384;
  if (k <= 20) hipLaunchKernelGGL(k_knn_cov6<20>, dim3(waves), dim3(WAVE), 0, s, P, start, g, n, k, go, guard);
  else hipLaunchKernelGGL(k_knn_cov6<32>, dim3(waves), dim3(WAVE), 0, s, P, start, g, n, k, go, guard);
}
void voxel_build_general(hipStream_t s, const float4* P, const double* c6, const int* start, Grid g, int n, const int* cell_voxel, double* vox,
                         int* vox_cell, int multiplicative, const int* guard) {
  hipLaunchKernelGGL(k_voxel_build_general, dim3(nblk(g.ncell, 256)), dim3(256), 0, s, P, c6, start, g.ncell, n, cell_voxel, vox, vox_cell, multiplicative, guard);
}
void linearize_general(hipStream_t s, const float4* P, const double* c6, int n, Pose T, Grid g, const int* cell_voxel, const double* vox, int noff,
                       int* corr_v, double* corr_M, int want_H, double* partials, int* ncorr_partials, double* out28, int* out_ncorr) {
  const int nb = linearize_blocks(n);
  hipLaunchKernelGGL(k_linearize_general, dim3(nb), dim3(LIN_T), 0, s, P, c6, n, T, g, cell_voxel, vox, noff, corr_v, corr_M, want_H, partials, ncorr_partials);
  hipLaunchKernelGGL(k_fold<kAccum>, dim3(kAccum + 1), dim3(WAVE), 0, s, partials, nb, out28, ncorr_partials, out_ncorr);
}
void unsort6(hipStream_t s, const double* c6, const float4* P, int n, double* out9) {
  hipLaunchKernelGGL(k_unsort6, dim3(nblk(n, 256)), dim3(256), 0, s, c6, P, n, out9);
}
void sort6(hipStream_t s, const double* in9, const float4* P, int n, double* c6) {
  hipLaunchKernelGGL(k_sort6, dim3(nblk(n, 256)), dim3(256), 0, s, in9, P, n, c6);
}
void compute_error(hipStream_t s, const float4* P, int n, Pose T, const double* vox, int noff, const int* corr_v,
                   const double* corr_M, double* partials, double* out1) {
  const int nb = linearize_blocks(n);
  hipLaunchKernelGGL(k_error, dim3(nb), dim3(LIN_T), 0, s, P, n, T, vox, noff, corr_v, corr_M, partials);
  hipLaunchKernelGGL(k_fold<1>, dim3(1), dim3(WAVE), 0, s, partials, nb, out1, (const int*)nullptr, (int*)nullptr);
}
void lm_try(hipStream_t s, double* out, const int* ncorr, LmIn in) {
  hipLaunchKernelGGL(k_lm_try, dim3(1), dim3(WAVE), 0, s, out, ncorr, in);
}
void compute_error_dev(hipStream_t s, const float4* P, int n, const double* Tdev, const double* vox, int noff, const int* corr_v,
                       const double* corr_M, double* partials, double* out1) {
  const int nb = linearize_blocks(n);
  hipLaunchKernelGGL(k_error_dev, dim3(nb), dim3(LIN_T), 0, s, P, n, Tdev, vox, noff, corr_v, corr_M, partials);
  hipLaunchKernelGGL(k_fold<1>, dim3(1), dim3(WAVE), 0, s, partials, nb, out1, (const int*)nullptr, (int*)nullptr);
}
// a map of at most this many points is scanned whole by the wave for a query its first cube does not settle (fitness_wave)
static int fitness_scan_all(int nt) { return nt > 0 && nt <= 32768 ? nt : 0; }
void lm_step(hipStream_t s, const float4* P, const double* nx, const double* ny, const double* nz, int n, Grid g, const int* cell_voxel,
             const double* vox, int noff, int* corr_v0, double* corr_M0, int* corr_v1, double* corr_M1, double* partials, LmState* st,
             int j, const LmInit* open, const int* nvox, const void* segs_t, const void* segs_s, LmState* h_post, int seq, const float4* TP,
             const int* tstart, double* fit_partials, int nt, const int* lazy_need, int lazy_stamp, const int* lazy_counts, LmEarly* h_early) {
  const FitArgs fa{TP, tstart, fit_partials, fitness_scan_all(nt), (TP && tstart && fit_partials) ? 1 : 0, lazy_need, lazy_stamp, lazy_counts, h_early};
  hipLaunchKernelGGL(k_lm_step, dim3(linearize_blocks(n)), dim3(LIN_T), 0, s, P, nx, ny, nz, n, g, cell_voxel, vox, noff, corr_v0, corr_M0, corr_v1,
                     corr_M1, partials, st, j, (j == 0 && open) ? *open : LmInit{}, nvox, (const int*)segs_t, (const int*)segs_s, h_post, seq, fa);
}
void fitness_lm(hipStream_t s, const float4* SP, int ns, LmState* st, const float4* TP, const int* tstart, Grid g, double* partials, LmState* h_post,
                int seq, int nt) {
  hipLaunchKernelGGL(k_fitness_lm, dim3(fitness_blocks(ns)), dim3(FIT_T), 0, s, SP, ns, st, TP, tstart, g, partials, h_post, seq, fitness_scan_all(nt));
}
void fitness(hipStream_t s, const float4* SP, int ns, PoseF T, const float4* TP, const int* tstart, Grid g, double* partials, double* out1, int nt) {
  const int nb = fitness_blocks(ns);
  hipLaunchKernelGGL(k_fitness, dim3(nb), dim3(FIT_T), 0, s, SP, ns, T, TP, tstart, g, partials, fitness_scan_all(nt));
  hipLaunchKernelGGL(k_fold<1>, dim3(1), dim3(WAVE), 0, s, partials, nb, out1, (const int*)nullptr, (int*)nullptr);
}
void icp_accumulate(hipStream_t s, const float4* SP, int ns, const float4* TP, const int* tstart, Grid g, double max_dist, double* partials,
                    double* out28) {
  const int nb = linearize_blocks(ns);
  hipLaunchKernelGGL(k_icp_accumulate, dim3(nb), dim3(LIN_T), 0, s, SP, ns, TP, tstart, g, max_dist, max_dist * max_dist, partials);
  hipLaunchKernelGGL(k_fold<kAccum>, dim3(kAccum), dim3(WAVE), 0, s, partials, nb, out28, (const int*)nullptr, (int*)nullptr);
}
void transform_f32(hipStream_t s, const float* in, int stride_f, int n, PoseF T, float* out, int out_stride_f) {
  hipLaunchKernelGGL(k_transform_f32, dim3(nblk(n, 256)), dim3(256), 0, s, in, stride_f, n, T, out, out_stride_f);
}
void unsort3(hipStream_t s, const double* a, const double* b, const double* c, const float4* P, int n, double* out3) {
  hipLaunchKernelGGL(k_unsort3, dim3(nblk(n, 256)), dim3(256), 0, s, a, b, c, P, n, out3);
}
void sort3(hipStream_t s, const double* in3, const float4* P, int n, double* a, double* b, double* c) {
  hipLaunchKernelGGL(k_sort3, dim3(nblk(n, 256)), dim3(256), 0, s, in3, P, n, a, b, c);
}

}  // namespace rgck
