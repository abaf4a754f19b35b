// rgc_kernels.h -- launch wrappers of the gfx950 kernels (internal; the public boundary is include/rgc_hip.h)
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>

namespace rgck {

// Dense voxel-aligned grid: cell c (per axis) covers [(c + minc + 0.5) * res, (c + minc + 1.5) * res): this is fast_gicp's
// voxel_coord = floor(x / res - 0.5) (fast_vgicp_voxel.hpp:158-160) shifted by minc, so the exact-kNN search grid and the Gaussian
// voxel map of the target are the SAME partition.
struct Grid {
  int minc[3];
  int dim[3];
  double res;      // cell size
  int ncell;
  double inv_res;  // 1 / res when res is a power of two (x / res == x * inv_res bit for bit), else 0: see grid_inv_res()
};
inline double grid_inv_res(double res) {
  int e;
  return std::frexp(res, &e) == 0.5 ? 1.0 / res : 0.0;
}
inline Grid make_grid(const int minc[3], const int dim[3], double res) {
  Grid g{};
  double nc = 1.0;
  for (int a = 0; a < 3; a++) { g.minc[a] = minc[a]; g.dim[a] = dim[a]; nc *= (double)dim[a]; }
  g.res = res;
  g.inv_res = grid_inv_res(res);
  g.ncell = nc <= 2.0e9 ? (int)nc : -1;
  return g;
}

struct Pose {  // row-major rotation + translation, fp64 (Eigen::Isometry3d in the reference)
  double R[9];
  double t[3];
};
struct PoseF {  // fp32 4x4 rows 0..2 (pcl::transformPointCloud in the reference)
  float m[12];
};

struct Quat { double x, y, z, w; };
// B9 folded into the counting pass of a cloud's preparation (count_cells): point i is read from src, re-expressed as q * p + t with
// k_transform_q's arithmetic, stored to the cloud's own buffer (x, y, z, intensity) and counted -- one pass over the map less
// copy / epoch / frame (nullable): the library's own copy of the source map, compared with it bit for bit (x, y, z) on the way; a point that
// differs is taken over and *epoch = frame says "the map is not the one the neighbour lists were made for" (KnnCache)
// force: the copy is not to be trusted (first frame of a buffer): *epoch = frame whatever the comparison says
struct Reframe { const float* src; int src_stride_f; Quat q; double t[3]; float4* copy; int* epoch; int frame; int force; };
// The neighbour-list cache of a map that is handed over by rgc_set_target_reframed again and again (round 5, second half).  A rigid
// re-expression does not change who a point's k nearest neighbours are -- only the fp32 rounding of the coordinates can, and only where
// the k-th and the (k+1)-th neighbour (or the edge of what the 3x3x3 block proves) are that close.  Points are named by their RANK: the
// position in the sorted array of the frame that built the lists (neighbours have nearby ranks in every later frame too: look-ups by rank
// are as local as look-ups by position).  Per rank: the k neighbours' ranks as that frame's exact search found them (nbr, 4 k bytes), the
// top bit of the first one a CERTIFICATE: the gap behind the
// k-th neighbour exceeded what rounding in any two frames can bridge (list_certified).  A certified query of an unchanged map takes its
// neighbours from the list (knn_point_cached: no search); the others -- on the todo lists since the frame that built the lists -- are
// searched exactly as before.  "Unchanged" is verified, not assumed: the counting pass compares the map with the library's own copy, bit
// for bit (Reframe); a frame that finds a difference searches everything and builds the lists again.
struct KnnCache {
  int* nbr = nullptr;        // [n][k] ranks of the k nearest neighbours of the point of rank r, in no particular order
  int* rank_of = nullptr;    // [n] rank of original point o (written by the frame that builds the lists, k_rank_gather)
  int* pos_of = nullptr;     // [n] this frame's position of rank r in the sorted array (k_rank_gather)
  int* qrank = nullptr;      // [n] ... and the rank of the point at position i
  const int* epoch = nullptr;  // *epoch == frame: the map changed this frame (or the lists are not trusted)
  int* overflow = nullptr;   // overflow[(frame - 1) & 1] == frame - 1: the previous frame's rebuild ran out of room in a todo list: rebuild again
  int frame = 0;
  int* todo = nullptr;       // kTodoLists lists of ranks of the queries without a certificate, todo_cap entries each
  int* todo_cnt = nullptr;   // ... their lengths (emptied by k_place in a frame that rebuilds)
  int todo_cap = 0;
  float cert_slack = 0.f;    // metres: what the coordinates' fp32 rounding in two frames can move a distance by, twice (4 sqrt(3) ulp of the largest coordinate)
};
constexpr int kTodoLists = 64;
struct LmIn { double x0[16]; double lambda; double init_factor; };
struct LmState {  // device-resident state of LsqRegistration::computeTransformation (lsq_registration_impl.hpp:53-172)
  double x0[16], lambda, nu, y0, yi, H[36], b[6], d[6], delta[16], xi[16], Hfin[36];
  double rot_eps, trans_eps, init_factor;
  int phase, done, conv, failed, outer, inner, n_lin, n_err, ncorr, ticketA, ticketB, max_outer, max_inner, has_fit;
  double fit_sum;                // sum of squared NN distances at the final pose (k_fitness_lm)
  int nvox, def_t, def_s, pad;   // frame counters carried home with the state; pad = grid guards, map | scan << 8
  int gen, cmd, mode, cur;       // step kernels: mode, valid correspondence buffer; gen: the posted solve's number (host image); cmd, ticketA: unused
  float src_sq; int pad2;        // sum over the scan's grid cells of count^2 (how crowded its cells are; steers the scan's cell size); pad2: the lazy target's miss flag
  int lazy_nq, lazy_ncell;       // lazy target: listed queries / cells of this frame (size the next frame's launches)
};
struct LmInit { double x0[16], rot_eps, trans_eps, init_factor; int max_outer, max_inner; };
// Mapped host memory: the final pose of a solve, posted by its deciding launch BEFORE that launch computes the score (C8) -- a caller that
// needs only the pose to go on (the next frame's target of a dependent sequence, rgc_align_end_reframe) starts ~25 us earlier; pad / pad2:
// the grid guards and the lazy target's miss flag as the finished state will carry them; gen: the solve's number, written last
struct LmEarly { double x0[16]; int pad, pad2, outer, gen; };
struct FeParams { int n_scans; double min_range, max_range; };
struct LeafGrid { int minb[3]; int div[3]; };  // pcl::VoxelGrid leaf grid

constexpr int kAccum = 28;  // 21 upper-triangular H + 6 b + 1 cost
constexpr int kVoxRec = 10; // mean(3) cov6(6) num(1), doubles

// Sorted points are float4 {x, y, z, original index (int bits)} grouped by grid cell.
// ---- grid build ----
void bbox(hipStream_t s, const float* in, int stride_f, int n, double res, int* mm6, int* flags, int hi = 0);
// rf (nullable): in[] is first WRITTEN from rf->src (see Reframe; in must then be a 16-byte-stride device buffer this call may write)
void count_cells(hipStream_t s, const float* in, int stride_f, int n, Grid g, int* cell_of, int* slot_of, int* cnt, int hi = 0, int* guard = nullptr,
                 const Reframe* rf = nullptr);
// cnt: n = cells + 1 entries; block_sums: >= 8 * (n / 2048 + 2) bytes; cell_voxel (n - 1 ints) and nvox may be null
// consumes the counters: cnt[0..n) is left ZERO
void scan_cells(hipStream_t s, int* cnt, int* start, int n, void* block_sums, int* cell_voxel, int* nvox, int hi = 0,
                float* sum_sq = nullptr /* += sum of count^2, nullable */);
// order_tmp: n 64-bit records {original index, position in the cell, cell population} (k_place)
void place(hipStream_t s, int n, const int* cell_of, const int* slot_of, const int* start, unsigned long long* order_tmp, int hi = 0,
           const KnnCache* cache = nullptr /* its todo lists are emptied on the way in a frame that rebuilds them */);
void exclusive_scan(hipStream_t s, const int* in, int* out, int n, int* block_sums /* >= n/2048+2 */, int hi = 0);
// cache (nullable): KnnCache::rank_of (a frame that builds the lists) or pos_of and qrank (any other), written on the way
void rank_gather(hipStream_t s, const float* in, int stride_f, int n, const int* cell_of, const int* start,
                 const unsigned long long* order_tmp, float4* P, int* zero_me = nullptr, int hi = 0, const KnnCache* cache = nullptr);
// ---- C2: exact kNN + PLANE covariance -> unit normal ----
// bulk kernel (one lane per query; defers what it cannot finish) then the cooperative kernel (one wave per deferred query).
// deferred: deferred_bytes(n) bytes, whose first int (the count) must be 0 on entry (rank_gather's zero_me)
size_t deferred_bytes(int n);
// Seeds of the map's search (knn_point_seeded): seed[original point index] = an upper bound of the k-th squared distance the last search
// of that point found (>= 1e30: none); slack = how much a k-th distance may have grown since, in metres; warm = a search has written
// them.  Results do not depend on the seeds' values (a search they do not serve runs again without them).
struct KnnSeeds {
  float* seed = nullptr;
  float slack = 0.f;
  bool warm = false;
  KnnCache cache;  // (nbr == nullptr: no lists)
};
bool knn_seeds_apply(int n, int k);  // the seeded search exists for this cloud size and k
// wide_r = 2: the four-lanes-per-query search on the 5^3 block whatever the cloud (a sparse map)
void knn_bulk(hipStream_t s, bool is_target, const float4* P, const int* start, Grid g, int n, int k, const void* deferred, double* nx,
              double* ny, double* nz, const int* guard = nullptr, int wide_r = 0,
              hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr /* dense-map launch only (knn_bulk_times_itself): the launch's own start / stop times */,
              // lazy target (dense-map launch only): the queries listed in qlist[0 .. *nq) are searched, nothing else; q_est sizes the launch
              const int* qlist = nullptr, const int* nq = nullptr, int q_est = 0, const KnnSeeds& seeds = KnnSeeds{},
              // the scan's launch only: > 0 = that many waves at the end of the launch resolve the deferred queries as they are published
              // (no knn_coop launch behind it); the deferred buffer's entry words must hold kDeferredSlotEmptyByte bytes on entry
              int stream_coop_waves = 0);
constexpr int kDeferredSlotEmptyByte = 0x80;
// lazy target: stamp the cells of grid g within `margin` cells of the cell each point of the cloud falls into under T and list the occupied
// ones (cell_list: their first sorted point; qlist: all their points; counts[0] / [1]: the lists' sizes, zeroed by rank_gather)
// guard (nullable): the speculative grid's flag -- set: the map's points may lie outside the grid, nothing is listed
void footprint(hipStream_t s, const float* in, int stride_f, int n, Pose T, Grid g, int* need, int stamp, int margin, const float4* P, int n_map,
               const int* start, int* qlist, int* cell_list, int* counts, const int* guard = nullptr);
// ... and the voxel pass over the listed cells, the map's deferred queries resolved beside it (k_voxel_cells_coop); voxel_patch follows
void voxel_cells_coop(hipStream_t s, const float4* P, double* nx, double* ny, double* nz, const int* start, Grid g, int n, const int* cell_voxel,
                      double* vox, int* vox_cell, int k, const void* deferred, const int* guard, int waves, const int* cell_list, const int* ncells,
                      int cells_est, const KnnSeeds& seeds = KnnSeeds{});
bool knn_bulk_times_itself(bool is_target, int wide_r);
// waves: one-wave workgroups that share the deferred list (clamped to [32, 8192])
void knn_coop(hipStream_t s, bool is_target, const float4* P, const int* start, Grid g, int n, int k, const void* deferred, double* nx,
              double* ny, double* nz, const int* guard, int waves, const KnnSeeds& seeds = KnnSeeds{});
// ---- C3: Gaussian voxel map ----
void voxel_build(hipStream_t s, const float4* P, const double* nx, const double* ny, const double* nz, const int* start, Grid g,
                 int n, const int* cell_voxel, double* vox, int* vox_cell);
// voxel_build and knn_coop (target) in one launch, followed by voxel_patch: see k_voxel_build_coop
void voxel_build_coop(hipStream_t s, const float4* P, double* nx, double* ny, double* nz, const int* start, Grid g, int n, const int* cell_voxel,
                      double* vox, int* vox_cell, int k, const void* deferred, const int* guard, int waves, const KnnSeeds& seeds = KnnSeeds{});
// the voxels that hold a deferred query, recomputed: lets the cooperative search run BESIDE voxel_build; lanes: about
// the number of deferred queries (grid-stride loop)
void voxel_patch(hipStream_t s, const float4* P, const double* nx, const double* ny, const double* nz, const int* start, Grid g, const void* deferred,
                 const int* cell_voxel, double* vox, int lanes, hipEvent_t done = nullptr /* signalled by the launch's own completion */);
// ---- C4/C5/C6 ----
void linearize(hipStream_t s, const float4* P, const double* nx, const double* ny, const double* nz, int n, Pose T, Grid g,
               const int* cell_voxel, const double* vox, int noff, int* corr_v, double* corr_M, int want_H, double* partials,
               int* ncorr_partials, double* out28, int* out_ncorr);
// ---- the general covariance route (rgc_set_regularization_method other than PLANE, VoxelAccumulationMode::MULTIPLICATIVE): every point
// through the cooperative search, a regularised 3x3 per point (c6: six doubles, SoA c6[a * n + i], sorted order) instead of a unit normal;
// unoptimised by design.  method = rgc_regularization_method; guard (nullable): a tripped speculative-grid guard makes the kernels stand still
void knn_cov6(hipStream_t s, const float4* P, const int* start, Grid g, int n, int k, int method, double* c6, const int* guard);
void voxel_build_general(hipStream_t s, const float4* P, const double* c6, const int* start, Grid g, int n, const int* cell_voxel, double* vox,
                         int* vox_cell, int multiplicative, const int* guard);
void linearize_general(hipStream_t s, const float4* P, const double* c6, int n, Pose T, Grid g, const int* cell_voxel, const double* vox, int noff,
                       int* corr_v, double* corr_M, int want_H, double* partials, int* ncorr_partials, double* out28, int* out_ncorr);
void unsort6(hipStream_t s, const double* c6, const float4* P, int n, double* out9);
void sort6(hipStream_t s, const double* in9, const float4* P, int n, double* c6);
void compute_error(hipStream_t s, const float4* P, int n, Pose T, const double* vox, int noff, const int* corr_v,
                   const double* corr_M, double* partials, double* out1);
void lm_try(hipStream_t s, double* out, const int* ncorr, LmIn in);
void compute_error_dev(hipStream_t s, const float4* P, int n, const double* Tdev, const double* vox, int noff, const int* corr_v,
                       const double* corr_M, double* partials, double* out1);
// launch number j of a solve of the device-chained LM (see k_lm_step; 0 opens the solve and takes *open); corr_*0 / corr_*1 are the two
// correspondence buffers, the finished state's `cur` the valid one.  st: the 4096-byte LM area, zeroed once -- two state images (launch j
// reads image (j - 1) & 1 and leaves image j & 1: lm_image(st, j_last) is the latest) and the loose words behind them.  partials:
// 2 * linearize_blocks(n) rows of kAccum + 2 doubles (the launches alternate between the halves).
inline LmState* lm_image(LmState* st, int j_last) { return st + (j_last & 1); }
void lm_step(hipStream_t s, const float4* P, const double* nx, const double* ny, const double* nz, int n, Grid g, const int* cell_voxel,
             const double* vox, int noff, int* corr_v0, double* corr_M0, int* corr_v1, double* corr_M1, double* partials, LmState* st,
             int j, const LmInit* open, const int* nvox, const void* segs_t, const void* segs_s,
             LmState* h_post = nullptr /* mapped host memory: a finished state is posted there, then seq in its `gen` */,
             int seq = 0 /* > 0: post when done; < 0: post when done AND scored (by this kernel or fitness_lm) */,
             // the fitness score chained to the solve (all non-null): the launch whose decision ends the solve scores the final pose (as does a
             // launch on a finished solve without a score); TP / tstart: the map's sorted points and cell starts, nt its point count
             const float4* TP = nullptr, const int* tstart = nullptr, double* fit_partials = nullptr, int nt = 0,
             // lazy target: the target is built for the cells stamped lazy_stamp in lazy_need[] only -- a look-up of any other occupied voxel
             // raises LmState::pad2; lazy_counts: the lists' sizes, carried home in LmState::lazy_nq / lazy_ncell
             const int* lazy_need = nullptr, int lazy_stamp = 0, const int* lazy_counts = nullptr,
             LmEarly* h_early = nullptr /* mapped host memory (nullable): the deciding launch posts the final pose there before it scores it */);
// nt: the target's point count (a small map is scanned whole by the wave for a query its first cube does not settle; 0: never)
void fitness_lm(hipStream_t s, const float4* SP, int ns, LmState* st, const float4* TP, const int* tstart, Grid g, double* partials,
                LmState* h_post = nullptr, int seq = 0, int nt = 0);
// ---- f1: mapping-node feature registration (RGC_mapping.cpp:1069-1358) ----
// factor record = 8 doubles per feature: edge {a[3], b[3], var, valid}, plane {n[3], d, 0, 0, var, valid}
struct MapregAssoc {  // one association loop: feature set (n x 4: x,y,z,weight), its pose, the map grid it is matched against
  const float* feat; int n; int edge;
  Quat q; double t[3];
  const float4* P; const int* start; Grid g;
  double* fac; int* nvalid;  // nvalid nullable: += factors created
};
void mapreg_associate(hipStream_t s, const MapregAssoc* sets, int nsets /* <= 4, one launch */);
int mapreg_blocks(int ne, int np);
// both poses in one launch.  feat/fac/nfeat: {corner cur, surf cur, corner last, surf last}; x14 = q_cur t_cur q_last t_last;
// out56 = per pose {21 upper-triangular H, 6 g, robust cost} (H, g only if want_H); partials: 2 * 28 * max mapreg_blocks doubles
void mapreg_terms(hipStream_t s, const float* const feat[4], const double* const fac[4], const int nfeat[4], const double x14[14], double huber_a,
                  int want_H, double* partials, double* out56);
// ---- f4: one ICP iteration (1-NN correspondences within max_dist + the sums of the rigid fit); out28[0..16] = n, sum p, sum q, sum p q^T, sum d^2
void icp_accumulate(hipStream_t s, const float4* SP, int ns, const float4* TP, const int* tstart, Grid g, double max_dist, double* partials,
                    double* out28);
// ---- C8 ----
void fitness(hipStream_t s, const float4* SP, int ns, PoseF T, const float4* TP, const int* tstart, Grid g, double* partials,
             double* out1, int nt = 0);
// ---- misc ----
void transform_f32(hipStream_t s, const float* in, int stride_f, int n, PoseF T, float* out, int out_stride_f);
void unsort3(hipStream_t s, const double* a, const double* b, const double* c, const float4* P, int n, double* out3);
void sort3(hipStream_t s, const double* in3, const float4* P, int n, double* a, double* b, double* c);
int  linearize_blocks(int n);
int  fitness_blocks(int n);
// ---- B2 / B3 / B9 (rgc_pre.hip) ----
// ---- f3: sensor_msgs/PointCloud2 <-> device arrays; fields in the order x, y, z, intensity, ring, time (offset < 0: absent) ----
struct Pc2Layout { int point_step; int off[6]; int type[6]; int big_endian; };
void pc2_unpack(hipStream_t s, const unsigned char* data, int n, const Pc2Layout& L, float4* xyzi, int* ring, float* time);
void pc2_pack(hipStream_t s, const float* in, int cols, int n, int kind, unsigned char* out);
void deskew(hipStream_t s, float* xyzi, int stride_f, int n, Quat qinv, const double t[3]);
void transform_q(hipStream_t s, const float* in, int stride_f, int n, Quat q, const double t[3], float* out, int ostride_f);
void vg_bbox(hipStream_t s, const float* in, int stride_f, int n, float inv, int* mm6, int* flags);
// sparse leaf grids: counting sort over (y, z) rows, rank by (leaf x, index) inside a row -- the whole filter as one chain of launches.
// edge > 0: g is a box kept from an earlier cloud (see rgc_pre.hip).  res[0] <- flags of this run (1 non-finite point, 2 point outside g,
// 4 point within `edge` leaves of g's faces), res[1] = the live flag word (must be 0 on entry, is 0 on exit), res[2] <- number of leaves.
// dense != 0: the counting sort runs over the leaves themselves ("rows" below = leaves; dense clouds).
// row_block_sums: 8 bytes x (rows / 2048 + 2); head_block_sums: n / 2048 + 2 ints; cnt: rows + 1 zeros, left at zero.
// seg_shift: the counting sort's bucket is 2^seg_shift leaves of a grid row (0: the leaf itself, 31: the whole row); cnt / start: rows x vg_segments() + 1 entries
int vg_segments(const LeafGrid& g, int seg_shift);
void vg_rows(hipStream_t s, const float* in, int stride_f, int n, float inv, LeafGrid g, int edge, int seg_shift, int* row_of, int* lx, int* slot_then_pos, int* cnt,
             int* start, void* row_block_sums, unsigned long long* tmp, int* order, unsigned long long* leaf, int* head_block_sums, float* out,
             int* res);

// ---- A1-A8 front-end (rgc_frontend.hip) ----
int fe_blocks(int n);
int fe_slot_ints();
void fe_filter(hipStream_t s, const float* in, int stride_f, int n, FeParams p, int* ring, int* st, int* rank_in_block, int* blk_hist);  // + per-block ring ranks / histograms
void fe_half(hipStream_t s, const float* in, int stride_f, int n, const int* ring, int* st);
void fe_bucket(hipStream_t s, const float* in, int stride_f, int n, int NS, const int* ring, int* rank_in_block, int* blk_hist, int* meta,
               const int* st, float4* C, int* inum2, int* z0, int* z1, int* z2, int* z3);  // z0..z3: four per-point arrays zeroed on the way (n ints each)
// cs: the launch's bound; csp (nullable): the sweep's size on the device (meta[128]), read by the kernels -- the host need not know it
void fe_stencils(hipStream_t s, const float4* C, int cs, const int* csp, float* range_vec, float* scan_angle, const int* inum2, int* inum, float* curv,
                 float* curv2, float* icurv, float* dsrc, float* osrc, int* picked);
void fe_ground(hipStream_t s, const float4* C, int cs, const int* csp, int NS, const float* range_vec, const int* meta, int* gmark, int* mult, int* seedcnt,
               double* partials, double* out11, double* fit);   // ... the ground sums into out11 and their plane fit into fit[16], on the device
// the distance sums into out2[2]
void fe_ground_dist(hipStream_t s, const float4* C, int cs, const int* csp, const int* mult, const double* fit, double* partials, double* out2);
void fe_ground_list(hipStream_t s, const float4* C, int cs, const int* csp, int NS, const float* range_vec, const int* meta, const int* seedcnt,
                    const int* seedpos, float4* out, int cap);
void fe_select(hipStream_t s, const float4* C, int NS, const int* meta, const float* curv, const float* curv2, const float* icurv, const int* inum,
               const int* gmark, int* picked, int* ipicked, int* label, int* ilabel, int* slots, int* flags, int max_ring /* points in the largest ring: sizes the LDS window */,
               int* sorted_curv, int* sorted_icurv /* cs ints each: per-sector sorted point indices (scratch) */);
void fe_emit(hipStream_t s, const float4* C, int NS, const int* slots, const float* dsrc, const float* osrc, float* sharp, float* flat, float* inten,
             int cap, int* counts);

#ifdef RGC_LAB_TURN
void lab_turn(unsigned long long* out8);
#endif
#if defined(RGC_LAB) || defined(RGC_LAB_BLK)
void lab_blocks(long long* out65536);  // developer build: {start, end, XCC, first query} of every workgroup of the map's bulk kNN launch
#endif
#ifdef RGC_LAB
void lab_lm_ts(unsigned long long* out16, hipStream_t s);  // developer build: phase timestamps of k_lm_step
void lab_why(int* out8);
void lab_declines(int* out16);                              // developer build: why knn_point_seeded declined, per lane (and resets)
void lab_iters(unsigned long long* out8);                   // developer build: wave-level loop counts of the map's bulk kNN kernel (and resets them)
void lab_wave_ts(long long* out16384, hipStream_t s);       // developer build: start / end of the scan kNN launch's waves
#endif
}  // namespace rgck
