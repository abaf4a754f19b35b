// rgc_api.hip -- host side of librgc_hip.so: context, device buffers, the LM driver and the C-ABI of
// include/rgc_hip.h.  The per-point work is in rgc_kernels.hip; this file holds the scalar control flow the
// reference runs in LsqRegistration (lsq_registration_impl.hpp:53-172) and the buffer plumbing.
//
// No CPU fallback exists: every entry point fails with RGC_ERR_HIP when the HIP runtime / device is missing.
// Reference citations are relative to /root/reference/rgc_slam/.
#include "../../include/rgc_hip.h"

#include <hip/hip_runtime.h>

#include <cfloat>
#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <chrono>
#include <mutex>
#include <new>
#include <unordered_set>
#include <vector>

#include "rgc_kernels.h"
#include "rgc_lm.h"

namespace {

constexpr int kMaxK = 32;
// Events that only order streams of ONE device against each other: no timing, and a device-scope release when recorded (the default is a
// system-scope one -- the L2s written back so that the HOST may read what came before; nobody's host does behind these).
constexpr unsigned kDevEvent = hipEventDisableTiming | hipEventReleaseToDevice;
// Build-time switches (RGC_EXTRA_FLAGS=-D...): alternative routes to the SAME results, kept for A/B measurements.  A caller's
// process reads RGC_LM_IMPL, RGC_SPEC_GRID, RGC_KNN_SEEDS, RGC_KNN_CACHE (a context's initial rgc_set_knn_reuse mode), RGC_TRACE_ALLOC, RGC_TRACE_CACHE, RGC_CHECK_POINTERS,
// RGC_FORCE_GENERAL (the odometer's settings on the general covariance route: a cross-check) and the
// three scheduling switches RGC_JOIN_SPIN_US / RGC_PREP_EVENT_EXT / RGC_COOP_STREAM from the environment, once, in rgc_create.
#ifndef RGC_LM_POST
#define RGC_LM_POST 1          // 0: rgc_align_end always waits for the stream and its copy of the state (round 2)
#endif
#ifndef RGC_SMALL_COPY
#define RGC_SMALL_COPY 0       // 1: the 32-byte host-to-device copy in front of every preparation
#endif
#ifndef RGC_FE_SPEC
#define RGC_FE_SPEC 1          // 0: the front-end reads every sweep's size back before its stencil kernels
#endif
#ifndef RGC_SOLVE_BEHIND_MAP
#define RGC_SOLVE_BEHIND_MAP 1 // 0: the solve always on the scan's (high-priority) stream (round 2)
#endif
#ifndef RGC_LM_SPARE_ASIDE
#define RGC_LM_SPARE_ASIDE 1   // 0: a solve's spare step launches stay on its own stream, in front of whatever comes next there (round 3)
#endif
#ifndef RGC_MARK_BEHIND_COUNT
#define RGC_MARK_BEHIND_COUNT 0  // 1: the map's stream mark is recorded BEHIND its counting pass (round 4).  An event record between two kernels of one
                                 // stream holds the second one back ~5.8 us (the timeline of round 5); in front of the frame's first launch the record is
                                 // processed while the GPU waits for the host anyway: two contexts steady 0.340 -> 0.334 ms per frame
#endif
#ifndef RGC_KNN_SEEDS
#define RGC_KNN_SEEDS 1        // 0: the map's exact search never starts from the previous search's k-th distances (round 4)
#endif
#ifndef RGC_COOP_STREAM
#define RGC_COOP_STREAM 0      // 1: the scan's deferred queries are resolved by waiting waves at the end of its bulk kNN launch (coop_stream): a frame at a time
                               // 3 % faster, a sequence on two contexts 9 % slower (the waiting waves hold slots the other context's map wants); measured, off
#endif
#ifndef RGC_PREP_EVENT_EXT
#define RGC_PREP_EVENT_EXT 1
#endif
#ifndef RGC_JOIN_SPIN_US
#define RGC_JOIN_SPIN_US 300
#endif
#ifndef RGC_EARLY_POSE
#define RGC_EARLY_POSE 1       // 0: rgc_align_end_reframe waits for a solve's score before it enqueues the next frame's target (round 5)
#endif
#ifndef RGC_KNN_CACHE
#define RGC_KNN_CACHE 1        // 0: no neighbour lists (rgck::KnnCache): every frame searches the whole map, seeded
#endif
#ifndef RGC_MAP_WIDE_R
#define RGC_MAP_WIDE_R 2       // block radius of the bulk kNN launch for a sparse map (0 = off, 2)
#endif
#ifndef RGC_MAP_WIDE
#define RGC_MAP_WIDE 0.25      // ... taken when the map has fewer points per grid cell than this
#endif
#ifndef RGC_SRC_RES
#define RGC_SRC_RES 0.0        // fixed cell size of the scan's kNN grid (0 = adaptive)
#endif
static_assert(RGC_MAP_WIDE_R == 0 || RGC_MAP_WIDE_R == 2, "RGC_MAP_WIDE_R: 0 or 2");
constexpr int kProfKinds = RGC_K_COUNT;

struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  bool borrowed = false;  // p belongs to another context (rgc_share_target): never freed, never grown here
};

struct Cloud {
  // input (device copy owned by ctx, or caller's device pointer)
  const float* in = nullptr;
  int stride_f = 0;
  int n = 0;
  bool ready = false;  // grid + normals (+ voxels for the target) enqueued
  bool covs_user = false;  // the normals were given by the caller (rgc_set_source/target_covariances), not computed from the neighbours
  DevBuf in_copy, cell_of, slot_of, cnt, start, block_sums, order_tmp, P, nx, ny, nz;  // P: sorted float4 {x,y,z,orig idx}
  DevBuf c6;            // the general covariance route only (general_route()): six doubles per point, SoA, instead of the normal
  bool general = false; // this cloud was prepared on the general route (its covariances are in c6, nx / ny / nz hold nothing)
  DevBuf segs;  // deferred-query list of the bulk kNN kernel: [count, pad x15][query n][bound n]
  int deferred_seen = -1;  // deferred count of the last cloud whose count came home (sizes the next cooperative launch)
  rgck::Grid grid{};   // the search grid: sorted array P, start[]; for the target also the voxel grid cell_voxel[] is laid out on
  // speculative grid (voxel level): the previous cloud's grid, widened, re-used without the bounding-box round trip; k_count guards it
  rgck::Grid spec_grid{};
  bool spec_ok = false;    // spec_grid is usable
  bool spec_used = false;  // this cloud was prepared on spec_grid and its guard has not been read yet
  bool reframe_pending = false;  // in[] has not been written yet: the next preparation produces it from rf (rgc_set_target_reframed)
  rgck::Reframe rf{};
  size_t cnt_clean = 0;    // cnt[0 .. cnt_clean) is known to be zero (the cell scan leaves the counters it consumed at zero)
  const void* cnt_seen = nullptr;  // the allocation cnt_clean refers to
  // target only
  DevBuf cell_voxel, vox, vox_cell;
  int nvox = -1;
  // lazy target (rgc_set_target_lazy): 0 = covariances and voxel map complete; 1 = the grid is built, nothing else (the solve's guess
  // decides which part is needed); 2 = built for the cells stamped need_stamp in `need` only
  int lazy = 0;
  DevBuf need, qlist, cell_list;  // one stamp per grid cell; the listed queries (points) and cells of this frame (k_footprint)
  int need_stamp = 0;
  const void* need_seen = nullptr;  // the allocation the stamps refer to
  int lazy_nq_seen = -1, lazy_ncell_seen = -1;  // the previous frame's list sizes (they size this frame's launches)
  // seeds of the exact search (rgck::KnnSeeds): kept while the target is a re-expression of the SAME buffer (rgc_set_target_reframed: the
  // key is the buffer it re-frames), one float per original point
  DevBuf seed;
  const void* seed_key = nullptr;
  int seed_n = 0;
  bool seed_on = false;    // this cloud's searches read and write them
  bool seed_warm = false;  // ... and some search has written them
  float seed_slack = 0.f;
  // the neighbour-list cache on top of the seeds (rgck::KnnCache): the same key, the same life
  DevBuf nbr, rank_of, pos_of, qrank, map_copy, todo, cache_small;  // cache_small: kTodoLists list lengths, the epoch word, the overflow word
  bool cache_on = false;    // this preparation compares the map with map_copy and its searches read / write the lists
  bool cache_live = false;  // the LAST preparation's searches ran with the lists attached (otherwise they are stale: the next frame starts over)
  int cache_frame = 0;
  int cache_e2 = 0;         // binary exponent of the largest coordinate the certificates were issued for
  int cache_e2_low = 0;     // frames in a row whose coordinates stayed below it
  int cache_e2_low_max = 0; // ... and the largest exponent among them
  int todo_cap = 0;
  bool slots_clean = false;        // segs' entry words hold the "empty slot" pattern (the scan's deferred queries resolved inside its bulk launch)
  const void* slots_seen = nullptr;  // ... of this allocation
  bool prepared_recorded = false;  // the preparation's last launch carried the context's tgt_prepared event (no record packet behind it)
  bool cache_searched_lists = false;  // the last preparation's search was the seeded launch that reads the lists (rgc_stats::searched_target)
  int searched_known = -1;            // rgc_stats::searched_target of this preparation once it has been fetched (-1: not yet)
};

struct ProfRegion {
  hipEvent_t a, b;
  int kind;
  long long points;
};

}  // namespace

// Contexts alive in this process: a context that borrows another one's target (rgc_share_target) checks its owner here before every
// solve, so that an owner destroyed too early is an error message and not a read of freed memory.
static std::mutex g_live_mutex;
static std::unordered_set<const rgc_ctx*> g_live;
static std::atomic<unsigned long long> g_next_uid{1};  // contexts are told apart by this, not by their address (an address is re-used)
static bool ctx_alive(const rgc_ctx* c) {
  std::lock_guard<std::mutex> lk(g_live_mutex);
  return g_live.count(c) != 0;
}

struct rgc_ctx {
  int device = 0;
  rgc_params prm{};
  hipStream_t stream = nullptr;   // main stream: target preprocessing, LM loop, fitness, getters
  hipStream_t stream2 = nullptr;  // source preprocessing runs here, concurrently with the (much larger) target's
  hipEvent_t src_ready = nullptr; // recorded on stream2 after the source is prepared
  hipEvent_t tgt_ready = nullptr; // recorded on the main stream at rgc_align_begin: the solve (on stream2) waits for the map's preparation
  hipEvent_t main_mark = nullptr; // recorded on the main stream before a source is prepared: stream2 waits for it (producers on rgc_stream())
  bool src_pending = false;       // main stream has not yet been ordered after src_ready
  bool mark_valid = false, main_has_target_prep = false;  // main_mark recorded; a map preparation was enqueued after it and may still run
  bool main_late_producer = false;  // ... and something that may WRITE a scan buffer (rgc_upload) was enqueued on the main stream behind it
  char err[512] = {0};
  Cloud src, tgt;
  // per-correspondence state frozen by linearize (fast_vgicp_impl.hpp:104-115)
  DevBuf corr_v, corr_M, partials, ipartials;
  DevBuf corr_v2, corr_M2;    // second correspondence buffer of the chained LM (speculative linearisation); corr_v / corr_M
                              // always name the VALID one after a solve
  int corr_noff = 0, corr_n = 0;
  bool corr_valid = false;
  // small device scratch + pinned host mirrors
  int* d_small = nullptr;     // [0..5] target bbox, [6] flags, [7] nvox, [8] ncorr, [16..22] source bbox + flags
  double* d_out = nullptr;    // 28 doubles
  int* h_small = nullptr;     // pinned, same layout
  double* h_out = nullptr;    // pinned
  DevBuf scratch;             // getters
  DevBuf lm_state;            // device-chained LM state (rgck::LmState)
  // f1: mapping-node feature registration (corner / surf feature maps: grid only, 1.5 m cells)
  Cloud mr_map[2];
  DevBuf mr_feat[4], mr_fac[4], mr_partials, mr_small;
  bool deferred_known = false;  // stats.deferred_* are those of the current clouds (carried home by the last align)
  DevBuf fit_partials;        // fitness rows when it is chained behind the LM slots
  rgck::LmState* h_lm = nullptr;  // pinned mirror (the stream-ordered copy behind every batch of LM launches)
  rgck::LmState* h_post = nullptr; // mapped host memory the DEVICE writes a finished solve's state into, then the solve's number into its `gen`
  rgck::LmState* d_post = nullptr; // ... its device address
  rgck::LmEarly* h_early = nullptr, *d_early = nullptr;  // mapped host memory / its device address: a solve's final pose, posted before its score (rgc_align_end_reframe)
  int lazy_margin = 0;             // rgc_set_target_lazy: > 0 = the target's covariances / voxels are built only where the solve can look (cells of margin)
  hipEvent_t src_in_ready = nullptr;  // recorded on stream2 behind a HOST scan's upload: the lazy target's footprint pass (main stream) reads the scan's input
  bool src_in_pending = false;
  int lm_seq = 0;                  // number of the pending solve (1, 2, ...)
  int lm_j = 0;                    // launches enqueued for it so far (rgck::lm_step's launch number: the state image alternates with it)
  rgck::LmState lm_res{};          // the finished solve's state as rgc_align_end took it (from h_post or h_lm): nothing writes it asynchronously
  hipEvent_t lm_mid = nullptr;     // recorded on the solve's stream behind its expected launches: the spare ones, on the context's other stream, wait for it
  hipEvent_t lm_tail = nullptr;    // recorded behind every batch of LM launches (and its copy into h_lm) on the stream they went to
  hipStream_t lm_tail_stream = nullptr;  // ... that stream: a solve enqueued on the OTHER stream waits for lm_tail first
  bool post_on = RGC_LM_POST != 0; // (build flag) 0: always wait for the stream and its copy, as in round 2
  struct { bool active = false; bool want_fitness = false; float guess[16]; } pend;  // rgc_align_begin .. rgc_align_end
  struct { bool on = false; int rc = 0; float T[16]; double H[36]; double fitness = 0; int iterations = 0, converged = 0, lm_failed = 0; bool has_fit = false; } gen_res;  // general route: rgc_align_begin solves at once, rgc_align_end hands this over
  int lm_last_outer = 0;      // outer iterations of the previous solve: sizes the next blind batch
  bool small_copy_always = RGC_SMALL_COPY != 0;  // (build flag) 1: the 32-byte copy in front of every preparation, as before
  bool small_clean[2] = {false, false};  // d_small block of the map / the scan holds its initial image (the last solve's first step restored it)
  hipStream_t solve_stream = nullptr;  // where the pending solve was enqueued (rgc_align_begin)
  bool solve_behind_map = RGC_SOLVE_BEHIND_MAP != 0;  // (build flag) 0: the solve always on the scan's (high-priority) stream, as in round 2
  bool lm_host = false;       // RGC_LM_IMPL=host: host-driven LM loop over the public fine-seam kernels (cross-check of the device-chained one)
  bool spec_on = true;        // RGC_SPEC_GRID=0 turns the speculative grid off
  bool coop_stream_on = RGC_COOP_STREAM != 0;  // (build flag; RGC_COOP_STREAM in the environment) the scan's deferred queries inside its bulk kNN launch
  bool prep_event_ext = RGC_PREP_EVENT_EXT != 0;  // (build flag; RGC_PREP_EVENT_EXT in the environment) the map's last launch signals tgt_prepared itself
  int join_spin_us = RGC_JOIN_SPIN_US;  // (build flag; RGC_JOIN_SPIN_US in the environment) how long the host waits for an almost-ready scan instead of putting a barrier into the map's stream (join_source)
  bool cache_on = RGC_KNN_CACHE != 0;  // (build flag; RGC_KNN_CACHE=0 in the environment) the neighbour lists of an unchanged map on top of the seeds
  bool seeds_on = RGC_KNN_SEEDS != 0;  // (build flag; RGC_KNN_SEEDS=0 in the environment) 0: every search of a re-framed map starts without a bound, as before round 5
  int reg_method = RGC_REG_PLANE, voxel_mode = RGC_VOXEL_ADDITIVE;  // as selected by the caller, implemented or not (rgc_set_regularization_method)
  bool test_fail_cache_alloc = false;  // RGC_TEST_FAIL_CACHE_ALLOC in the environment (rgc_create)
  bool force_general = false;          // RGC_FORCE_GENERAL=1 in the environment (rgc_create): PLANE / ADDITIVE on the general route too (a test's cross-check of the two routes)
  bool cache_dropped = false;          // the lists' buffers did not fit on the device: the context went down to the seeds by itself (rgc_get_knn_reuse)
  bool check_ptrs = false;             // RGC_CHECK_POINTERS in the environment (rgc_create): every pointer a caller calls "device" is looked up before it is used (check_device_range)
  bool trace_cache = false;            // RGC_TRACE_CACHE in the environment (rgc_create): rgc_get_stats reports the lists' state on stderr
  double src_res = RGC_SRC_RES;  // (build flag) fixed cell size of the SCAN's kNN grid (only the map's grid must be the voxel grid); 0 = adaptive
  int map_wide_r = RGC_MAP_WIDE_R;          // (build flag; 0 = off, 2) block radius of the bulk kNN launch for a sparse map
  double map_wide_density = RGC_MAP_WIDE;   // (build flag) ... when the map has fewer points per grid cell than this
  double src_res_auto = 0.0;  // adaptive cell size of the scan's kNN grid, steered by how crowded its cells were in the previous frame (0 = voxel_res)
  Cloud aux;                  // grid scratch of rgc_voxelgrid
  DevBuf pre_in, pre_out, vg_order, vg_pos, vg_tmp, vg_leaf;  // B2/B3/B9 staging
  struct VgBox { float leaf = 0.f; bool valid = false; rgck::LeafGrid g{}; } vg_box[4];  // measured leaf boxes of earlier clouds, by leaf size
  int vg_box_next = 0;
  // Bounding boxes the library knows WITHOUT measuring: rgc_set_target_reframed maps the input's box (measured once per input buffer)
  // through the transform it applies -- the box of a sub-map re-framed by a new pose (RGC_odometer.cpp:1248-1256) follows from the
  // pose.  The target's preparation takes its grid from the hint: no bounding-box kernel, no host round trip, and no speculative-grid
  // miss when the re-framed map's box swings with the vehicle's yaw.  k_count's guard still checks it.
  struct BoxHint { const void* p = nullptr; int n = 0; double lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0}; double reach_xy = 0, reach_z = 0; } box_hint[4];
  int box_hint_next = 0;
  bool vg_flags_clean = false;  // d_small[24 + 6] is known to be zero (a finished rows chain leaves it so)
  // rgc_voxelgrid_begin / _end: one filter of a device cloud in flight (enqueued on its kept box, result not yet looked at)
  struct VgPending { bool active = false, ready = false; const float* d_in = nullptr; int n = 0, stride_bytes = 0; float leaf = 0.f; float* d_out = nullptr;
                     rgck::LeafGrid g{};  // the leaf grid the pending filter was enqueued on
                     int n_out = 0; } vg_pend;
  hipEvent_t vg_done = nullptr;
  int* h_vg = nullptr;  // pinned: the pending filter's three result ints (h_small's words are all taken: the front-end stages 16 ints at +32)
  DevBuf fe[34];              // front-end buffers
  unsigned char* h_stage = nullptr;  // pinned staging of the front-end's small read-backs and feature clouds (a copy into pageable
  size_t h_stage_cap = 0;            // memory is staged by the runtime anyway, one blocking hop per call)
  bool fe_spec_on = RGC_FE_SPEC != 0;  // (build flag) 0: read every sweep's size back before its stencil kernels
  int fe_last_ns = 0, fe_last_max_ring = 0;  // the previous sweep's scan lines and largest ring: sizes the next sweep's launches without a read-back
  int fe_n_cloud = 0;         // points of the last front-end's ring-major cloud (fe[5]), for rgc_frontend_cloud_device
  // f2: rolling local map.  World-frame points (relative to map_origin, x,y,z,intensity, 16 B) of the live keyframes as
  // contiguous segments in insertion order in map_store[map_cur]; the other buffer is the compaction / re-basing target.
  struct MapKf { int id; size_t off; int n; double t[3]; };
  DevBuf map_store[2], map_target;
  int map_cur = 0;
  size_t map_n = 0;
  std::vector<MapKf> map_kf;
  int map_next_id = 0;
  double map_origin[3] = {0, 0, 0};
  bool map_dirty = false;     // keyframes changed since the last commit
  bool map_bound = false;     // the context's target IS the committed map (rgc_set_target* unbinds it)
  unsigned long long tgt_generation = 0;   // bumped whenever this context prepares a target (what borrowers check)
  const rgc_ctx* tgt_owner = nullptr;      // rgc_share_target: whose target this context aliases, and at which generation
  unsigned long long tgt_owner_gen = 0, tgt_owner_uid = 0;
  unsigned long long uid = 0;              // process-wide, never re-used
  hipEvent_t tgt_prepared = nullptr;       // recorded on the main stream behind every target preparation (rgc_hold_source_until_target_of of another context waits for it)
  hipEvent_t src_read_done = nullptr;      // recorded on the main stream behind a kernel that reads the source's INPUT buffer (rgc_get_aligned*)
  bool src_read_pending = false;           // ... and not yet waited for by the stream a host source is copied on
  float map_leaf = 0.f;
  int map_ntarget = 0;
  unsigned long long map_rev = 0;
  rgc_stats stats{};
  // profiling
  bool prof_on = false;
  unsigned prof_mask = ~0u;
  std::vector<ProfRegion> prof_open;
  std::vector<hipEvent_t> ev_pool;
  long long prof_launches[kProfKinds] = {0};
  double prof_ms[kProfKinds] = {0};
  long long prof_points[kProfKinds] = {0};
};

namespace {

int fail(rgc_ctx* c, int code, const char* fmt, ...) {
  if (c) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(c->err, sizeof(c->err), fmt, ap);
    va_end(ap);
  }
  return code;
}

// The odometer's settings (PLANE, ADDITIVE: fast_gicp_impl.hpp:20, fast_vgicp_impl.hpp:24) run on the tuned kernels, which keep a point's
// covariance as the unit normal of I - 0.999 n n^T.  Every other RegularizationMethod, and VoxelAccumulationMode::MULTIPLICATIVE, runs on
// the GENERAL route: every point through the cooperative search with a regularised 3x3 per point (rgck::knn_cov6), a plain voxel pass, and
// the host-driven LM loop over a linearisation that takes the full source covariance.  Unoptimised; the same entry points, the same results
// as the reference's arithmetic for those settings (fast_gicp_impl.hpp:262-293, fast_vgicp_voxel.hpp:76-99).
bool general_route(const rgc_ctx* c) { return c->force_general || c->reg_method != RGC_REG_PLANE || c->voxel_mode == RGC_VOXEL_MULTIPLICATIVE; }
int check_supported(rgc_ctx*) { return RGC_OK; }   // (every value of both enums is implemented since round 6)

#define HIPCHK(c, expr)                                                                                    \
  do {                                                                                                     \
    hipError_t _e = (expr);                                                                                \
    if (_e != hipSuccess) {                                                                                \
      (void)hipGetLastError(); /* the runtime keeps a failed call's error until it is read: reported HERE, it must not surface again   \
                                  from the hipGetLastError() of the next, unrelated call (tests/fuzz/fuzz_bad_args.py) */             \
      return fail((c), RGC_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__);                       \
    }                                                                                                      \
  } while (0)

int ensure(rgc_ctx* c, DevBuf& b, size_t bytes) {
  if (b.borrowed) { b.p = nullptr; b.cap = 0; b.borrowed = false; }  // an alias is dropped, never resized: this context gets its own buffer
  if (bytes <= b.cap && b.p) return RGC_OK;
  if (b.p) {
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream2));
    HIPCHK(c, hipFree(b.p));
    b.p = nullptr;
    b.cap = 0;
  }
  // head-room: clouds of similar size arrive every frame, and a re-allocation costs a device synchronisation plus hipFree /
  // hipMalloc (hundreds of microseconds: 10 % of the first 20 frames of a sequence when buffers crept up by 1/8 at a time).
  // HBM is not the scarce resource here: half again for anything below 256 MB, an eighth above.
  size_t want = bytes + (bytes < ((size_t)256 << 20) ? bytes / 2 : bytes / 8) + 256;
  HIPCHK(c, hipMalloc(&b.p, want));
  b.cap = want;
  static const bool trace = getenv("RGC_TRACE_ALLOC") != nullptr;  // developer aid: which buffer grew, and when
  if (trace) fprintf(stderr, "[rgc] buffer at ctx+%ld grew to %zu bytes (asked %zu)\n", (long)((char*)&b - (char*)c), want, bytes);
  return RGC_OK;
}

// An integrator's aid (RGC_CHECK_POINTERS=1; off by default: a look-up per pointer per call, microseconds on a frame's critical path): is what
// the caller calls a device buffer one -- device memory of THIS context's device, with room for `bytes` behind p?  A host pointer handed to
// a *_device entry, a buffer of another GPU, a count larger than the allocation: RGC_ERR_INVALID instead of a memory fault on the device.
static int check_device_range(rgc_ctx* c, const void* p, size_t bytes, const char* what) {
  if (!c->check_ptrs || !p) return RGC_OK;
  hipPointerAttribute_t at;
  memset(&at, 0, sizeof(at));
  if (hipPointerGetAttributes(&at, p) != hipSuccess) {
    (void)hipGetLastError();
    return fail(c, RGC_ERR_INVALID, "%s: %p is not memory the HIP runtime knows (a host pointer passed as a device pointer?)", what, p);
  }
  if (at.type != hipMemoryTypeDevice) return fail(c, RGC_ERR_INVALID, "%s: %p is not device memory", what, p);
  if (at.device != c->device) return fail(c, RGC_ERR_INVALID, "%s: %p lives on device %d, the context on device %d", what, p, at.device, c->device);
  void* base = nullptr;
  size_t size = 0;
  if (hipMemGetAddressRange((hipDeviceptr_t*)&base, &size, (hipDeviceptr_t)const_cast<void*>(p)) != hipSuccess) { (void)hipGetLastError(); return RGC_OK; }
  const size_t off = (size_t)((const char*)p - (const char*)base);
  if (off + bytes > size) return fail(c, RGC_ERR_INVALID, "%s: %zu bytes asked of an allocation that has %zu behind %p", what, bytes, size - off, p);
  return RGC_OK;
}

void release(DevBuf& b) {
  if (b.p && !b.borrowed) (void)hipFree(b.p);
  b.p = nullptr;
  b.cap = 0;
  b.borrowed = false;
}

void release_cloud(Cloud& cl) {
  for (DevBuf* b : {&cl.in_copy, &cl.cell_of, &cl.slot_of, &cl.cnt, &cl.start, &cl.block_sums, &cl.order_tmp, &cl.P, &cl.nx, &cl.ny, &cl.nz, &cl.segs,
                    &cl.cell_voxel, &cl.vox, &cl.vox_cell, &cl.need, &cl.qlist, &cl.cell_list, &cl.seed, &cl.nbr, &cl.rank_of, &cl.pos_of, &cl.qrank,
                    &cl.map_copy, &cl.todo, &cl.cache_small, &cl.c6})
    release(*b);
}

// ---- profiling regions (HIP events on the context's stream) ----
struct ProfScope {
  rgc_ctx* c;
  bool on;
  ProfRegion r{};
  hipStream_t st;
  ProfScope(rgc_ctx* ctx, int kind, long long points, hipStream_t stream = nullptr) : c(ctx), on(ctx->prof_on && ((ctx->prof_mask >> kind) & 1u)), st(stream ? stream : ctx->stream) {
    if (!on) return;
    auto get = [&](hipEvent_t* e) {
      if (!c->ev_pool.empty()) { *e = c->ev_pool.back(); c->ev_pool.pop_back(); return true; }
      return hipEventCreate(e) == hipSuccess;
    };
    if (!get(&r.a)) { on = false; return; }
    if (!get(&r.b)) { c->ev_pool.push_back(r.a); on = false; return; }
    r.kind = kind;
    r.points = points;
    (void)hipEventRecord(r.a, st);
  }
  ~ProfScope() {
    if (!on) return;
    (void)hipEventRecord(r.b, st);
    c->prof_open.push_back(r);
  }
};

void prof_collect(rgc_ctx* c) {
  for (auto& r : c->prof_open) {
    float ms = 0.f;
    if (hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
      c->prof_launches[r.kind]++;
      c->prof_ms[r.kind] += (double)ms;
      c->prof_points[r.kind] += r.points;
    }
    c->ev_pool.push_back(r.a);
    c->ev_pool.push_back(r.b);
  }
  c->prof_open.clear();
}

int noff_of(int method) { return method == RGC_DIRECT1 ? 1 : (method == RGC_DIRECT7 ? 7 : 27); }

int check_params(rgc_ctx* c, const rgc_params* p) {
  if (!(p->voxel_res > 0.0) || !std::isfinite(p->voxel_res)) return fail(c, RGC_ERR_INVALID, "voxel_res must be > 0");
  if (p->k_correspondences < 2 || p->k_correspondences > kMaxK) return fail(c, RGC_ERR_INVALID, "k_correspondences must be in [2,%d]", kMaxK);
  if (p->neighbor_method < RGC_DIRECT27 || p->neighbor_method > RGC_DIRECT1) return fail(c, RGC_ERR_INVALID, "bad neighbor_method");
  if (p->max_iterations < 0 || p->lm_max_iterations < 1) return fail(c, RGC_ERR_INVALID, "bad iteration limits");
  if (!(p->rotation_eps > 0) || !(p->translation_eps > 0)) return fail(c, RGC_ERR_INVALID, "epsilons must be > 0");
  if (p->max_cells < 1) return fail(c, RGC_ERR_INVALID, "max_cells must be >= 1");
  return RGC_OK;
}

const rgc_ctx::BoxHint* find_hint(const rgc_ctx* c, const void* p, int n) {
  for (const auto& h : c->box_hint)
    if (h.p == p && h.n == n && p) return &h;
  return nullptr;
}
void put_hint(rgc_ctx* c, const void* p, int n, const double lo[3], const double hi[3], double reach_xy = 0, double reach_z = 0) {
  rgc_ctx::BoxHint* h = nullptr;
  for (auto& e : c->box_hint)
    if (e.p == p) h = &e;  // a buffer has one box
  if (!h) { h = &c->box_hint[c->box_hint_next]; c->box_hint_next = (c->box_hint_next + 1) % 4; }
  h->p = p; h->n = n;
  h->reach_xy = reach_xy; h->reach_z = reach_z;
  for (int a = 0; a < 3; a++) { h->lo[a] = lo[a]; h->hi[a] = hi[a]; }
}
void drop_hints(rgc_ctx* c) {
  for (auto& e : c->box_hint) e.p = nullptr;
}
// The leaf filter's output lies inside the leaf grid it was sorted on (a leaf's centroid lies in the leaf): a box the library knows without
// measuring.  rgc_set_target_device on that buffer takes its grid from it -- the node's sub-map (three keyframes through the 0.3 m filter)
// changes its bounding box with every keyframe, and a target that left the previous target's widened grid cost a second preparation and a
// second solve (the node's 1.4 ms frames among 0.63 ms ones).
void hint_from_leaf_grid(rgc_ctx* c, const float* out /* device, or the caller's host buffer */, int n_out, const rgck::LeafGrid& g, float leaf) {
  if (!c->spec_on || n_out <= 0) return;
  double lo[3], hi[3];
  for (int a = 0; a < 3; a++) {
    lo[a] = (double)g.minb[a] * (double)leaf - 1.0e-3;
    hi[a] = ((double)g.minb[a] + (double)g.div[a]) * (double)leaf + 1.0e-3;
  }
  put_hint(c, out, n_out, lo, hi);
}

// Has the map preparation enqueued last on the main stream finished?  Asked of the event recorded behind it -- NOT of the stream:
// hipStreamQuery on a stream whose last command is a kernel puts a marker packet with a completion signal behind it (a system-scope
// release in front of whatever is enqueued next: ~10 us between the map's last kernel and the solve's first step, every frame).
bool map_prep_finished(rgc_ctx* c) {
  const hipError_t q = hipEventQuery(c->tgt_prepared);
  if (q != hipSuccess) (void)hipGetLastError();  // ("not ready" is an answer, not an error to be found by a later check)
  return q == hipSuccess;
}

int cloud_covariances(rgc_ctx* c, Cloud& cl, bool is_target);
// a sparse map (points per cell of its grid below map_wide_density) takes the wider block of the bulk kNN launch, see k_knn_sp_wide
int map_wide_r_of(const rgc_ctx* c, const Cloud& cl) {
  return (c->map_wide_r > 0 && (double)cl.n < c->map_wide_density * (double)cl.grid.ncell) ? c->map_wide_r : 0;
}

// C1-C3: grid + exact-kNN covariances (+ voxel map for the target), all enqueued on the stream.
// The first cloud of a context costs one host<->device round trip -- the 6-int bounding box the dense grid is sized from; later
// clouds re-use the previous (widened) grid speculatively and need none (see `spec` below).
rgck::KnnSeeds cloud_seeds(const Cloud& cl, bool is_target);

int prepare_cloud(rgc_ctx* c, Cloud& cl, bool is_target, bool force_bbox = false) {
  const int n = cl.n;
  cl.covs_user = false;
  if (is_target && &cl == &c->tgt) {
    c->tgt_generation++;       // borrowers of the previous target must share again
    c->tgt_owner = nullptr;    // (a borrowed target's aliases are dropped buffer by buffer in ensure())
  }
  hipStream_t s = is_target ? c->stream : c->stream2;
  int* dsm = c->d_small + (is_target ? 0 : 16);
  int* hsm = c->h_small + (is_target ? 0 : 16);
  const int hi = is_target ? 0 : 1;  // the scan's kernels share CUs with the map's kNN launch: raised wave priority
  // The scan is prepared on stream2, concurrently with the map's preparation on the main stream.  Whatever produced the scan was
  // enqueued on the main stream (rgc_upload, the front-end, a caller's own kernels on rgc_stream()): stream2 waits for a mark
  // recorded on the main stream BEFORE this frame's map preparation was enqueued (waiting for the map's kNN launch would serialise
  // the two) -- i.e. at rgc_set_target*, or here when no map preparation is pending.  See rgc_set_source_device in rgc_hip.h.
  if (!is_target && c->main_has_target_prep && map_prep_finished(c)) c->main_has_target_prep = false;  // it has drained
  // (RGC_MARK_BEHIND_COUNT=1, round 4: the map's own mark BEHIND its counting pass -- the record is a call of its own in front of the dependent
  // sequence's first launch, and a scan that waits for one 17 us kernel more has lost nothing; round 5 measured the record's cost on the
  // GPU between the two kernels and put it back in front)
  bool mark_behind_count = false;
  if (is_target || !c->main_has_target_prep || c->main_late_producer) {
    if (is_target && RGC_MARK_BEHIND_COUNT) mark_behind_count = true;
    else HIPCHK(c, hipEventRecord(c->main_mark, c->stream));
    c->mark_valid = true;
    c->main_late_producer = false;
  }
  if (is_target) c->main_has_target_prep = true;
  else if (c->mark_valid) HIPCHK(c, hipStreamWaitEvent(c->stream2, c->main_mark, 0));

  {
    ProfScope ps(c, RGC_K_GRID, n, s);
    // bbox accumulators + flag; the map's copy also zeroes [7], its voxel counter ([8] ncorr stays untouched; the scan's
    // block lives at +16 and must not touch the map's counter)
    // ... unless the previous solve's first step has already put the block back to this image on the device (reinit_small_blocks)
    if (!c->small_clean[is_target ? 0 : 1] || c->small_copy_always) {
      const int init[8] = {INT_MAX, INT_MAX, INT_MAX, INT_MIN, INT_MIN, INT_MIN, 0, 0};
      const size_t init_bytes = 8 * sizeof(int);  // the scan's eighth int (d_small[23]) is its sum of count^2, a float accumulated by the cell scan
      memcpy(hsm, init, init_bytes);
      HIPCHK(c, hipMemcpyAsync(dsm, hsm, init_bytes, hipMemcpyHostToDevice, s));
    }
    c->small_clean[is_target ? 0 : 1] = false;  // this preparation uses it
    // Speculative grid: consecutive clouds of a sequence cover (almost) the same cells, so the previous grid -- widened by two
    // cells in x and y, one in z -- is re-used WITHOUT the bounding-box kernel and its host round trip (the only synchronisation
    // between setInputTarget and the end of align).  A larger bounding grid changes nothing in the results: cells keep their
    // relative order (voxel ids come from the cell scan), neighbourhoods are the same.  k_count guards it; the guard comes home
    // with the LM state (or is read by the first other consumer) and a miss re-prepares the cloud on its own bounding box.
    // The scan's kNN grid need not be the voxel grid (only the map's doubles as the voxel map), and the exact search returns the
    // same neighbours on any grid: a raw 64-beam sweep puts thousands of points into the 1 m cells near the sensor (every query
    // scans its whole cell: O(c^2)), so its cell size follows the crowding measured on the previous frame of the sequence.
    const double res = is_target ? c->prm.voxel_res : (c->src_res > 0.0 ? c->src_res : (c->src_res_auto > 0.0 ? c->src_res_auto : c->prm.voxel_res));
    bool spec = c->spec_on && !c->lm_host && cl.spec_ok && cl.spec_grid.res == res && !force_bbox;
    const rgc_ctx::BoxHint* hint = (is_target && c->spec_on && !c->lm_host && !force_bbox) ? find_hint(c, cl.in, n) : nullptr;
    if (hint)  // (a box that is not one -- it was derived from a pose that was not finite -- is no hint: the float -> int conversions below are undefined on it)
      for (int a = 0; a < 3; a++)
        if (!(std::isfinite(hint->lo[a]) && std::isfinite(hint->hi[a]) && hint->hi[a] >= hint->lo[a] && std::fabs(hint->lo[a]) <= 1.0e8 && std::fabs(hint->hi[a]) <= 1.0e8)) { hint = nullptr; break; }
    rgck::Grid g{};
    if (hint) {  // the box is known (rgc_set_target_reframed / rgc_transform_cloud): its cells plus one on every side, guarded like a speculative grid
      int lo[3], dm[3];
      double ncell = 1.0;
      for (int a = 0; a < 3; a++) {
        lo[a] = (int)std::floor(hint->lo[a] / res - 0.5) - 1;
        dm[a] = (int)std::floor(hint->hi[a] / res - 0.5) + 1 - lo[a] + 1;
        ncell *= (double)dm[a];
      }
      if (ncell <= (double)c->prm.max_cells && ncell <= 2.0e9) g = rgck::make_grid(lo, dm, res);
      else hint = nullptr;
    }
    // A re-framed map (rgc_set_target_reframed) has not been written yet.  With its box known the counting pass below produces it on
    // the way (one pass over the map and one launch less); any other route measures the cloud first and needs it in memory.
    const bool fuse_reframe = cl.reframe_pending && hint != nullptr && cl.stride_f == 4 && ((uintptr_t)cl.in & 15) == 0;  // (k_count<true> stores float4)
    if (cl.reframe_pending && !fuse_reframe)
      rgck::transform_q(s, cl.rf.src, cl.rf.src_stride_f, n, cl.rf.q, cl.rf.t, const_cast<float*>(cl.in), 4);
    // Seeds of the exact search: a re-framed map is the point set of cl.rf.src moved rigidly, so what the last search of that buffer found
    // bounds this one (rgck::KnnSeeds; exactness does not depend on it).  Any other target: no seeds.
    cl.seed_on = false;
    if (is_target && &cl == &c->tgt && cl.reframe_pending && c->seeds_on && !general_route(c) && rgck::knn_seeds_apply(n, c->prm.k_correspondences)) {
      int rc;
      if (cl.seed_key != (const void*)cl.rf.src || cl.seed_n != n || !cl.seed.p) {
        if ((rc = ensure(c, cl.seed, sizeof(float) * (size_t)n))) return rc;
        HIPCHK(c, hipMemsetAsync(cl.seed.p, 0x7f, sizeof(float) * (size_t)n, s));  // 3.4e38: "no seed"
        cl.seed_key = cl.rf.src;
        cl.seed_n = n;
        cl.seed_warm = false;
      }
      cl.seed_on = true;
      // the coordinates' fp32 rounding, twice (two frames), on either end of a distance: 4 ulp of the largest coordinate of the box
      double maxabs = 1.0;
      if (hint) for (int a = 0; a < 3; a++) maxabs = std::max(maxabs, std::max(std::fabs(hint->lo[a]), std::fabs(hint->hi[a])));
      else maxabs = 1024.0;
      int e2;
      (void)std::frexp(1.5 * maxabs, &e2);
      cl.seed_slack = (float)(4.0 * std::ldexp(1.0, e2 - 24));
      // The neighbour lists on top (rgck::KnnCache): the counting pass that produces the map compares it with the library's copy on the
      // way, so only that route has them; a lazy target searches a part of the map per frame and keeps none.
      const double qn = cl.rf.q.x * cl.rf.q.x + cl.rf.q.y * cl.rf.q.y + cl.rf.q.z * cl.rf.q.z + cl.rf.q.w * cl.rf.q.w;
      cl.cache_on = false;
      // (the certificate's error budget is that of a RIGID motion: reframe_point applies v + 2w(u x v) + 2u x (u x v) as Eigen does, without
      // normalising q, so |q|^2 - 1 shows up as a relative error of that order on every distance.  1e-9 is far inside the 4e-6 the
      // certificate allows for and is met by any quaternion normalised in fp64; one normalised in fp32 gets seeds, not lists.)
      bool lists = c->cache_on && fuse_reframe && c->lazy_margin <= 0 && std::fabs(qn - 1.0) < 1.0e-9;
      const size_t cap = (size_t)std::max(256, n / (4 * rgck::kTodoLists) + 1);
      if (lists) {
        // The lists are an optimisation: if the device cannot hold them (112 B per point) the context goes down to the seeds and carries on.
        const size_t want[7] = {sizeof(int) * (size_t)n * 20, sizeof(int) * (size_t)n, sizeof(int) * (size_t)n, sizeof(int) * (size_t)n,
                                sizeof(float4) * (size_t)n, sizeof(int) * cap * rgck::kTodoLists, sizeof(int) * (rgck::kTodoLists + 16)};
        DevBuf* bufs[7] = {&cl.nbr, &cl.pos_of, &cl.rank_of, &cl.qrank, &cl.map_copy, &cl.todo, &cl.cache_small};
        for (int b = 0; b < 7 && lists; b++)
          if (ensure(c, *bufs[b], want[b]) != RGC_OK) lists = false;
        if (c->test_fail_cache_alloc) lists = false;  // (RGC_TEST_FAIL_CACHE_ALLOC at rgc_create: a test's way to walk the path below, as if the device were full)
        if (!lists) {
          (void)hipGetLastError();
          for (DevBuf* b : bufs) release(*b);
          c->cache_on = false;
          c->cache_dropped = true;
          cl.cache_live = false;
          static const bool trace = getenv("RGC_TRACE_ALLOC") != nullptr;
          if (trace) fprintf(stderr, "[rgc] neighbour lists of %d points do not fit on the device: this context keeps seeds only from here on\n", n);
        }
      }
      if (lists) {
        bool fresh = !cl.seed_warm || !cl.cache_live || !cl.nbr.p;
        cl.todo_cap = (int)cap;
        int ce2;  // (a cell of margin around the box, as the grid has)
        (void)std::frexp(1.5 * maxabs + 2.0 * res, &ce2);
        // larger coordinates than the certificates allow for: issue them again.  Smaller ones for sixteen frames in a row: new certificates
        // need the smaller gap only (the old ones, issued for a wider one, stand).
        if (fresh || ce2 > cl.cache_e2) { cl.cache_e2 = ce2; fresh = true; }
        if (ce2 < cl.cache_e2) {
          cl.cache_e2_low_max = cl.cache_e2_low ? std::max(cl.cache_e2_low_max, ce2) : ce2;
          if (++cl.cache_e2_low >= 16) { cl.cache_e2 = cl.cache_e2_low_max; cl.cache_e2_low = 0; }  // (the largest of those sixteen frames)
        } else {
          cl.cache_e2_low = 0;
        }
        if (cl.cache_frame >= (1 << 30)) { cl.cache_frame = 0; fresh = true; }
        cl.cache_frame++;
        if (fresh) HIPCHK(c, hipMemsetAsync(cl.cache_small.p, 0, sizeof(int) * (rgck::kTodoLists + 16), s));  // (list lengths, epoch, overflow)
        cl.rf.copy = (float4*)cl.map_copy.p;
        cl.rf.epoch = (int*)cl.cache_small.p + rgck::kTodoLists;
        cl.rf.frame = cl.cache_frame;
        cl.rf.force = fresh ? 1 : 0;
        cl.cache_on = true;
      }
    } else if (is_target && &cl == &c->tgt) {
      cl.seed_key = nullptr;
      cl.cache_on = false;
    }
    if (is_target && &cl == &c->tgt) {
      cl.cache_live = false;  // (set again by the search that attaches the lists, cloud_covariances)
      cl.searched_known = -1;
      if (!cl.cache_on) { cl.rf.copy = nullptr; cl.rf.epoch = nullptr; }
    }
    cl.reframe_pending = false;
    if (hint) {
      spec = true;
      cl.spec_used = true;
      cl.spec_ok = true;
      cl.spec_grid = g;
    } else if (spec) {
      g = cl.spec_grid;
      cl.spec_used = true;
    } else {
      cl.spec_used = false;
      rgck::bbox(s, cl.in, cl.stride_f, n, res, dsm, dsm + 6, hi);
      HIPCHK(c, hipMemcpyAsync(hsm, dsm, 7 * sizeof(int), hipMemcpyDeviceToHost, s));
      HIPCHK(c, hipStreamSynchronize(s));
      if (hsm[6]) return fail(c, RGC_ERR_NONFINITE, "%s cloud contains non-finite or absurd coordinates", is_target ? "target" : "source");
      double ncell = 1.0;
      for (int a = 0; a < 3; a++) {
        g.minc[a] = hsm[a];
        g.dim[a] = hsm[3 + a] - hsm[a] + 1;
        ncell *= (double)g.dim[a];
      }
      if (ncell > (double)c->prm.max_cells || ncell > 2.0e9)
        return fail(c, RGC_ERR_GRID_TOO_LARGE, "%s grid %d x %d x %d exceeds max_cells", is_target ? "target" : "source", g.dim[0], g.dim[1], g.dim[2]);
      g.res = res;
      g.inv_res = rgck::grid_inv_res(res);
      g.ncell = (int)ncell;
      // The grid the NEXT cloud will try.  The map's box is stable and its grid large: 2 / 2 / 1 cells of margin.  A raw scan's box
      // jumps with every far return, but it stays inside the sensor's range envelope and its grid is small: 16 / 16 / 4 cells of
      // margin, united with the box tried before (a miss costs the scan's whole preparation and a second solve, so the box only
      // ever grows -- at most to four times the measured one).
      rgck::Grid w = g;
      double wcell = 1.0;
      for (int a = 0; a < 3; a++) {
        const int m = is_target ? (a < 2 ? 2 : 1) : (a < 2 ? 16 : 4);
        int lo = g.minc[a] - m, hi = g.minc[a] + g.dim[a] - 1 + m;
        if (!is_target && cl.spec_ok && cl.spec_grid.res == g.res) {
          lo = std::min(lo, cl.spec_grid.minc[a]);
          hi = std::max(hi, cl.spec_grid.minc[a] + cl.spec_grid.dim[a] - 1);
        }
        w.minc[a] = lo; w.dim[a] = hi - lo + 1;
        wcell *= (double)w.dim[a];
      }
      if (!is_target && wcell > 4.0 * ncell + 1.0e6) {  // the union ran away (a sequence that really moves its box): start over from this cloud
        wcell = 1.0;
        for (int a = 0; a < 3; a++) { const int m = a < 2 ? 16 : 4; w.minc[a] = g.minc[a] - m; w.dim[a] = g.dim[a] + 2 * m; wcell *= (double)w.dim[a]; }
      }
      w.ncell = (int)wcell;
      cl.spec_ok = wcell <= (double)c->prm.max_cells && wcell <= 2.0e9;
      cl.spec_grid = w;
      if (cl.spec_ok) {  // its cell arrays now, in the frame that is slow anyway, not in the next one
        const size_t wc1 = (size_t)w.ncell + 1;
        int rc;
        if ((rc = ensure(c, cl.cnt, sizeof(int) * wc1 + 256))) return rc;
        if ((rc = ensure(c, cl.start, sizeof(int) * wc1))) return rc;
        if ((rc = ensure(c, cl.block_sums, sizeof(long long) * (wc1 / 2048 + 2)))) return rc;
        if (is_target && (rc = ensure(c, cl.cell_voxel, sizeof(int) * (size_t)w.ncell))) return rc;
        if (is_target) {  // ... and the voxel table (80 B per cell of a map denser than its grid: 1.3 GB at 16 M cells -- growing it in the next
                          // frame, when the widened grid is first used, was a 77 ms allocation inside c5's three timed frames on a fresh box)
          const size_t vw = (size_t)(n < w.ncell ? n : w.ncell);
          if ((rc = ensure(c, cl.vox, sizeof(double) * rgck::kVoxRec * vw))) return rc;
          if ((rc = ensure(c, cl.vox_cell, sizeof(int) * vw))) return rc;
        }
      }
    }
    cl.grid = g;
    const size_t nc1 = (size_t)g.ncell + 1;  // counters (+ sentinel)
    const size_t ntot = nc1;
    int rc;
    if ((rc = ensure(c, cl.cell_of, sizeof(int) * n))) return rc;
    if ((rc = ensure(c, cl.slot_of, sizeof(int) * n))) return rc;
    // (a hinted box is that of a map re-framed by the vehicle's pose: as the yaw changes it swings between the map's own box and one
    // with twice the cells -- the cell arrays are sized for the largest it can get, once, not grown a few per cent per frame)
    size_t want_cells = ntot;
    if (hint && hint->reach_xy > 0) {
      const double e = hint->reach_xy / res + 6.0, ez = hint->reach_z / res + 6.0;
      const double cells = e * e * ez * 1.05;
      if (cells < 2.0e9 && cells <= (double)c->prm.max_cells && (size_t)cells > want_cells) want_cells = (size_t)cells;
    }
    if ((rc = ensure(c, cl.cnt, sizeof(int) * std::max(ntot, want_cells) + 256))) return rc;
    if ((rc = ensure(c, cl.start, sizeof(int) * std::max(nc1, want_cells)))) return rc;
    if ((rc = ensure(c, cl.block_sums, sizeof(long long) * (std::max(ntot, want_cells) / 2048 + 2)))) return rc;
    if ((rc = ensure(c, cl.order_tmp, sizeof(long long) * n))) return rc;
    if ((rc = ensure(c, cl.P, sizeof(float4) * ((size_t)n + 4)))) return rc;
    if ((rc = ensure(c, cl.segs, rgck::deferred_bytes(n)))) return rc;
    if ((rc = ensure(c, cl.nx, sizeof(double) * n))) return rc;
    if ((rc = ensure(c, cl.ny, sizeof(double) * n))) return rc;
    if ((rc = ensure(c, cl.nz, sizeof(double) * n))) return rc;
    if (is_target && (rc = ensure(c, cl.cell_voxel, sizeof(int) * std::max((size_t)g.ncell, want_cells)))) return rc;
    if (cl.cnt.p != cl.cnt_seen) { cl.cnt_clean = 0; cl.cnt_seen = cl.cnt.p; }  // re-allocated: contents unknown
    if (cl.cnt_clean < ntot) {  // first use or a larger grid; afterwards the scan leaves the counters clean: no fill kernel per frame
      const size_t fill = std::min(cl.cnt.cap, (sizeof(int) * ntot + 255) & ~(size_t)255);
      HIPCHK(c, hipMemsetAsync(cl.cnt.p, 0, fill, s));
      cl.cnt_clean = fill / sizeof(int);
    }  // (a smaller grid leaves the counters beyond it as clean as they were: a re-framed map's box breathes with the yaw)
    rgck::count_cells(s, cl.in, cl.stride_f, n, g, (int*)cl.cell_of.p, (int*)cl.slot_of.p, (int*)cl.cnt.p, hi, spec ? dsm + 6 : nullptr,
                      fuse_reframe ? &cl.rf : nullptr);
    if (mark_behind_count) HIPCHK(c, hipEventRecord(c->main_mark, c->stream));
    rgck::scan_cells(s, (int*)cl.cnt.p, (int*)cl.start.p, (int)ntot, cl.block_sums.p, is_target ? (int*)cl.cell_voxel.p : nullptr,
                     is_target ? c->d_small + 7 : nullptr, hi, is_target ? nullptr : (float*)(c->d_small + 23));
    const bool with_cache = is_target && &cl == &c->tgt && cl.cache_on;
    const rgck::KnnSeeds sd = cloud_seeds(cl, is_target);
    rgck::place(s, n, (const int*)cl.cell_of.p, (const int*)cl.slot_of.p, (const int*)cl.start.p, (unsigned long long*)cl.order_tmp.p, hi,
                with_cache ? &sd.cache : nullptr);
    rgck::rank_gather(s, cl.in, cl.stride_f, n, (const int*)cl.cell_of.p, (const int*)cl.start.p, (const unsigned long long*)cl.order_tmp.p,
                      (float4*)cl.P.p, (int*)cl.segs.p, hi, with_cache ? &sd.cache : nullptr);
  }
  cl.lazy = 0;
  if (is_target && &cl == &c->tgt && c->lazy_margin > 0 && !c->lm_host && !general_route(c) && map_wide_r_of(c, cl) == 0) {
    // lazy target: which part of the map needs covariances and voxels is decided by the solve's guess (rgc_align_begin: lazy_build);
    // any other consumer completes the map first (validate_clouds)
    int rc;
    const size_t vmax = (size_t)(n < cl.grid.ncell ? n : cl.grid.ncell);
    if ((rc = ensure(c, cl.vox, sizeof(double) * rgck::kVoxRec * vmax))) return rc;
    if ((rc = ensure(c, cl.vox_cell, sizeof(int) * vmax))) return rc;
    cl.lazy = 1;
    cl.nvox = -1;
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipEventRecord(c->tgt_prepared, s));
    cl.ready = true;
    return RGC_OK;
  }
  cl.prepared_recorded = false;
  {
    int rc = cloud_covariances(c, cl, is_target);
    if (rc) return rc;
  }
  HIPCHK(c, hipGetLastError());
  if (!is_target) {
    HIPCHK(c, hipEventRecord(c->src_ready, s));
    c->src_pending = true;
  } else if (!cl.prepared_recorded) {  // (otherwise the map's last launch carries the event, cloud_covariances)
    HIPCHK(c, hipEventRecord(c->tgt_prepared, s));
  }
  cl.ready = true;
  return RGC_OK;
}

rgck::KnnSeeds cloud_seeds(const Cloud& cl, bool is_target) {
  rgck::KnnSeeds sd;
  if (is_target && cl.seed_on) {
    sd.seed = (float*)cl.seed.p; sd.slack = cl.seed_slack; sd.warm = cl.seed_warm;
    if (cl.cache_on) {
      rgck::KnnCache& kc = sd.cache;
      kc.nbr = (int*)cl.nbr.p;
      kc.pos_of = (int*)cl.pos_of.p;
      kc.rank_of = (int*)cl.rank_of.p;
      kc.qrank = (int*)cl.qrank.p;
      kc.todo = (int*)cl.todo.p;
      kc.todo_cnt = (int*)cl.cache_small.p;
      kc.epoch = (int*)cl.cache_small.p + rgck::kTodoLists;
      kc.overflow = (int*)cl.cache_small.p + rgck::kTodoLists + 1;
      kc.frame = cl.cache_frame;
      kc.todo_cap = cl.todo_cap;
      // what the coordinates' fp32 rounding in two frames can move a distance by, twice: 4 sqrt(3) ulp of the largest coordinate, and a tenth
      kc.cert_slack = (float)(4.0 * 1.7320508 * 1.1 * std::ldexp(1.0, cl.cache_e2 - 24));
    }
  }
  return sd;
}

// C2 / C3 of a cloud whose grid is built: exact-kNN covariances (+ the Gaussian voxel map for the target), enqueued on the cloud's stream.
int cloud_covariances(rgc_ctx* c, Cloud& cl, bool is_target) {
  const int n = cl.n, k = c->prm.k_correspondences;
  hipStream_t s = is_target ? c->stream : c->stream2;
  int* dsm = c->d_small + (is_target ? 0 : 16);
  cl.general = general_route(c);
  if (cl.general) {  // the general covariance route (general_route above): every point through the cooperative search, a 3x3 per point
    int rc;
    if ((rc = ensure(c, cl.c6, sizeof(double) * 6 * (size_t)n))) return rc;
    const int* guard = cl.spec_used ? dsm + 6 : nullptr;
    {
      ProfScope ps(c, is_target ? RGC_K_KNN_COV : RGC_K_KNN_COV_SRC, n, s);
      rgck::knn_cov6(s, (const float4*)cl.P.p, (const int*)cl.start.p, cl.grid, n, k, c->reg_method, (double*)cl.c6.p, guard);
    }
    if (is_target) {
      const size_t vmax = (size_t)(n < cl.grid.ncell ? n : cl.grid.ncell);
      if ((rc = ensure(c, cl.vox, sizeof(double) * rgck::kVoxRec * vmax))) return rc;
      if ((rc = ensure(c, cl.vox_cell, sizeof(int) * vmax))) return rc;
      ProfScope ps(c, RGC_K_VOXEL, n);
      rgck::voxel_build_general(s, (const float4*)cl.P.p, (const double*)cl.c6.p, (const int*)cl.start.p, cl.grid, n, (const int*)cl.cell_voxel.p,
                                (double*)cl.vox.p, (int*)cl.vox_cell.p, c->voxel_mode == RGC_VOXEL_MULTIPLICATIVE ? 1 : 0, guard);
      cl.nvox = -1;
      cl.cache_searched_lists = false;
    }
    // (the deferred-list counter is zeroed by the grid build and stays zero: no query is deferred on this route)
    return RGC_OK;
  }
  bool stream_coop = false;
  {
    // a sparse map (points per cell of its grid below map_wide_density): the wider block, see k_knn_sp_wide
    const int wide_r = is_target ? map_wide_r_of(c, cl) : 0;
    const int kind = is_target ? RGC_K_KNN_COV : RGC_K_KNN_COV_SRC;
    // The dense map's launch (the dominant kernel) is timed by ITS OWN start / stop times (hipExtLaunchKernelGGL fills the two events):
    // two hipEventRecord packets around it cost ~3 % of a frame of a dependent sequence on two contexts (bench.py's timed region).
    ProfRegion own{};
    bool self_timed = false;
    if (c->prof_on && ((c->prof_mask >> kind) & 1u) && rgck::knn_bulk_times_itself(is_target, wide_r)) {
      auto get = [&](hipEvent_t* e) {
        if (!c->ev_pool.empty()) { *e = c->ev_pool.back(); c->ev_pool.pop_back(); return true; }
        return hipEventCreate(e) == hipSuccess;
      };
      self_timed = get(&own.a);
      if (self_timed && !get(&own.b)) { c->ev_pool.push_back(own.a); self_timed = false; }
      own.kind = kind;
      own.points = n;
    }
    const rgck::KnnSeeds seeds = cloud_seeds(cl, is_target);
    if (is_target) cl.slots_clean = false;
    if (self_timed) {
      rgck::knn_bulk(s, is_target, (const float4*)cl.P.p, (const int*)cl.start.p, cl.grid, n, k, cl.segs.p, (double*)cl.nx.p, (double*)cl.ny.p,
                     (double*)cl.nz.p, cl.spec_used ? dsm + 6 : nullptr, wide_r, own.a, own.b, nullptr, nullptr, 0, seeds);
      c->prof_open.push_back(own);
    } else {
      // the scan: its deferred queries are resolved by the last workgroups of the same launch (coop_stream) -- unless the stage-by-stage
      // profile wants the two apart.  The entry words must be "empty" on entry: filled once per allocation, the readers put it back.
      const int coop_waves_s = cl.deferred_seen >= 0 ? cl.deferred_seen + cl.deferred_seen / 4 + 32 : n / 64 + 32;
      stream_coop = !is_target && c->coop_stream_on && k <= 32 && !(c->prof_on && ((c->prof_mask >> RGC_K_KNN_COOP_SRC) & 1u));
      if (stream_coop && (!cl.slots_clean || cl.slots_seen != cl.segs.p)) {
        HIPCHK(c, hipMemsetAsync((int*)cl.segs.p + 16, rgck::kDeferredSlotEmptyByte, sizeof(int) * 2 * (size_t)n, s));
        cl.slots_clean = true;
        cl.slots_seen = cl.segs.p;
      }
      if (!stream_coop) cl.slots_clean = false;  // (the plain list will be written over the slots; a cloud can change roles: rgc_swap_source_and_target)
      ProfScope ps(c, kind, n, s);
      rgck::knn_bulk(s, is_target, (const float4*)cl.P.p, (const int*)cl.start.p, cl.grid, n, k, cl.segs.p, (double*)cl.nx.p, (double*)cl.ny.p,
                     (double*)cl.nz.p, cl.spec_used ? dsm + 6 : nullptr, wide_r, nullptr, nullptr, nullptr, nullptr, 0, seeds, stream_coop ? coop_waves_s : 0);
    }
    if (is_target) cl.cache_searched_lists = seeds.cache.nbr && seeds.warm && wide_r == 0;
    if (seeds.seed && wide_r == 0) {
      cl.seed_warm = true;
      if (seeds.cache.nbr) cl.cache_live = true;
    }
  }
  // The map's deferred queries (~100 of a million, one wave each: 20 us of latency) are resolved in the SAME launch as the voxel map's
  // build (k_voxel_build_coop); the few voxels that hold one are recomputed behind it (k_voxel_patch).  The scan has no voxel map:
  // its chain stays serial.
  const bool coop_beside = is_target;
  // grid of the cooperative launch: twice the deferred count of the previous cloud prepared here (consecutive clouds of a sequence
  // defer about the same queries), n / 64 for the first one
  const int coop_waves = cl.deferred_seen >= 0 ? cl.deferred_seen + cl.deferred_seen / 4 + 32 : n / 64 + 32;
  if (!coop_beside && !stream_coop) {
    ProfScope ps(c, is_target ? RGC_K_KNN_COOP : RGC_K_KNN_COOP_SRC, n, s);
    rgck::knn_coop(s, is_target, (const float4*)cl.P.p, (const int*)cl.start.p, cl.grid, n, k, cl.segs.p, (double*)cl.nx.p,
                   (double*)cl.ny.p, (double*)cl.nz.p, cl.spec_used ? dsm + 6 : nullptr, coop_waves);
  }
  if (is_target) {
    int rc;
    const size_t vmax = (size_t)(n < cl.grid.ncell ? n : cl.grid.ncell);
    if ((rc = ensure(c, cl.vox, sizeof(double) * rgck::kVoxRec * vmax))) return rc;
    if ((rc = ensure(c, cl.vox_cell, sizeof(int) * vmax))) return rc;
    ProfScope ps(c, RGC_K_VOXEL, n);
    {
      rgck::voxel_build_coop(s, (const float4*)cl.P.p, (double*)cl.nx.p, (double*)cl.ny.p, (double*)cl.nz.p, (const int*)cl.start.p, cl.grid, n,
                             (const int*)cl.cell_voxel.p, (double*)cl.vox.p, (int*)cl.vox_cell.p, k, cl.segs.p, cl.spec_used ? dsm + 6 : nullptr, coop_waves,
                             cloud_seeds(cl, true));
      rgck::voxel_patch(s, (const float4*)cl.P.p, (const double*)cl.nx.p, (const double*)cl.ny.p, (const double*)cl.nz.p, (const int*)cl.start.p,
                        cl.grid, cl.segs.p, (const int*)cl.cell_voxel.p, (double*)cl.vox.p, cl.deferred_seen >= 0 ? 2 * cl.deferred_seen + 64 : n,
                        (c->prep_event_ext && &cl == &c->tgt) ? c->tgt_prepared : nullptr);
      cl.prepared_recorded = c->prep_event_ext && &cl == &c->tgt;
    }
    cl.nvox = -1;  // fetched lazily
  }
  return RGC_OK;
}

// Lazy target, at rgc_align_begin: the cells within lazy_margin cells of where the scan falls at the guess are stamped (the look-up's own
// arithmetic), and covariances + voxels are built for those only -- at c-main 8 % of the map's points at a margin of two cells.  The
// solve checks every look-up (linearize_point): one that lands on an occupied voxel outside the stamped set makes rgc_align_end
// complete the map and solve again, so the result is the full build's bit for bit either way.
rgck::Pose pose_from(const double T[16]);
int lazy_build(rgc_ctx* c, const float guess[16]) {
  Cloud& cl = c->tgt;
  hipStream_t s = c->stream;
  int rc;
  const int n = cl.n, k = c->prm.k_correspondences;
  const size_t cells = (size_t)cl.grid.ncell;
  if ((rc = ensure(c, cl.need, sizeof(int) * cells + 256))) return rc;
  if ((rc = ensure(c, cl.qlist, sizeof(int) * (size_t)n + 256))) return rc;
  if ((rc = ensure(c, cl.cell_list, sizeof(int) * (size_t)n + 256))) return rc;
  cl.need_stamp++;
  if (cl.need.p != cl.need_seen || cl.need_stamp > 0x3fffffff) {  // a new allocation (or the stamps ran out): start over from a clean array
    HIPCHK(c, hipMemsetAsync(cl.need.p, 0, cl.need.cap, s));
    cl.need_seen = cl.need.p;
    cl.need_stamp = 1;
  }
  if (c->src_in_pending) {  // the scan came from the host: its upload ran on stream2
    HIPCHK(c, hipStreamWaitEvent(s, c->src_in_ready, 0));
    c->src_in_pending = false;
  }
  double T[16];
  for (int i = 0; i < 16; i++) T[i] = (double)guess[i];
  int* counts = (int*)cl.segs.p + 1;  // [0] listed queries, [1] listed cells: behind the deferred-query counter, zeroed with it by k_rank_gather
  const int* guard = cl.spec_used ? c->d_small + 6 : nullptr;
  rgck::footprint(s, c->src.in, c->src.stride_f, c->src.n, pose_from(T), cl.grid, (int*)cl.need.p, cl.need_stamp, c->lazy_margin, (const float4*)cl.P.p, n,
                  (const int*)cl.start.p, (int*)cl.qlist.p, (int*)cl.cell_list.p, counts, guard);
  const int q_est = cl.lazy_nq_seen >= 0 ? cl.lazy_nq_seen + cl.lazy_nq_seen / 4 + 4096 : n;
  const int c_est = cl.lazy_ncell_seen >= 0 ? cl.lazy_ncell_seen + cl.lazy_ncell_seen / 4 + 1024 : n / 8 + 1024;
  {
    ProfScope ps(c, RGC_K_KNN_COV, n, s);
    rgck::knn_bulk(s, true, (const float4*)cl.P.p, (const int*)cl.start.p, cl.grid, n, k, cl.segs.p, (double*)cl.nx.p, (double*)cl.ny.p, (double*)cl.nz.p,
                   guard, 0, nullptr, nullptr, (const int*)cl.qlist.p, counts, q_est, cloud_seeds(cl, true));
    if (cl.seed_on) cl.seed_warm = true;
  }
  {
    const int coop_waves = cl.deferred_seen >= 0 ? cl.deferred_seen + cl.deferred_seen / 4 + 32 : n / 64 + 32;
    ProfScope ps(c, RGC_K_VOXEL, n);
    rgck::voxel_cells_coop(s, (const float4*)cl.P.p, (double*)cl.nx.p, (double*)cl.ny.p, (double*)cl.nz.p, (const int*)cl.start.p, cl.grid, n,
                           (const int*)cl.cell_voxel.p, (double*)cl.vox.p, (int*)cl.vox_cell.p, k, cl.segs.p, guard, coop_waves, (const int*)cl.cell_list.p,
                           counts + 1, c_est, cloud_seeds(cl, true));
    rgck::voxel_patch(s, (const float4*)cl.P.p, (const double*)cl.nx.p, (const double*)cl.ny.p, (const double*)cl.nz.p, (const int*)cl.start.p,
                      cl.grid, cl.segs.p, (const int*)cl.cell_voxel.p, (double*)cl.vox.p, cl.deferred_seen >= 0 ? 2 * cl.deferred_seen + 64 : n);
  }
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipEventRecord(c->tgt_prepared, s));  // (the map's preparation ends HERE now: what the solve goes behind, what another context's held scan waits for)
  cl.lazy = 2;
  return RGC_OK;
}

// every consumer of the target other than the chained solve: the whole map, as without the lazy mode
int complete_target(rgc_ctx* c) {
  Cloud& cl = c->tgt;
  if (!cl.ready || cl.lazy == 0 || c->tgt_owner) return RGC_OK;
  HIPCHK(c, hipMemsetAsync(cl.segs.p, 0, sizeof(int), c->stream));  // the deferred-query counter of the bulk launch
  int rc = cloud_covariances(c, cl, true);
  if (rc) return rc;
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipEventRecord(c->tgt_prepared, c->stream));
  cl.lazy = 0;
  // (corr_valid stays as it is: the completed map has the voxel ids and records of the part a solve has used, bit for bit, so the
  // correspondences that solve froze are still the ones rgc_compute_error needs -- as after a solve on a fully built target)
  c->deferred_known = false;
  c->main_has_target_prep = true;
  return RGC_OK;
}

// between rgc_align_begin and rgc_align_end -- on the general route too, where the solve has already run and its result waits to be handed over:
// the same calls are refused on both routes
static inline bool solve_in_flight(const rgc_ctx* c) { return c->pend.active || c->gen_res.on; }

// rf (nullable, device clouds only): xyz has not been written yet -- the preparation produces it from rf (rgc_set_target_reframed)
int set_cloud(rgc_ctx* c, Cloud& cl, bool is_target, const float* xyz, int n, int stride_bytes, bool on_device, const rgck::Reframe* rf = nullptr) {
  if (!c) return RGC_ERR_INVALID;
  cl.reframe_pending = false;
  if (solve_in_flight(c)) return fail(c, RGC_ERR_INVALID, "a solve is in flight on this context: call rgc_align_end first");
  cl.ready = false;
  cl.n = 0;
  c->corr_valid = false;
  c->deferred_known = false;
  if (is_target) c->map_bound = false;
  { const int rs = check_supported(c); if (rs) return rs; }
  if (!xyz || n < 0) return fail(c, RGC_ERR_INVALID, "null cloud");
  if (stride_bytes < 12 || (stride_bytes & 3) || stride_bytes > 4096) return fail(c, RGC_ERR_INVALID, "stride_bytes must be a multiple of 4 and >= 12");
  if (n > (1 << 27)) return fail(c, RGC_ERR_INVALID, "cloud has %d points, the limit is 2^27 (32-bit byte offsets into the sorted array)", n);
  if (n < c->prm.k_correspondences)
    return fail(c, RGC_ERR_TOO_FEW_POINTS, "%s cloud has %d points, need >= k = %d", is_target ? "target" : "source", n, c->prm.k_correspondences);
  HIPCHK(c, hipSetDevice(c->device));
  const int stride_f = stride_bytes / 4;
  if (on_device) { const int rk = check_device_range(c, xyz, (size_t)n * stride_bytes - (stride_bytes - 12), is_target ? "target cloud" : "source cloud"); if (rk) return rk; }
  if (on_device) {
    cl.in = xyz;
    if (!is_target) c->src_in_pending = false;  // (no upload of this scan to wait for)
  } else {
    const size_t bytes = (size_t)n * stride_bytes;
    int rc = ensure(c, cl.in_copy, bytes);
    if (rc) return rc;
    if (!is_target && c->src_read_pending) {  // a kernel on the main stream still reads the previous scan out of this buffer (rgc_get_aligned_device)
      HIPCHK(c, hipStreamWaitEvent(c->stream2, c->src_read_done, 0));
      c->src_read_pending = false;
    }
    // pageable host memory: hipMemcpyAsync stages and returns once the source has been consumed
    HIPCHK(c, hipMemcpyAsync(cl.in_copy.p, xyz, bytes - (stride_bytes - 12), hipMemcpyHostToDevice, is_target ? c->stream : c->stream2));
    if (!is_target) {  // (a lazy target's footprint pass reads the scan's input on the main stream -- whenever rgc_set_target_lazy was called)
      HIPCHK(c, hipEventRecord(c->src_in_ready, c->stream2));
      c->src_in_pending = true;
    }
    cl.in = (const float*)cl.in_copy.p;
    // (a box known for the caller's HOST buffer -- the leaf filter's output, rgc_voxelgrid -- goes with the cloud to its device copy)
    if (is_target) {
      if (const rgc_ctx::BoxHint* h = find_hint(c, xyz, n)) { const rgc_ctx::BoxHint hh = *h; put_hint(c, cl.in, n, hh.lo, hh.hi); }
      else for (auto& e : c->box_hint) if (e.p == (const void*)cl.in) e.p = nullptr;  // (the staging buffer's last cloud's box says nothing about this one)
    }
  }
  cl.stride_f = stride_f;
  cl.n = n;
  if (rf && on_device) { cl.rf = *rf; cl.reframe_pending = true; }
  int rc = prepare_cloud(c, cl, is_target);
  if (rc) { cl.n = 0; return rc; }
  if (is_target) { c->stats.n_target = n; c->stats.target_cells = cl.grid.ncell; }
  else { c->stats.n_source = n; c->stats.source_cells = cl.grid.ncell; }
  return RGC_OK;
}

rgck::Pose pose_from(const double T[16]) {
  rgck::Pose P;
  for (int a = 0; a < 3; a++) {
    for (int b = 0; b < 3; b++) P.R[a * 3 + b] = T[a * 4 + b];
    P.t[a] = T[a * 4 + 3];
  }
  return P;
}
rgck::PoseF posef_from(const float T[16]) {
  rgck::PoseF P;
  for (int a = 0; a < 12; a++) P.m[a] = T[a];
  return P;
}

// order the main stream after the source preprocessing (which runs on stream2)
int join_source(rgc_ctx* c) {
  if (c->src_pending) {
    // (a scan prepared ahead -- two contexts taking turns -- has usually finished by now: then no barrier packet goes into the main
    // stream at all; a dependency that has to be resolved across streams costs ~10 us in front of the kernel behind it, even a met one)
    // A scan that is ALMOST ready -- two contexts taking turns: its cooperative search is still running beside the map's preparation --
    // is waited for here, on the host, for as long as the map's preparation is still running anyway (bounded: join_spin_us): the solve's
    // launches are not needed in the queue before that, and the host has nothing else to do until the solve ends.
    bool ready = hipEventQuery(c->src_ready) == hipSuccess;
    if (!ready && c->join_spin_us > 0) {
      const auto t0 = std::chrono::steady_clock::now();
      for (;;) {
        (void)hipGetLastError();
        if (hipEventQuery(c->src_ready) == hipSuccess) { ready = true; break; }
        if (map_prep_finished(c)) break;
        if (std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() > (double)c->join_spin_us) break;
        __builtin_ia32_pause();  // (a sibling hyper-thread may be driving another sequence's context)
      }
    }
    if (!ready) {
      (void)hipGetLastError();  // ("not ready" is an answer, not an error to be found by a later check)
      HIPCHK(c, hipStreamWaitEvent(c->stream, c->src_ready, 0));
    }
    c->src_pending = false;
  }
  return RGC_OK;
}

// Guards of speculative grids (bit 0: non-finite point, bit 1: point outside the grid).  Returns < 0 on error, 1 if a cloud had
// to be prepared again on its own bounding box (whatever was computed from it must be redone), 0 if everything stands.
int resolve_guards(rgc_ctx* c, int guard_t, int guard_s) {
  int redo = 0;
  Cloud* cl[2] = {&c->tgt, &c->src};
  const int gd[2] = {guard_t, guard_s};
  for (int a = 0; a < 2; a++) {
    if (!cl[a]->spec_used) continue;
    cl[a]->spec_used = false;
    if (!cl[a]->ready || cl[a]->n <= 0) continue;  // (a cloud cleared since -- rgc_clear_source / _target --: its guard word is the last cloud's, there is nothing to prepare again)
    if (!gd[a]) continue;
    cl[a]->ready = false;
    c->corr_valid = false;
    c->deferred_known = false;
    if (gd[a] & 1) { cl[a]->n = 0; return fail(c, RGC_ERR_NONFINITE, "%s cloud contains non-finite or absurd coordinates", a == 0 ? "target" : "source"); }
    static const bool trace = getenv("RGC_TRACE_ALLOC") != nullptr;
    if (trace) fprintf(stderr, "[rgc] %s cloud left its speculative grid: prepared again\n", a == 0 ? "target" : "source");
    if (a == 0) drop_hints(c);  // (a hinted box that did not hold: its buffer was rewritten behind the library's back -- measure again)
    int rc = prepare_cloud(c, *cl[a], a == 0, /*force_bbox=*/true);
    if (rc) { cl[a]->n = 0; return rc; }
    redo = 1;
  }
  return redo;
}

// A borrowed target (rgc_share_target) is only as good as its owner: alive (the same context, not a new one at its address) and not
// prepared again since.  Called by EVERY consumer of c->tgt -- the solve, the fine seam, the fitness score, the getters.
int check_target_owner(rgc_ctx* c) {
  if (!c->tgt_owner) return RGC_OK;
  bool alive;
  unsigned long long owner_gen = 0;
  {
    std::lock_guard<std::mutex> lk(g_live_mutex);  // (the owner is only looked at while it cannot be destroyed)
    alive = g_live.count(c->tgt_owner) != 0 && c->tgt_owner->uid == c->tgt_owner_uid;
    if (alive) owner_gen = c->tgt_owner->tgt_generation;
  }
  if (!alive) {  // its device buffers are gone with it
    c->tgt_owner = nullptr;
    for (DevBuf* b : {&c->tgt.in_copy, &c->tgt.cell_of, &c->tgt.slot_of, &c->tgt.cnt, &c->tgt.start, &c->tgt.block_sums, &c->tgt.order_tmp, &c->tgt.P,
                      &c->tgt.nx, &c->tgt.ny, &c->tgt.nz, &c->tgt.segs, &c->tgt.cell_voxel, &c->tgt.vox, &c->tgt.vox_cell, &c->tgt.c6})
      release(*b);
    c->tgt.ready = false;
    c->tgt.n = 0;
    return fail(c, RGC_ERR_NO_INPUT, "the context whose target this one shared has been destroyed");
  }
  if (owner_gen != c->tgt_owner_gen)
    return fail(c, RGC_ERR_INVALID, "the shared target was rebuilt by its owner: rgc_share_target again");
  return RGC_OK;
}

// for every consumer except rgc_align (which gets the guards with its state read-back): one synchronisation, once per cloud
int validate_clouds(rgc_ctx* c, bool whole_target = true) {
  if (solve_in_flight(c)) return fail(c, RGC_ERR_INVALID, "a solve is in flight on this context: call rgc_align_end first");
  { int rc = check_target_owner(c); if (rc) return rc; }
  // The guards FIRST: a lazy target whose speculative grid did not hold must be prepared again on its own box BEFORE it is completed -- the
  // completion's kernels leave at once on a tripped guard, and the re-preparation behind it would put back an unbuilt lazy target that
  // nobody completes any more (a target replaced while a scan is set, then read through a getter: tests/fuzz/fuzz_api.py found it).
  if ((c->tgt.ready && c->tgt.spec_used) || (c->src.ready && c->src.spec_used)) {
    HIPCHK(c, hipStreamSynchronize(c->stream2));
    HIPCHK(c, hipMemcpyAsync(c->h_small + 6, c->d_small + 6, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->h_small + 22, c->d_small + 22, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    const int r = resolve_guards(c, c->tgt.spec_used ? c->h_small[6] : 0, c->src.spec_used ? c->h_small[22] : 0);
    if (r < 0) return r;
  }
  if (whole_target) { int rc = complete_target(c); if (rc) return rc; }  // (lazy target: whoever comes this way reads covariances or voxels the solve may not have needed)
  return RGC_OK;
}

int need_inputs(rgc_ctx* c, bool validate = true) {
  if (!c) return RGC_ERR_INVALID;
  { const int rs = check_supported(c); if (rs) return rs; }
  if (!c->src.ready || !c->tgt.ready) return fail(c, RGC_ERR_NO_INPUT, "source and target must be set first");
  if (validate) {
    int rc = validate_clouds(c);
    if (rc) return rc;
    if (!c->src.ready || !c->tgt.ready) return fail(c, RGC_ERR_NO_INPUT, "source and target must be set first");
  }
  return join_source(c);
}

int do_linearize(rgc_ctx* c, const double T[16], double* H, double* b, double* cost) {
  int rc = need_inputs(c);
  if (rc) return rc;
  const int n = c->src.n, noff = noff_of(c->prm.neighbor_method);
  if ((rc = ensure(c, c->corr_v, sizeof(int) * (size_t)n * noff))) return rc;
  if ((rc = ensure(c, c->corr_M, sizeof(double) * 6 * (size_t)n * noff))) return rc;
  const int nb = rgck::linearize_blocks(n);
  if ((rc = ensure(c, c->partials, sizeof(double) * rgck::kAccum * (size_t)nb))) return rc;
  if ((rc = ensure(c, c->ipartials, sizeof(int) * (size_t)nb))) return rc;
  const int want = (H && b) ? 1 : 0;
  {
    ProfScope ps(c, RGC_K_LINEARIZE, n);
    if (c->src.general)
      rgck::linearize_general(c->stream, (const float4*)c->src.P.p, (const double*)c->src.c6.p, n, pose_from(T), c->tgt.grid, (const int*)c->tgt.cell_voxel.p,
                              (const double*)c->tgt.vox.p, noff, (int*)c->corr_v.p, (double*)c->corr_M.p, want, (double*)c->partials.p,
                              (int*)c->ipartials.p, c->d_out, c->d_small + 8);
    else
    rgck::linearize(c->stream, (const float4*)c->src.P.p,
                    (const double*)c->src.nx.p, (const double*)c->src.ny.p, (const double*)c->src.nz.p, n, pose_from(T), c->tgt.grid,
                    (const int*)c->tgt.cell_voxel.p, (const double*)c->tgt.vox.p, noff, (int*)c->corr_v.p, (double*)c->corr_M.p, want,
                    (double*)c->partials.p, (int*)c->ipartials.p, c->d_out, c->d_small + 8);
  }
  HIPCHK(c, hipMemcpyAsync(c->h_out, c->d_out, sizeof(double) * rgck::kAccum, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->h_small + 8, c->d_small + 8, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipGetLastError());
  c->corr_noff = noff;
  c->corr_n = n;
  c->corr_valid = true;
  c->stats.n_corr = c->h_small[8];
  c->stats.n_linearize++;
  if (want) {
    int u = 0;
    for (int a = 0; a < 6; a++)
      for (int d = a; d < 6; d++) {
        H[a * 6 + d] = c->h_out[u];
        H[d * 6 + a] = c->h_out[u];
        u++;
      }
    for (int a = 0; a < 6; a++) b[a] = c->h_out[21 + a];
  }
  if (cost) *cost = c->h_out[27];
  return RGC_OK;
}

// One enqueue per outer LM iteration: linearize at x0, fold, FIRST LM try on the device (solve, so3_exp, xi = delta*x0),
// compute_error at xi, fold -- then a single 57-double read-back.  lambda < 0: lambda = factor * max|H_ii| on the device.
int do_linearize_try(rgc_ctx* c, const double x0[16], double lambda, double H[36], double b[6], double* y0, double d[6], double xi[16],
                     double* lambda_used, double* yi) {
  int rc = need_inputs(c);
  if (rc) return rc;
  const int n = c->src.n, noff = noff_of(c->prm.neighbor_method);
  if ((rc = ensure(c, c->corr_v, sizeof(int) * (size_t)n * noff))) return rc;
  if ((rc = ensure(c, c->corr_M, sizeof(double) * 6 * (size_t)n * noff))) return rc;
  const int nb = rgck::linearize_blocks(n);
  if ((rc = ensure(c, c->partials, sizeof(double) * rgck::kAccum * (size_t)nb))) return rc;
  if ((rc = ensure(c, c->ipartials, sizeof(int) * (size_t)nb))) return rc;
  rgck::LmIn in;
  for (int i = 0; i < 16; i++) in.x0[i] = x0[i];
  in.lambda = lambda;
  in.init_factor = c->prm.lm_init_lambda_factor;
  {
    ProfScope ps(c, RGC_K_LINEARIZE, n);
    if (c->src.general)
      rgck::linearize_general(c->stream, (const float4*)c->src.P.p, (const double*)c->src.c6.p, n, pose_from(x0), c->tgt.grid, (const int*)c->tgt.cell_voxel.p,
                              (const double*)c->tgt.vox.p, noff, (int*)c->corr_v.p, (double*)c->corr_M.p, 1, (double*)c->partials.p,
                              (int*)c->ipartials.p, c->d_out, c->d_small + 8);
    else
    rgck::linearize(c->stream, (const float4*)c->src.P.p, (const double*)c->src.nx.p, (const double*)c->src.ny.p, (const double*)c->src.nz.p, n,
                    pose_from(x0), c->tgt.grid, (const int*)c->tgt.cell_voxel.p, (const double*)c->tgt.vox.p, noff, (int*)c->corr_v.p,
                    (double*)c->corr_M.p, 1, (double*)c->partials.p, (int*)c->ipartials.p, c->d_out, c->d_small + 8);
    rgck::lm_try(c->stream, c->d_out, c->d_small + 8, in);
  }
  {
    ProfScope ps(c, RGC_K_ERROR, n);
    rgck::compute_error_dev(c->stream, (const float4*)c->src.P.p, n, c->d_out + 38, (const double*)c->tgt.vox.p, noff, (const int*)c->corr_v.p,
                            (const double*)c->corr_M.p, (double*)c->partials.p, c->d_out + 56);
  }
  HIPCHK(c, hipMemcpyAsync(c->h_out, c->d_out, sizeof(double) * 57, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipGetLastError());
  c->corr_noff = noff;
  c->corr_n = n;
  c->corr_valid = true;
  c->stats.n_corr = (int)c->h_out[28];
  c->stats.n_linearize++;
  c->stats.n_error++;
  int u = 0;
  for (int a = 0; a < 6; a++)
    for (int e = a; e < 6; e++) { H[a * 6 + e] = c->h_out[u]; H[e * 6 + a] = c->h_out[u]; u++; }
  for (int a = 0; a < 6; a++) b[a] = c->h_out[21 + a];
  *y0 = c->h_out[27];
  for (int a = 0; a < 6; a++) d[a] = c->h_out[32 + a];
  for (int a = 0; a < 16; a++) xi[a] = c->h_out[38 + a];
  *lambda_used = c->h_out[54];
  *yi = c->h_out[56];
  return RGC_OK;
}

int do_error(rgc_ctx* c, const double T[16], double* cost) {
  int rc = need_inputs(c);
  if (rc) return rc;
  if (!c->corr_valid) return fail(c, RGC_ERR_INVALID, "rgc_compute_error needs a preceding rgc_linearize");
  const int n = c->corr_n;
  {
    ProfScope ps(c, RGC_K_ERROR, n);
    rgck::compute_error(c->stream, (const float4*)c->src.P.p, n, pose_from(T),
                        (const double*)c->tgt.vox.p, c->corr_noff, (const int*)c->corr_v.p, (const double*)c->corr_M.p,
                        (double*)c->partials.p, c->d_out);
  }
  HIPCHK(c, hipMemcpyAsync(c->h_out, c->d_out, sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipGetLastError());
  c->stats.n_error++;
  *cost = c->h_out[0];
  return RGC_OK;
}

int do_fitness(rgc_ctx* c, const float T[16], double* out) {
  int rc = need_inputs(c);
  if (rc) return rc;
  const int n = c->src.n;
  if ((rc = ensure(c, c->partials, sizeof(double) * (size_t)rgck::fitness_blocks(n) + 64))) return rc;
  {
    ProfScope ps(c, RGC_K_FITNESS, n);
    rgck::fitness(c->stream, (const float4*)c->src.P.p, n, posef_from(T), (const float4*)c->tgt.P.p, (const int*)c->tgt.start.p,
                  c->tgt.grid, (double*)c->partials.p, c->d_out, c->tgt.n);
  }
  HIPCHK(c, hipMemcpyAsync(c->h_out, c->d_out, sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipGetLastError());
  *out = c->h_out[0] / (double)n;
  return RGC_OK;
}

// ---- scalar helpers of the LM driver (so3_exp, LDLT solve, 4x4 product: rgc_lm.h, shared with the device) ----
// lsq_registration_impl.hpp:82-91
bool is_converged(const double d[16], double rot_eps, double trans_eps) {
  double m = 0;
  for (int a = 0; a < 3; a++) {
    for (int b = 0; b < 3; b++) m = std::fmax(m, std::fabs(d[a * 4 + b] - (a == b ? 1.0 : 0.0)) / rot_eps);
    m = std::fmax(m, std::fabs(d[a * 4 + 3]) / trans_eps);
  }
  return m < 1;
}

// ---- f1: grid of a feature map (bbox -> counting sort; no covariances) and the host side of the robust LM ----------------
// grid cell of a feature map = the largest 5th-neighbour distance that still yields a factor (1 m for edges :1098, sqrt(2) m for
// planes :1200): the 3x3x3 block of cells then proves every accepted neighbourhood, and holds as few candidates as possible
constexpr double kMapregCell[2] = {1.0, 1.4143};

int prepare_map_grid(rgc_ctx* c, Cloud& cl, double cell) {
  const int n = cl.n;
  hipStream_t s = c->stream;
  int rc;
  if ((rc = ensure(c, c->mr_small, 64))) return rc;
  int* dsm = (int*)c->mr_small.p;
  int hsm[8] = {INT_MAX, INT_MAX, INT_MAX, INT_MIN, INT_MIN, INT_MIN, 0, 0};
  HIPCHK(c, hipMemcpyAsync(dsm, hsm, 7 * sizeof(int), hipMemcpyHostToDevice, s));
  rgck::bbox(s, cl.in, cl.stride_f, n, cell, dsm, dsm + 6);
  HIPCHK(c, hipMemcpyAsync(hsm, dsm, 7 * sizeof(int), hipMemcpyDeviceToHost, s));
  HIPCHK(c, hipStreamSynchronize(s));
  if (hsm[6]) return fail(c, RGC_ERR_NONFINITE, "feature map contains non-finite or absurd coordinates");
  rgck::Grid g{};
  double ncell = 1.0;
  for (int a = 0; a < 3; a++) {
    g.minc[a] = hsm[a];
    g.dim[a] = hsm[3 + a] - hsm[a] + 1;
    ncell *= (double)g.dim[a];
  }
  if (ncell > (double)c->prm.max_cells || ncell > 2.0e9) return fail(c, RGC_ERR_GRID_TOO_LARGE, "feature-map grid exceeds max_cells");
  g.res = cell;
  g.inv_res = rgck::grid_inv_res(cell);
  g.ncell = (int)ncell;
  cl.grid = g;
  const size_t nc1 = (size_t)g.ncell + 1;
  if ((rc = ensure(c, cl.cell_of, sizeof(int) * n))) return rc;
  if ((rc = ensure(c, cl.slot_of, sizeof(int) * n))) return rc;
  if ((rc = ensure(c, cl.cnt, sizeof(int) * nc1 + 256))) return rc;
  if ((rc = ensure(c, cl.start, sizeof(int) * nc1))) return rc;
  if ((rc = ensure(c, cl.block_sums, sizeof(long long) * (nc1 / 2048 + 2)))) return rc;
  if ((rc = ensure(c, cl.order_tmp, sizeof(long long) * n))) return rc;
  if ((rc = ensure(c, cl.P, sizeof(float4) * ((size_t)n + 4)))) return rc;
  HIPCHK(c, hipMemsetAsync(cl.cnt.p, 0, (sizeof(int) * nc1 + 255) & ~(size_t)255, s));
  rgck::count_cells(s, cl.in, cl.stride_f, n, g, (int*)cl.cell_of.p, (int*)cl.slot_of.p, (int*)cl.cnt.p);
  rgck::scan_cells(s, (int*)cl.cnt.p, (int*)cl.start.p, (int)nc1, cl.block_sums.p, nullptr, nullptr);
  rgck::place(s, n, (const int*)cl.cell_of.p, (const int*)cl.slot_of.p, (const int*)cl.start.p, (unsigned long long*)cl.order_tmp.p);
  rgck::rank_gather(s, cl.in, cl.stride_f, n, (const int*)cl.cell_of.p, (const int*)cl.start.p, (const unsigned long long*)cl.order_tmp.p, (float4*)cl.P.p);
  HIPCHK(c, hipGetLastError());
  cl.ready = true;
  return RGC_OK;
}

// Cholesky solve of a symmetric positive definite n x n, n <= 12 (the damped normal equations of the two poses: block
// diagonal unless the IMU block couples the rotations)
bool chol_solve(const double* A, const double* rhs, double* x, int n) {
  double L[144] = {0}, y[12];
  for (int i = 0; i < n; i++)
    for (int j = 0; j <= i; j++) {
      double s = A[i * n + j];
      for (int k = 0; k < j; k++) s -= L[i * n + k] * L[j * n + k];
      if (i == j) { if (!(s > 0)) return false; L[i * n + i] = std::sqrt(s); }
      else L[i * n + j] = s / L[j * n + j];
    }
  for (int i = 0; i < n; i++) { double s = rhs[i]; for (int k = 0; k < i; k++) s -= L[i * n + k] * y[k]; y[i] = s / L[i * n + i]; }
  for (int i = n - 1; i >= 0; i--) { double s = y[i]; for (int k = i + 1; k < n; k++) s -= L[k * n + i] * x[k]; x[i] = s / L[i * n + i]; }
  return true;
}

void quat_plus(const double q[4], const double d[3], double out[4]) {  // EigenQuaternionParameterization::Plus [3P-memory]
  const double nd = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
  double dq[4];
  if (nd > 0.0) { const double s = std::sin(nd) / nd; dq[0] = s * d[0]; dq[1] = s * d[1]; dq[2] = s * d[2]; dq[3] = std::cos(nd); }
  else { dq[0] = d[0]; dq[1] = d[1]; dq[2] = d[2]; dq[3] = 1.0; }
  const double ax = dq[0], ay = dq[1], az = dq[2], aw = dq[3], bx = q[0], by = q[1], bz = q[2], bw = q[3];
  out[0] = aw * bx + ax * bw + ay * bz - az * by;
  out[1] = aw * by - ax * bz + ay * bw + az * bx;
  out[2] = aw * bz + ax * by - ay * bx + az * bw;
  out[3] = aw * bw - ax * bx - ay * by - az * bz;
}

void quat_rot_h(const double q[4], const double p[3], double out[3]) {  // Eigen: quaternion * vector (x,y,z,w)
  const double tx = 2 * (q[1] * p[2] - q[2] * p[1]), ty = 2 * (q[2] * p[0] - q[0] * p[2]), tz = 2 * (q[0] * p[1] - q[1] * p[0]);
  out[0] = p[0] + q[3] * tx + (q[1] * tz - q[2] * ty);
  out[1] = p[1] + q[3] * ty + (q[2] * tx - q[0] * tz);
  out[2] = p[2] + q[3] * tz + (q[0] * ty - q[1] * tx);
}
void quat_mul_h(const double a[4], const double b[4], double o[4]) {
  o[0] = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
  o[1] = a[3] * b[1] - a[0] * b[2] + a[1] * b[3] + a[2] * b[0];
  o[2] = a[3] * b[2] + a[0] * b[1] - a[1] * b[0] + a[2] * b[3];
  o[3] = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
}
// Ground_DeltaFactor_goable::operator() (lidarFactor.hpp:357-391)
void ground_residual(const rgc_mapreg_ground* G, const double q[4], const double t[3], double r[3]) {
  const double lqc[4] = {-G->last_q[0], -G->last_q[1], -G->last_q[2], G->last_q[3]};
  double q_lc[4], dt[3] = {t[0] - G->last_t[0], t[1] - G->last_t[1], t[2] - G->last_t[2]}, t_lc[3], gn[3], delta_t[3];
  quat_mul_h(lqc, q, q_lc);
  quat_rot_h(lqc, dt, t_lc);
  quat_rot_h(q_lc, G->cur_norm, gn);
  quat_rot_h(G->q_history, t_lc, delta_t);
  const double dist_cur = G->cur_distance + delta_t[2];
  r[0] = (G->last_distance - dist_cur) / (G->p_var / 1000);
  r[1] = std::fabs(G->last_v1[0] * gn[0] + G->last_v1[1] * gn[1] + G->last_v1[2] * gn[2]) / (G->p_var * 10);
  r[2] = std::fabs(G->last_v2[0] * gn[0] + G->last_v2[1] * gn[1] + G->last_v2[2] * gn[2]) / (G->p_var * 10);
}
// the ground block of one pose added on the host (three scalars: not worth a launch).  NULL loss; the Jacobian on the local
// parameterisation by central differences (step 1e-6; Ceres differentiates abs() as sign(), which this matches away from 0).
void ground_terms(const rgc_mapreg_ground* G, const double q[4], const double t[3], bool want_H, double S28[28]) {
  if (!G) return;
  double r[3];
  ground_residual(G, q, t, r);
  S28[27] += 0.5 * (r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
  if (!want_H) return;
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
    for (int e = a; e < 6; e++) {
      double v = 0;
      for (int k = 0; k < 3; k++) v += J[k * 6 + a] * J[k * 6 + e];
      S28[u++] += v;
    }
  for (int a = 0; a < 6; a++) {
    double v = 0;
    for (int k = 0; k < 3; k++) v += J[k * 6 + a] * r[k];
    S28[21 + a] += v;
  }
}

// Quaternion2EulerAngle (lidarFactor.hpp:405-433) on x,y,z,w: pitch and roll only
void pitch_roll(const double q[4], double* pitch, double* roll) {
  const double sinp = 2 * (q[3] * q[1] - q[0] * q[2]);
  *pitch = sinp >= 1 ? M_PI / 2 : (sinp <= -1 ? -M_PI / 2 : std::asin(sinp));
  *roll = std::atan2(2 * (q[3] * q[0] + q[1] * q[2]), 1 - 2 * (q[0] * q[0] + q[1] * q[1]));
}
// RelativeRFactor on (q_last, q_cur) (lidarFactor.hpp:174-226; QuaternionInverse = conjugate, :124-130) followed by the
// PitchRollFactor of the current and of the last pose (:434-468): 3 + 2 + 2 residuals
void imu_residual(const rgc_mapreg_imu* I, const double q_cur[4], const double q_last[4], double r[7]) {
  const double li[4] = {-q_last[0], -q_last[1], -q_last[2], q_last[3]};
  const double di[4] = {-I->delta_q[0], -I->delta_q[1], -I->delta_q[2], I->delta_q[3]};
  double q_ij[4], e[4], p, ro;
  quat_mul_h(li, q_cur, q_ij);
  quat_mul_h(di, q_ij, e);
  for (int a = 0; a < 3; a++) r[a] = 2 * e[a] / I->imu_cov;
  pitch_roll(q_cur, &p, &ro);
  r[3] = 2 * (p - I->pitch_cur) / I->pr_var;
  r[4] = 2 * (ro - I->roll_cur) / I->pr_var;
  pitch_roll(q_last, &p, &ro);
  r[5] = 2 * (p - I->pitch_last) / I->pr_var;
  r[6] = 2 * (ro - I->roll_last) / I->pr_var;
}
// the IMU block (RGC_mapping.cpp:1285-1312) added on the host: seven scalars over the two rotations, NULL loss, Jacobian
// on the local parameterisation by central differences (step 1e-6) like the ground block.  H is the full 12 x 12.
void imu_terms(const rgc_mapreg_imu* I, const double x[14], bool want_H, double H[144], double g[12], double* cost) {
  if (!I) return;
  double r[7];
  imu_residual(I, x, x + 7, r);
  for (int k = 0; k < 7; k++) *cost += 0.5 * r[k] * r[k];
  if (!want_H) return;
  double J[7][12] = {};
  const double h = 1e-6;
  for (int b = 0; b < 2; b++)
    for (int a = 0; a < 3; a++) {  // the translations do not enter
      double rp[7], rm[7];
      for (int sgn = 0; sgn < 2; sgn++) {
        double qc[4], ql[4], d[3] = {0, 0, 0};
        memcpy(qc, x, sizeof(qc)); memcpy(ql, x + 7, sizeof(ql));
        d[a] = sgn ? -h : h;
        quat_plus(x + 7 * b, d, b ? ql : qc);
        imu_residual(I, qc, ql, sgn ? rm : rp);
      }
      for (int k = 0; k < 7; k++) J[k][6 * b + a] = (rp[k] - rm[k]) / (2 * h);
    }
  for (int a = 0; a < 12; a++) {
    for (int e = 0; e < 12; e++) {
      double v = 0;
      for (int k = 0; k < 7; k++) v += J[k][a] * J[k][e];
      H[a * 12 + e] += v;
    }
    double v = 0;
    for (int k = 0; k < 7; k++) v += J[k][a] * r[k];
    g[a] += v;
  }
}

// the robustified normal equations of both poses at x (14 doubles): H 12 x 12 (two 6 x 6 pose blocks, plus the IMU block's
// coupling of the rotations), g 12, the cost.  Feature sets 0/1 = corner/surf of the current pose, 2/3 = of the last pose;
// the kernels return 21 H + 6 g + cost per pose.
struct MapregSystem { double H[144], g[12], cost; };
int mapreg_eval(rgc_ctx* c, const int nfeat[4], const double x[14], bool want_H, const rgc_mapreg_ground* const ground[2],
                const rgc_mapreg_imu* imu, MapregSystem* out) {
  const float* feat[4];
  const double* fac[4];
  for (int s = 0; s < 4; s++) { feat[s] = (const float*)c->mr_feat[s].p; fac[s] = (const double*)c->mr_fac[s].p; }
  rgck::mapreg_terms(c->stream, feat, fac, nfeat, x, 0.1, want_H ? 1 : 0, (double*)c->mr_partials.p, c->d_out);
  HIPCHK(c, hipMemcpyAsync(c->h_out, c->d_out, sizeof(double) * 56, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipGetLastError());
  double S[2][28];
  memcpy(S, c->h_out, sizeof(S));
  for (int b = 0; b < 2; b++) ground_terms(ground[b], x + 7 * b, x + 7 * b + 4, want_H, S[b]);
  memset(out->H, 0, sizeof(out->H));
  memset(out->g, 0, sizeof(out->g));
  for (int b = 0; b < 2 && want_H; b++) {
    int u = 0;
    for (int a = 0; a < 6; a++)
      for (int e = a; e < 6; e++, u++) out->H[(6 * b + a) * 12 + 6 * b + e] = out->H[(6 * b + e) * 12 + 6 * b + a] = S[b][u];
    for (int a = 0; a < 6; a++) out->g[6 * b + a] = S[b][21 + a];
  }
  out->cost = S[0][27] + S[1][27];
  imu_terms(imu, x, want_H, out->H, out->g, &out->cost);
  return RGC_OK;
}

}  // namespace

// =================================================================================================
// C-ABI
// =================================================================================================
extern "C" {

void rgc_default_params(rgc_params* p) {
  if (!p) return;
  p->voxel_res = 1.0;
  p->max_iterations = 25;
  p->lm_max_iterations = 10;
  p->rotation_eps = 2e-3;
  p->translation_eps = 1e-6;
  p->lm_init_lambda_factor = 1e-9;
  p->k_correspondences = 20;
  p->neighbor_method = RGC_DIRECT1;
  p->max_cells = 1ll << 29;
}

const char* rgc_version(void) { return "rgc_hip 0.1 (gfx950)"; }

const char* rgc_status_string(int s) {
  switch (s) {
    case RGC_OK: return "ok";
    case RGC_ERR_INVALID: return "invalid argument";
    case RGC_ERR_HIP: return "HIP runtime error";
    case RGC_ERR_TOO_FEW_POINTS: return "too few points";
    case RGC_ERR_GRID_TOO_LARGE: return "grid too large";
    case RGC_ERR_NO_INPUT: return "no input cloud";
    case RGC_ERR_NONFINITE: return "non-finite input";
    case RGC_ERR_UNSUPPORTED: return "unsupported setting selected";
  }
  return "unknown";
}

int rgc_create(int hip_device, const rgc_params* params, rgc_ctx** out) {
  if (!out) return RGC_ERR_INVALID;
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || hipSetDevice(hip_device) != hipSuccess) return RGC_ERR_HIP;
  rgc_ctx* c = new (std::nothrow) rgc_ctx();
  if (!c) return RGC_ERR_HIP;
  c->device = hip_device;
  rgc_default_params(&c->prm);
  if (params) {
    int rc = check_params(c, params);
    if (rc) { delete c; return rc; }
    c->prm = *params;
  }
  bool ok = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) == hipSuccess;
  {  // the source's small kernels must not queue behind the map's 15k-wave kNN launch: highest stream priority
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    ok = ok && hipStreamCreateWithPriority(&c->stream2, hipStreamNonBlocking, hi) == hipSuccess;
  }
  ok = ok && hipEventCreateWithFlags(&c->lm_mid, kDevEvent) == hipSuccess;
  ok = ok && hipEventCreateWithFlags(&c->src_in_ready, kDevEvent) == hipSuccess;
  ok = ok && hipEventCreateWithFlags(&c->src_ready, kDevEvent) == hipSuccess;
  ok = ok && hipEventCreateWithFlags(&c->main_mark, kDevEvent) == hipSuccess;
  ok = ok && hipEventCreateWithFlags(&c->tgt_ready, kDevEvent) == hipSuccess;
  ok = ok && hipMalloc((void**)&c->d_small, 48 * sizeof(int)) == hipSuccess;
  ok = ok && hipMalloc((void**)&c->d_out, 64 * sizeof(double)) == hipSuccess;
  ok = ok && hipHostMalloc((void**)&c->h_small, 48 * sizeof(int), hipHostMallocDefault) == hipSuccess;
  ok = ok && hipHostMalloc((void**)&c->h_out, 64 * sizeof(double), hipHostMallocDefault) == hipSuccess;
  ok = ok && hipHostMalloc((void**)&c->h_lm, sizeof(rgck::LmState), hipHostMallocDefault) == hipSuccess;
  ok = ok && hipHostMalloc((void**)&c->h_post, sizeof(rgck::LmState), hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess;
  if (ok) {
    memset(c->h_post, 0, sizeof(rgck::LmState));
    if (hipHostGetDevicePointer((void**)&c->d_post, c->h_post, 0) != hipSuccess) c->d_post = nullptr;  // (no fast path then)
  }
  if (ok && hipHostMalloc((void**)&c->h_early, sizeof(rgck::LmEarly), hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess) {
    memset(c->h_early, 0, sizeof(rgck::LmEarly));
    if (hipHostGetDevicePointer((void**)&c->d_early, c->h_early, 0) != hipSuccess) c->d_early = nullptr;  // (no early pose then)
  }
  c->uid = g_next_uid.fetch_add(1);
  ok = ok && hipEventCreateWithFlags(&c->src_read_done, kDevEvent) == hipSuccess;
  ok = ok && hipEventCreateWithFlags(&c->lm_tail, kDevEvent) == hipSuccess;
  ok = ok && hipEventCreateWithFlags(&c->tgt_prepared, kDevEvent) == hipSuccess;
  ok = ok && hipEventCreateWithFlags(&c->vg_done, hipEventDisableTiming) == hipSuccess;
  ok = ok && hipHostMalloc((void**)&c->h_vg, 4 * sizeof(int), hipHostMallocDefault) == hipSuccess;
  if (const char* e = getenv("RGC_SPEC_GRID")) c->spec_on = atoi(e) != 0;
  if (const char* e = getenv("RGC_KNN_SEEDS")) c->seeds_on = atoi(e) != 0;
  if (const char* e = getenv("RGC_KNN_CACHE")) c->cache_on = atoi(e) != 0;
  if (!c->seeds_on) c->cache_on = false;  // (the lists sit on top of the seeds)
  c->trace_cache = getenv("RGC_TRACE_CACHE") != nullptr;
  if (const char* e = getenv("RGC_CHECK_POINTERS")) c->check_ptrs = atoi(e) != 0;
  if (const char* e = getenv("RGC_FORCE_GENERAL")) c->force_general = atoi(e) != 0;
  c->test_fail_cache_alloc = getenv("RGC_TEST_FAIL_CACHE_ALLOC") != nullptr;
  if (const char* e = getenv("RGC_JOIN_SPIN_US")) c->join_spin_us = atoi(e);
  if (const char* e = getenv("RGC_PREP_EVENT_EXT")) c->prep_event_ext = atoi(e) != 0;
  if (const char* e = getenv("RGC_COOP_STREAM")) c->coop_stream_on = atoi(e) != 0;
  if (const char* e = getenv("RGC_LM_IMPL")) c->lm_host = strcmp(e, "host") == 0;
  if (!ok) { rgc_destroy(c); return RGC_ERR_HIP; }
  { std::lock_guard<std::mutex> lk(g_live_mutex); g_live.insert(c); }
  *out = c;
  return RGC_OK;
}

void rgc_destroy(rgc_ctx* c) {
  if (!c) return;
  { std::lock_guard<std::mutex> lk(g_live_mutex); g_live.erase(c); }
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->stream2) (void)hipStreamSynchronize(c->stream2);
  prof_collect(c);
  for (auto e : c->ev_pool) (void)hipEventDestroy(e);
  release_cloud(c->src);
  release_cloud(c->tgt);
  release_cloud(c->aux);
  release_cloud(c->mr_map[0]);
  release_cloud(c->mr_map[1]);
  for (DevBuf& b : c->mr_feat) release(b);
  for (DevBuf& b : c->mr_fac) release(b);
  release(c->mr_partials);
  release(c->mr_small);
  for (DevBuf& b : c->fe) release(b);
  for (DevBuf* b : {&c->map_store[0], &c->map_store[1], &c->map_target}) release(*b);
  for (DevBuf* b : {&c->corr_v, &c->corr_M, &c->corr_v2, &c->corr_M2, &c->partials, &c->ipartials, &c->scratch, &c->pre_in, &c->pre_out, &c->vg_order, &c->vg_pos, &c->vg_tmp, &c->vg_leaf}) release(*b);
  if (c->d_small) (void)hipFree(c->d_small);
  if (c->d_out) (void)hipFree(c->d_out);
  if (c->h_small) (void)hipHostFree(c->h_small);
  if (c->h_out) (void)hipHostFree(c->h_out);
  if (c->h_lm) (void)hipHostFree(c->h_lm);
  if (c->h_post) (void)hipHostFree(c->h_post);
  if (c->h_early) (void)hipHostFree(c->h_early);
  if (c->h_stage) (void)hipHostFree(c->h_stage);
  release(c->lm_state);
  release(c->fit_partials);
  if (c->vg_done) (void)hipEventDestroy(c->vg_done);
  if (c->h_vg) (void)hipHostFree(c->h_vg);
  if (c->src_ready) (void)hipEventDestroy(c->src_ready);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  if (c->main_mark) (void)hipEventDestroy(c->main_mark);
  if (c->tgt_ready) (void)hipEventDestroy(c->tgt_ready);
  if (c->src_read_done) (void)hipEventDestroy(c->src_read_done);
  if (c->lm_tail) (void)hipEventDestroy(c->lm_tail);
  if (c->lm_mid) (void)hipEventDestroy(c->lm_mid);
  if (c->src_in_ready) (void)hipEventDestroy(c->src_in_ready);
  if (c->tgt_prepared) (void)hipEventDestroy(c->tgt_prepared);
  if (c->stream2) (void)hipStreamDestroy(c->stream2);
  delete c;
}

int rgc_set_params(rgc_ctx* c, const rgc_params* p) {
  if (!c || !p) return RGC_ERR_INVALID;
  int rc = check_params(c, p);
  if (rc) return rc;
  const bool redo = p->voxel_res != c->prm.voxel_res || p->k_correspondences != c->prm.k_correspondences;
  if (redo && solve_in_flight(c)) return fail(c, RGC_ERR_INVALID, "a solve is in flight on this context: call rgc_align_end first");  // (the clouds are prepared again below)
  if (redo) HIPCHK(c, hipSetDevice(c->device));  // (the re-preparation launches kernels: like every entry point that does, whatever device the calling thread had current)
  if (p->voxel_res != c->prm.voxel_res) c->src_res_auto = 0.0;
  c->prm = *p;
  c->corr_valid = false;
  if (redo) {  // covariances / voxel map depend on these: recompute from the resident inputs
    if (c->tgt_owner) {
      // a BORROWED target (rgc_share_target) is the owner's, prepared under the owner's settings, and its input buffer is the owner's to
      // keep or free: nothing to recompute from here (tests/fuzz/fuzz_api.py: the re-preparation read an input the owner had long replaced --
      // a memory fault on the device).  The alias goes; share again once the owner holds a target under these settings.
      release_cloud(c->tgt);
      c->tgt.ready = false; c->tgt.n = 0; c->tgt.in = nullptr;
      c->tgt_owner = nullptr;
    }
    if (c->src.ready) { c->src.ready = false; if ((rc = prepare_cloud(c, c->src, false))) return rc; }
    if (c->tgt.ready) { c->tgt.ready = false; if ((rc = prepare_cloud(c, c->tgt, true))) return rc; }
  }
  return RGC_OK;
}

int rgc_get_params(const rgc_ctx* c, rgc_params* p) {
  if (!c || !p) return RGC_ERR_INVALID;
  *p = c->prm;
  return RGC_OK;
}

const char* rgc_last_error(const rgc_ctx* c) { return c ? c->err : "null context"; }

int rgc_set_target(rgc_ctx* c, const float* xyz, int n, int stride_bytes) { return c ? set_cloud(c, c->tgt, true, xyz, n, stride_bytes, false) : RGC_ERR_INVALID; }
int rgc_set_source(rgc_ctx* c, const float* xyz, int n, int stride_bytes) { return c ? set_cloud(c, c->src, false, xyz, n, stride_bytes, false) : RGC_ERR_INVALID; }
int rgc_set_target_device(rgc_ctx* c, const float* xyz, int n, int stride_bytes) { return c ? set_cloud(c, c->tgt, true, xyz, n, stride_bytes, true) : RGC_ERR_INVALID; }
int rgc_set_source_device(rgc_ctx* c, const float* xyz, int n, int stride_bytes) { return c ? set_cloud(c, c->src, false, xyz, n, stride_bytes, true) : RGC_ERR_INVALID; }

static int fetch_nvox(rgc_ctx* c);

// A second context registers scans to the SAME prepared target (a resident map) without preparing or copying it: its target
// becomes a non-owning alias of the owner's buffers.  What it is for: two contexts taking turns on a sequence whose map does not
// change from frame to frame -- the next scan is prepared on one while the current one is solved on the other (PipelinedVGICP).
int rgc_share_target(rgc_ctx* c, rgc_ctx* owner) {
  if (!c || !owner || c == owner) return RGC_ERR_INVALID;
  if (c->device != owner->device) return fail(c, RGC_ERR_INVALID, "rgc_share_target: the contexts are on different devices");
  if (c->lm_host || owner->lm_host) return fail(c, RGC_ERR_INVALID, "rgc_share_target is not available with RGC_LM_IMPL=host");
  if (!owner->tgt.ready) return fail(c, RGC_ERR_NO_INPUT, "rgc_share_target: the owner has no target");
  HIPCHK(c, hipSetDevice(c->device));
  int rc = validate_clouds(owner);  // a speculative grid the owner's target did not fit is resolved now (one synchronisation)
  if (rc) return fail(c, rc, "rgc_share_target: %s", owner->err);
  if (!owner->tgt.ready) return fail(c, RGC_ERR_NO_INPUT, "rgc_share_target: the owner has no target");
  if ((rc = fetch_nvox(owner))) return fail(c, rc, "rgc_share_target: %s", owner->err);
  HIPCHK(c, hipStreamSynchronize(owner->stream2));
  HIPCHK(c, hipStreamSynchronize(owner->stream));   // the target is complete in memory
  HIPCHK(c, hipStreamSynchronize(c->stream2));
  HIPCHK(c, hipStreamSynchronize(c->stream));       // nothing of this context still reads its old target
  Cloud& d = c->tgt;
  const Cloud& o = owner->tgt;
  release_cloud(d);
  DevBuf* db[] = {&d.in_copy, &d.cell_of, &d.slot_of, &d.cnt, &d.start, &d.block_sums, &d.order_tmp, &d.P, &d.nx, &d.ny, &d.nz, &d.segs,
                  &d.cell_voxel, &d.vox, &d.vox_cell, &d.c6};
  const DevBuf* ob[] = {&o.in_copy, &o.cell_of, &o.slot_of, &o.cnt, &o.start, &o.block_sums, &o.order_tmp, &o.P, &o.nx, &o.ny, &o.nz, &o.segs,
                        &o.cell_voxel, &o.vox, &o.vox_cell, &o.c6};
  d.general = o.general;
  for (size_t k = 0; k < sizeof(db) / sizeof(db[0]); k++) { db[k]->p = ob[k]->p; db[k]->cap = ob[k]->cap; db[k]->borrowed = ob[k]->p != nullptr; }
  d.in = o.in; d.stride_f = o.stride_f; d.n = o.n; d.grid = o.grid; d.grid = o.grid; d.nvox = o.nvox; d.deferred_seen = o.deferred_seen;
  d.spec_ok = false; d.spec_used = false; d.cnt_clean = 0; d.cnt_seen = nullptr; d.lazy = 0;
  d.ready = true;
  const int small[2] = {0, o.nvox};  // this context's copy of the target's guard (clear) and voxel count, which the solve reads
  HIPCHK(c, hipMemcpyAsync(c->d_small + 6, small, sizeof(small), hipMemcpyHostToDevice, c->stream));  // (not the blocking form: it goes through the NULL stream)
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->tgt_owner = owner;
  c->tgt_owner_gen = owner->tgt_generation;
  c->tgt_owner_uid = owner->uid;
  c->map_bound = false;
  c->corr_valid = false;
  c->deferred_known = false;
  c->main_has_target_prep = false;
  c->stats.n_target = o.n; c->stats.target_cells = o.grid.ncell; c->stats.n_voxels = o.nvox;
  return RGC_OK;
}

// Two contexts taking turns on a DEPENDENT sequence (each frame's target is a function of the previous pose, RGC_odometer.cpp:1248-1256):
// the next scan can be prepared ahead -- it depends on no pose -- but enqueued beside the current frame's map preparation its small
// kernels share the chip with the 15 k-wave kNN launch the frame is waiting for (207 us instead of 155).  Held back until that
// preparation is done, they run under the current frame's SOLVE, a chain of short launches that leaves the chip mostly idle.
int rgc_set_target_lazy(rgc_ctx* c, int margin_cells) {
  if (!c) return RGC_ERR_INVALID;
  if (margin_cells < 0 || margin_cells > 16) return fail(c, RGC_ERR_INVALID, "rgc_set_target_lazy: margin_cells must be in [0, 16]");
  if (solve_in_flight(c)) return fail(c, RGC_ERR_INVALID, "a solve is in flight on this context: call rgc_align_end first");
  c->lazy_margin = margin_cells;  // (takes effect with the next target; one already set keeps the state it is in)
  return RGC_OK;
}

int rgc_set_regularization_method(rgc_ctx* c, int method) {
  if (!c) return RGC_ERR_INVALID;
  if (method < RGC_REG_NONE || method > RGC_REG_FROBENIUS) return fail(c, RGC_ERR_INVALID, "rgc_set_regularization_method: %d is not a RegularizationMethod", method);
  if (method != c->reg_method) {  // the covariances of the clouds set so far were computed under the other method (the reference computes them at align())
    if (solve_in_flight(c)) return fail(c, RGC_ERR_INVALID, "a solve is in flight on this context: call rgc_align_end first");
    c->src.ready = c->tgt.ready = false; c->corr_valid = false; c->deferred_known = false; c->map_bound = false;
  }
  c->reg_method = method;
  return RGC_OK;
}

int rgc_set_voxel_accumulation_mode(rgc_ctx* c, int mode) {
  if (!c) return RGC_ERR_INVALID;
  if (mode < RGC_VOXEL_ADDITIVE || mode > RGC_VOXEL_MULTIPLICATIVE) return fail(c, RGC_ERR_INVALID, "rgc_set_voxel_accumulation_mode: %d is not a VoxelAccumulationMode", mode);
  const bool was = c->voxel_mode == RGC_VOXEL_MULTIPLICATIVE, is = mode == RGC_VOXEL_MULTIPLICATIVE;
  if (was != is) {  // (ADDITIVE <-> ADDITIVE_WEIGHTED changes nothing: one voxel class in the vendored FastVGICP, fast_vgicp_voxel.hpp:137-141)
    if (solve_in_flight(c)) return fail(c, RGC_ERR_INVALID, "a solve is in flight on this context: call rgc_align_end first");
    c->src.ready = c->tgt.ready = false; c->corr_valid = false; c->deferred_known = false; c->map_bound = false;
  }
  c->voxel_mode = mode;
  return RGC_OK;
}

int rgc_set_knn_reuse(rgc_ctx* c, int mode) {
  if (!c) return RGC_ERR_INVALID;
  if (mode < RGC_REUSE_NONE || mode > RGC_REUSE_LISTS) return fail(c, RGC_ERR_INVALID, "rgc_set_knn_reuse: mode must be RGC_REUSE_NONE, _SEEDS or _LISTS");
  if (solve_in_flight(c)) return fail(c, RGC_ERR_INVALID, "a solve is in flight on this context: call rgc_align_end first");
  const bool seeds = mode >= RGC_REUSE_SEEDS && RGC_KNN_SEEDS != 0, lists = seeds && mode >= RGC_REUSE_LISTS && RGC_KNN_CACHE != 0;
  Cloud& cl = c->tgt;
  if ((!lists && cl.nbr.p) || (!seeds && cl.seed.p)) {  // buffers this context no longer needs: nothing may still be reading them
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream2));
  }
  if (!lists) {
    for (DevBuf* b : {&cl.nbr, &cl.pos_of, &cl.rank_of, &cl.qrank, &cl.map_copy, &cl.todo, &cl.cache_small}) release(*b);
    cl.cache_on = cl.cache_live = cl.cache_searched_lists = false;
    cl.rf.copy = nullptr; cl.rf.epoch = nullptr;
  }
  if (!seeds) {
    release(cl.seed);
    cl.seed_key = nullptr; cl.seed_n = 0;
    cl.seed_on = cl.seed_warm = false;
  }
  c->seeds_on = seeds;
  c->cache_on = lists;
  c->cache_dropped = false;
  return RGC_OK;
}

int rgc_get_knn_reuse(const rgc_ctx* c, int* mode) {
  if (!c || !mode) return RGC_ERR_INVALID;
  *mode = !c->seeds_on ? RGC_REUSE_NONE : (c->cache_on ? RGC_REUSE_LISTS : RGC_REUSE_SEEDS);
  return RGC_OK;
}

int rgc_hold_source_until_target_of(rgc_ctx* c, rgc_ctx* other) {
  if (!c || !other) return RGC_ERR_INVALID;
  // (the other context must be alive while its event is handed to the runtime: checked and used under the registry's lock)
  std::lock_guard<std::mutex> lk(g_live_mutex);
  if (!g_live.count(other)) return fail(c, RGC_ERR_INVALID, "rgc_hold_source_until_target_of: the other context has been destroyed");
  if (c->device != other->device) return fail(c, RGC_ERR_INVALID, "rgc_hold_source_until_target_of: the contexts are on different devices");
  HIPCHK(c, hipSetDevice(c->device));
  // the scan's stream waits for the end of other's latest target preparation (a wait on an event nobody recorded yet is no wait)
  HIPCHK(c, hipStreamWaitEvent(c->stream2, other->tgt_prepared, 0));
  return RGC_OK;
}

int rgc_linearize(rgc_ctx* c, const double T[16], double H[36], double b[6], double* cost) {
  if (!c || !T) return RGC_ERR_INVALID;
  HIPCHK(c, hipSetDevice(c->device));
  return do_linearize(c, T, H, b, cost);
}

int rgc_compute_error(rgc_ctx* c, const double T[16], double* cost) {
  if (!c || !T || !cost) return RGC_ERR_INVALID;
  HIPCHK(c, hipSetDevice(c->device));
  return do_error(c, T, cost);
}

int rgc_num_correspondences(rgc_ctx* c, int* n) {
  if (!c || !n) return RGC_ERR_INVALID;
  if (!c->corr_valid) return fail(c, RGC_ERR_INVALID, "no linearisation yet");
  *n = c->stats.n_corr;
  return RGC_OK;
}

// One batch of blind LM steps (the first launch of a solve opens it), the fitness kernel behind them and the state's read-back:
// enqueued, not waited for.
static int lm_enqueue_batch(rgc_ctx* c, int batch, const rgck::LmInit* open, bool want_fitness) {
  const int n = c->src.n, noff = noff_of(c->prm.neighbor_method);
  hipStream_t s = c->solve_stream;  // see rgc_align_begin
  rgck::LmState* post = (c->post_on && c->d_post) ? c->d_post : nullptr;
  const int seq = want_fitness ? -c->lm_seq : c->lm_seq;  // what is posted: a finished state (> 0), or a finished state with its score (< 0)
  constexpr int kSpare = 3;
  // the stage-by-stage pass (events around the solve's regions) keeps the two apart: all steps, then the score
  const bool staged = c->prof_on && ((c->prof_mask >> RGC_K_LINEARIZE) & 1u || (c->prof_mask >> RGC_K_FITNESS) & 1u);
  // The score is chained INTO the steps: the launch whose decision ends the solve scores the final pose and posts the result (k_lm_step).
  // No separate score launches, no blind steps between the deciding launch and the score.
  const bool fit_in_steps = want_fitness && !staged;
  if (open) c->lm_j = 0;
  auto step = [&](const rgck::LmInit* op, hipStream_t on) {
    rgck::lm_step(on, (const float4*)c->src.P.p, (const double*)c->src.nx.p, (const double*)c->src.ny.p, (const double*)c->src.nz.p, n,
                  c->tgt.grid, (const int*)c->tgt.cell_voxel.p, (const double*)c->tgt.vox.p, noff, (int*)c->corr_v.p, (double*)c->corr_M.p,
                  (int*)c->corr_v2.p, (double*)c->corr_M2.p, (double*)c->partials.p, (rgck::LmState*)c->lm_state.p, c->lm_j++, op, c->d_small + 7,
                  c->tgt.segs.p, c->src.segs.p, post, seq, fit_in_steps ? (const float4*)c->tgt.P.p : nullptr,
                  fit_in_steps ? (const int*)c->tgt.start.p : nullptr, fit_in_steps ? (double*)c->fit_partials.p : nullptr, c->tgt.n,
                  c->tgt.lazy == 2 ? (const int*)c->tgt.need.p : nullptr, c->tgt.need_stamp, c->tgt.lazy == 2 ? (const int*)c->tgt.segs.p + 1 : nullptr,
                  (fit_in_steps && post) ? c->d_early : nullptr);
  };
  auto score = [&]() {  // getFitnessScore at the final pose, chained blindly (on the image the last launch left)
    rgck::fitness_lm(s, (const float4*)c->src.P.p, n, rgck::lm_image((rgck::LmState*)c->lm_state.p, c->lm_j - 1), (const float4*)c->tgt.P.p,
                     (const int*)c->tgt.start.p, c->tgt.grid, (double*)c->fit_partials.p, post, c->lm_seq, c->tgt.n);
  };
  hipStream_t tail = s;
  if (staged) {
    {
      ProfScope ps(c, RGC_K_LINEARIZE, (long long)n * batch, s);
      for (int k = 0; k < batch; k++) { step(open, s); open = nullptr; }
    }
    if (want_fitness) {
      ProfScope ps(c, RGC_K_FITNESS, n, s);
      score();
    }
  } else {
    // The launches a solve is EXPECTED to need go to the solve's stream; the spare ones -- enqueued blind in case it needs more: each costs
    // ~5 us of its stream's time even when it finds the solve finished -- go to the context's OTHER stream behind an event, where they
    // drain beside whatever the caller enqueues next on the solve's stream (the next frame's map preparation) instead of in front of it.
    // (Not a third stream: two contexts with three streams each outnumber the hardware queues, and streams that share a queue serialise --
    // the two-context sequence lost 130 us per frame that way.)
    const int spare = (RGC_LM_SPARE_ASIDE && batch > kSpare + 1) ? kSpare : 0;
    hipStream_t other = s == c->stream ? c->stream2 : c->stream;
    for (int k = 0; k < batch - spare; k++) { step(open, s); open = nullptr; }
    if (spare > 0) {
      HIPCHK(c, hipEventRecord(c->lm_mid, s));
      HIPCHK(c, hipStreamWaitEvent(other, c->lm_mid, 0));
      for (int k = 0; k < spare; k++) step(nullptr, other);
      tail = other;
    }
  }
  // (a solve that POSTS its finished state needs no stream-ordered copy of it: the one case that reads the state otherwise -- a batch that
  // ends without a finished solve -- fetches it with a blocking copy once the stream has drained, rgc_align_end)
  if (!post) HIPCHK(c, hipMemcpyAsync(c->h_lm, rgck::lm_image((rgck::LmState*)c->lm_state.p, c->lm_j - 1), sizeof(rgck::LmState), hipMemcpyDeviceToHost, tail));
  HIPCHK(c, hipEventRecord(c->lm_tail, tail));
  c->lm_tail_stream = tail;
  return RGC_OK;
}

// The solve in two halves, so that a caller with two contexts can prepare the clouds of the next frame (on the other context)
// while this one's LM runs: rgc_align_begin enqueues the device-chained LM behind the clouds' preparation and returns;
// rgc_align_end waits for it (and enqueues further batches if the solve needs more than six outer iterations).
int rgc_align_begin(rgc_ctx* c, const float guess[16], int want_fitness) {
  if (!c || !guess) return RGC_ERR_INVALID;
  if (solve_in_flight(c)) return fail(c, RGC_ERR_INVALID, "a solve is in flight on this context: call rgc_align_end first");
  if (general_route(c)) {  // no asynchronous form on the general route: the solve runs now, rgc_align_end hands its result over
    auto& g = c->gen_res;
    g.on = false;
    g.rc = rgc_align(c, guess, g.T, g.H, want_fitness ? &g.fitness : nullptr, &g.iterations, &g.converged, &g.lm_failed);
    if (g.rc) return g.rc;
    g.has_fit = want_fitness != 0;
    g.on = true;
    return RGC_OK;
  }
  if (c->lm_host) return fail(c, RGC_ERR_INVALID, "the host-driven LM loop (RGC_LM_IMPL=host) has no asynchronous form");
  HIPCHK(c, hipSetDevice(c->device));
  c->pend.active = false;
  { const int rs = check_supported(c); if (rs) return rs; }
  // the guards of speculative grids come home with the LM state: no synchronisation here
  if (!c->src.ready || !c->tgt.ready) return fail(c, RGC_ERR_NO_INPUT, "source and target must be set first");
  { int rc = check_target_owner(c); if (rc) return rc; }
  if (c->tgt.lazy == 1) { int rc = lazy_build(c, guess); if (rc) return rc; }        // lazy target: built where this solve can look
  else if (c->tgt.lazy == 2) { int rc = complete_target(c); if (rc) return rc; }      // ... a second solve on the same target: all of it
  // The solve is a chain of short launches: it runs on the HIGH-PRIORITY stream -- the one the scan was prepared on, so it is already
  // behind that -- ordered after the map's preparation on the main stream by one event.  With a second context preparing the next
  // frame's map meanwhile (15 k waves that fill every CU), the dispatcher places the solve's ~100 workgroups as soon as slots free up
  // instead of behind that launch.
  int rc;
  bool join_late = false;
  if (c->solve_behind_map && c->main_has_target_prep && !map_prep_finished(c)) {
    // the map is still being prepared (a dependent sequence: it could only start when the previous pose was known): the solve goes
    // directly behind it on the main stream -- a dependency that resolves across streams costs ~10 us on this runtime, and the scan's
    // preparation, which the solve also waits for, has long finished
    c->solve_stream = c->stream;
    join_late = true;  // (join_source may wait on the host for an almost-ready scan: done last, right in front of the first launch)
  } else {
    c->solve_stream = c->stream2;
    HIPCHK(c, hipEventRecord(c->tgt_ready, c->stream));
    HIPCHK(c, hipStreamWaitEvent(c->stream2, c->tgt_ready, 0));
  }
  // rgc_align_end returns as soon as the device POSTS the finished state: the previous solve's spare steps, score launches and the copy
  // into h_lm may still be queued on the stream they were enqueued on.  A solve that goes to the other stream is ordered behind them
  // (they work on the same LM state images and rows); a wait on an event that has already fired costs nothing.
  if (c->lm_tail_stream && c->lm_tail_stream != c->solve_stream && hipEventQuery(c->lm_tail) != hipSuccess) {
    (void)hipGetLastError();
    HIPCHK(c, hipStreamWaitEvent(c->solve_stream, c->lm_tail, 0));
  }
  const rgc_params& P = c->prm;
  // device-chained LM: the loop of :65-75 / :125-172 runs as a state machine on the device (k_lm_step);
  // the host only enqueues slots and reads the state back once per batch.
  const int n = c->src.n, noff = noff_of(P.neighbor_method);
  if ((rc = ensure(c, c->corr_v, sizeof(int) * (size_t)n * noff))) return rc;
  if ((rc = ensure(c, c->corr_M, sizeof(double) * 6 * (size_t)n * noff))) return rc;
  const int nb = rgck::linearize_blocks(n);
  if ((rc = ensure(c, c->corr_v2, sizeof(int) * (size_t)n * noff))) return rc;
  if ((rc = ensure(c, c->corr_M2, sizeof(double) * 6 * (size_t)n * noff))) return rc;
  if ((rc = ensure(c, c->partials, sizeof(double) * (rgck::kAccum + 2) * (size_t)nb * 2))) return rc;  // (two halves: rgck::lm_step)
  if ((rc = ensure(c, c->ipartials, sizeof(int) * (size_t)nb))) return rc;
  if (!c->lm_state.p) {
    if ((rc = ensure(c, c->lm_state, 4096))) return rc;
    HIPCHK(c, hipMemsetAsync(c->lm_state.p, 0, 4096, c->solve_stream));  // the score's ticket and the lazy target's miss flag start at 0
  }
  if ((rc = ensure(c, c->fit_partials, sizeof(double) * (size_t)rgck::fitness_blocks(n) + 64))) return rc;
  rgck::LmInit in;
  for (int i = 0; i < 12; i++) in.x0[i] = (double)guess[i];
  in.x0[12] = in.x0[13] = in.x0[14] = 0.0;
  in.x0[15] = 1.0;
  in.rot_eps = P.rotation_eps; in.trans_eps = P.translation_eps; in.init_factor = P.lm_init_lambda_factor;
  in.max_outer = P.max_iterations; in.max_inner = P.lm_max_iterations;
  c->stats.n_linearize = c->stats.n_error = c->stats.outer_iterations = 0;
  // The launches are enqueued blind: enough for the outer iterations the PREVIOUS solve on this context took (consecutive frames of a
  // sequence need about the same number; a launch on a finished solve costs ~5 us of its stream's time, a read-back and a second batch
  // ~40), at least the six that cover a tracking frame, at most what max_iterations allows.
  // (a solve of o outer iterations without a rejected try needs o + 2 launches: the opening linearisation, one per try, and the one
  // whose decision ends it and scores the pose.  That many plus one stay on the solve's stream -- a launch too many there costs ~5 us in
  // front of the next frame, one too few a ~13 us hop to the other stream --, three more go aside as spares: lm_enqueue_batch)
  int batch = 9;
  if (c->lm_last_outer + 6 > batch) batch = c->lm_last_outer + 6;
  if (batch > P.max_iterations + 2) batch = P.max_iterations + 2;
  if (batch < 2) batch = 2;
  c->lm_seq = c->lm_seq >= 0x3fffffff ? 1 : c->lm_seq + 1;
  if (join_late && (rc = join_source(c))) return rc;
  if ((rc = lm_enqueue_batch(c, batch, &in, want_fitness != 0))) return rc;
  memcpy(c->pend.guess, guess, sizeof(c->pend.guess));
  c->pend.want_fitness = want_fitness != 0;
  c->pend.active = true;
  return RGC_OK;
}

int rgc_align_end(rgc_ctx* c, float final_T[16], double final_H[36], double* fitness, int* iterations, int* converged, int* lm_failed) {
  if (!c) return RGC_ERR_INVALID;
  HIPCHK(c, hipSetDevice(c->device));  // (before the general route's score below, too)
  if (c->gen_res.on) {  // general route: solved in rgc_align_begin
    auto& g = c->gen_res;
    g.on = false;
    if (final_T) memcpy(final_T, g.T, sizeof(g.T));
    if (final_H) memcpy(final_H, g.H, sizeof(g.H));
    if (iterations) *iterations = g.iterations;
    if (converged) *converged = g.converged;
    if (lm_failed) *lm_failed = g.lm_failed;
    if (fitness) { if (g.has_fit) *fitness = g.fitness; else return do_fitness(c, g.T, fitness); }
    return RGC_OK;
  }
  if (!c->pend.active) return fail(c, RGC_ERR_INVALID, "rgc_align_end without rgc_align_begin");
  c->pend.active = false;
  int rc;
  const bool want_fitness = c->pend.want_fitness;
  rgck::LmState& S = c->lm_res;  // (not h_lm itself: the stream's copy into it may still be in flight when a posted state is taken)
  for (int guard = 0;; guard++) {
    // The device posts a finished solve's state into mapped host memory and then the solve's number: the host spins on that word and
    // leaves as soon as it shows up -- the blind launches behind the deciding one and the stream's copy drain in the background.
    // A batch that ends without a finished solve shows up as a drained stream: then the copy has landed and more launches are enqueued.
    bool posted = false;
    if (c->post_on && c->d_post) {
      volatile int* gen = &c->h_post->gen;
      for (unsigned spin = 0;; spin++) {
        if (*gen == c->lm_seq) { posted = true; break; }
        if (spin & 31u) continue;  // the posted word is the usual way out: look at it often, at the event now and then
        const hipError_t q = hipEventQuery(c->lm_tail);  // (recorded behind the batch's last launch and its copy of the state, on whichever stream they went to)
        if (q == hipSuccess) { posted = *gen == c->lm_seq; break; }
        if (q != hipErrorNotReady) return fail(c, RGC_ERR_HIP, "hipEventQuery failed: %s", hipGetErrorString(q));
        (void)hipGetLastError();
      }
      if (posted) {
        std::atomic_thread_fence(std::memory_order_acquire);
        memcpy(&S, c->h_post, sizeof(S));
      }
    }
    if (!posted) {
      HIPCHK(c, hipStreamSynchronize(c->lm_tail_stream));
      if (c->post_on && c->d_post) {  // (no copy was chained)
        // on the solve's own stream, never with the blocking hipMemcpy: that one goes through the NULL stream, whose hardware queue the
        // runtime creates at its first use -- 9 ms inside whichever frame first needed more launches than its batch held (the node's
        // "one slow frame" of rounds 3 and 4: frame 14 of the c2 stand-in)
        HIPCHK(c, hipMemcpyAsync(c->h_lm, rgck::lm_image((rgck::LmState*)c->lm_state.p, c->lm_j - 1), sizeof(rgck::LmState), hipMemcpyDeviceToHost,
                                 c->lm_tail_stream));
        HIPCHK(c, hipStreamSynchronize(c->lm_tail_stream));
      }
      memcpy(&S, c->h_lm, sizeof(S));
    }
    HIPCHK(c, hipGetLastError());
    if (S.done || guard >= 400) break;
    // a solve that is still running after six outer iterations usually runs many more (up to 25): batches of six, fewer read-backs
    if ((rc = lm_enqueue_batch(c, 6, nullptr, want_fitness))) return rc;
  }
  c->src_pending = false;  // the solve came after the scan's preparation and has finished (its spare launches may still drain: lm_tail)
  c->main_has_target_prep = false;  // (the solve came after the map's preparation)
  c->small_clean[0] = c->small_clean[1] = true;  // the solve's first step re-initialised both blocks after capturing them
  {  // a cloud that did not fit its speculative grid: everything above ran on a parked cloud -- prepare it properly, solve again
    const int r = resolve_guards(c, S.pad & 0xff, (S.pad >> 8) & 0xff);
    if (r < 0) return r;
    if (r > 0) {
      float guess[16];
      memcpy(guess, c->pend.guess, sizeof(guess));
      return rgc_align(c, guess, final_T, final_H, fitness, iterations, converged, lm_failed);
    }
  }
  if (c->tgt.lazy == 2 && S.pad2) {
    // lazy target: a look-up landed on an occupied voxel outside the part that was built (the pose moved further from the guess than the
    // margin covers).  The map is completed and the solve repeated from the same guess: the full build's result, bit for bit.
    c->stats.lazy_misses++;
    int rc2 = complete_target(c);
    if (rc2) return rc2;
    float guess[16];
    memcpy(guess, c->pend.guess, sizeof(guess));
    return rgc_align(c, guess, final_T, final_H, fitness, iterations, converged, lm_failed);
  }
  const int n = c->src.n, noff = noff_of(c->prm.neighbor_method);
  if (S.cur) { std::swap(c->corr_v, c->corr_v2); std::swap(c->corr_M, c->corr_M2); }  // corr_v / corr_M = the valid buffer
  c->corr_noff = noff; c->corr_n = n; c->corr_valid = S.n_lin > 0;
  c->stats.n_corr = S.ncorr; c->stats.n_linearize = S.n_lin; c->stats.n_error = S.n_err;
  c->tgt.nvox = c->stats.n_voxels = S.nvox;
  c->stats.deferred_target = S.def_t; c->stats.deferred_source = S.def_s;
  c->tgt.deferred_seen = S.def_t; c->src.deferred_seen = S.def_s;
  if (c->tgt.lazy == 2) { c->tgt.lazy_nq_seen = S.lazy_nq; c->tgt.lazy_ncell_seen = S.lazy_ncell; }
  // (S.src_sq == 0: this scan's figure was consumed by an earlier solve on the same clouds -- the first step hands the counter back
  // zeroed -- and the steering it caused stands)
  if (S.src_sq > 0.f) c->stats.source_crowding = c->src.n > 0 ? (double)S.src_sq / (double)c->src.n : 0.0;
  // steer the next scan's cell size: halve above 500 points per own cell, double below 40 -- up to twice the voxel size, which is where a
  // 0.2 m leaf-filtered VLP-16 sweep (the odometer's source cloud) ends up: fewer of its far-field queries go to the cooperative
  // kernel (frame body 0.76 -> 0.73 ms); a raw VLP-16 sweep (117) stays at the voxel size (2837 scans/s against 2683 at twice that).
  // (With four lanes per query a crowded cell is
  // cheap and a fine grid's far field -- every sparse query deferred to the cooperative kernel -- is what costs: an HDL-64 sweep,
  // 1270 / 380 / 107 at 1 / 0.5 / 0.25 m, is fastest at 0.5 m (c3 529 -> 594 scans/s against the former threshold of 300, which took it
  // to 0.25 m); two fused 64-beam sweeps, 2420 / 720 / 201, still want 0.25 m (184 against 139 scans/s at 0.5 m).)
  if (S.src_sq > 0.f && c->src_res <= 0.0 && c->src.n > 0) {
    const double cur = c->src.grid.res, crowd = c->stats.source_crowding;
    double next = cur;
    if (crowd > 500.0 && cur > 0.26 * c->prm.voxel_res) next = cur * 0.5;
    else if (crowd < 40.0 && cur < 2.0 * c->prm.voxel_res) next = std::fmin(cur * 2.0, 2.0 * c->prm.voxel_res);  // a leaf-filtered sweep: 2 x voxel size
    c->src_res_auto = next;
  }
  c->deferred_known = true;
  const int iters = S.failed ? S.outer + 1 : S.outer;  // iterations started, like nr_iterations_ + 1
  c->lm_last_outer = S.outer;
  c->stats.outer_iterations = iters;
  float fin[16];
  for (int i = 0; i < 16; i++) fin[i] = (float)S.x0[i];  // :77
  if (final_T) memcpy(final_T, fin, sizeof(fin));
  if (final_H) memcpy(final_H, S.Hfin, sizeof(double) * 36);
  if (iterations) *iterations = iters;
  if (converged) *converged = S.conv != 0 ? 1 : 0;
  if (lm_failed) *lm_failed = S.failed != 0 ? 1 : 0;
  if (fitness) {
    if (want_fitness && S.has_fit) *fitness = S.fit_sum / (double)n;
    else if ((rc = do_fitness(c, fin, fitness))) return rc;
  }
  return RGC_OK;
}

// lsq_registration_impl.hpp:53-79 (computeTransformation) + :125-172 (step_lm); SURVEY A.5
int rgc_align(rgc_ctx* c, const float guess[16], float final_T[16], double final_H[36], double* fitness, int* iterations,
              int* converged, int* lm_failed) {
  if (!c || !guess) return RGC_ERR_INVALID;
  if (!c->lm_host && !general_route(c)) {
    const int rc0 = rgc_align_begin(c, guess, fitness != nullptr);
    return rc0 ? rc0 : rgc_align_end(c, final_T, final_H, fitness, iterations, converged, lm_failed);
  }
  // (the general covariance route solves here as well: the device-chained driver's step kernel takes the source's NORMAL)
  // ---- RGC_LM_IMPL=host: the loop on the host over the public fine-seam kernels (cross-check of the device-chained driver) ----
  HIPCHK(c, hipSetDevice(c->device));
  int rc = need_inputs(c, /*validate=*/true);
  if (rc) return rc;
  const rgc_params& P = c->prm;
  double x0[16];
  for (int i = 0; i < 12; i++) x0[i] = (double)guess[i];
  x0[12] = x0[13] = x0[14] = 0.0;
  x0[15] = 1.0;
  double lambda = -1.0;  // :56
  bool conv = false, failed = false;
  int iters = 0;
  double Hfin[36];
  memset(Hfin, 0, sizeof(Hfin));
  for (int i = 0; i < 6; i++) Hfin[i * 7] = 1.0;  // final_hessian_.setIdentity(), :21
  c->stats.n_linearize = c->stats.n_error = c->stats.outer_iterations = 0;
  for (int it = 0; it < P.max_iterations && !conv; it++) {  // :65
    iters = it + 1;
    double H[36], b[6], y0, delta[16], d[6], xi[16], yi, lam_used;
    // :128 linearize + the first try of :135-144 in one enqueue
    if ((rc = do_linearize_try(c, x0, lambda, H, b, &y0, d, xi, &lam_used, &yi))) return rc;
    lambda = lam_used;  // :130-132 (first call: lambda0 = factor * max|H_ii|, computed on the device)
    double nu = 2.0;
    bool ok = false;
    for (int k = 0; k < P.lm_max_iterations; k++) {  // :135
      if (k > 0) {  // further tries of this outer iteration (rho < 0): solve on the host, evaluate on the device
        double dl[16];
        rgclm::lm_try(H, b, lambda, x0, d, dl, xi);  // :136-143
        if ((rc = do_error(c, xi, &yi))) return rc;  // :144
      }
      double R[9];
      rgclm::so3_exp_R(d, R);
      memset(delta, 0, sizeof(delta));
      for (int a = 0; a < 3; a++) { for (int e = 0; e < 3; e++) delta[a * 4 + e] = R[a * 3 + e]; delta[a * 4 + 3] = d[3 + a]; }
      delta[15] = 1.0;
      double den = 0;
      for (int i = 0; i < 6; i++) den += d[i] * (lambda * d[i] - b[i]);
      const double rho = (y0 - yi) / den;            // :145
      if (rho < 0) {                                 // :155-163
        if (is_converged(delta, P.rotation_eps, P.translation_eps)) { ok = true; break; }
        lambda = nu * lambda;
        nu = 2 * nu;
        continue;
      }
      memcpy(x0, xi, sizeof(xi));                    // :165
      lambda = lambda * std::fmax(1.0 / 3.0, 1 - std::pow(2 * rho - 1, 3));  // :166
      memcpy(Hfin, H, sizeof(Hfin));                 // :167
      ok = true;
      break;
    }
    if (!ok) { failed = true; break; }               // :69-72 "lm not converged!!"
    conv = is_converged(delta, P.rotation_eps, P.translation_eps);  // :74
  }
  c->stats.outer_iterations = iters;
  float fin[16];
  for (int i = 0; i < 16; i++) fin[i] = (float)x0[i];  // :77
  if (final_T) memcpy(final_T, fin, sizeof(fin));
  if (final_H) memcpy(final_H, Hfin, sizeof(Hfin));
  if (iterations) *iterations = iters;
  if (converged) *converged = conv ? 1 : 0;
  if (lm_failed) *lm_failed = failed ? 1 : 0;
  if (fitness && (rc = do_fitness(c, fin, fitness))) return rc;
  return RGC_OK;
}

int rgc_fitness(rgc_ctx* c, const float T[16], double* fitness) {
  if (!c || !T || !fitness) return RGC_ERR_INVALID;
  HIPCHK(c, hipSetDevice(c->device));
  return do_fitness(c, T, fitness);
}

int rgc_get_aligned_device(rgc_ctx* c, const float T[16], float* d_out, int stride_bytes) {
  if (!c || !T || !d_out) return RGC_ERR_INVALID;
  if (!c->src.ready) return fail(c, RGC_ERR_NO_INPUT, "source not set");
  if (stride_bytes < 12 || (stride_bytes & 3) || stride_bytes > 4096) return fail(c, RGC_ERR_INVALID, "bad stride");
  { int rk = check_device_range(c, d_out, (size_t)c->src.n * stride_bytes - (stride_bytes - 12), "rgc_get_aligned_device: d_out"); if (rk) return rk; }
  HIPCHK(c, hipSetDevice(c->device));
  int rc = join_source(c);
  if (rc) return rc;
  rgck::transform_f32(c->stream, c->src.in, c->src.stride_f, c->src.n, posef_from(T), d_out, stride_bytes / 4);
  // the kernel reads the source's input buffer on the MAIN stream and nothing here waits for it: the next host source is copied into
  // that buffer on the scan's stream, which must queue behind this read (set_cloud)
  HIPCHK(c, hipEventRecord(c->src_read_done, c->stream));
  c->src_read_pending = true;
  HIPCHK(c, hipGetLastError());
  return RGC_OK;
}

int rgc_get_aligned(rgc_ctx* c, const float T[16], float* out, int stride_bytes) {
  if (!c || !T || !out) return RGC_ERR_INVALID;
  if (!c->src.ready) return fail(c, RGC_ERR_NO_INPUT, "source not set");
  if (stride_bytes < 12 || (stride_bytes & 3) || stride_bytes > 4096) return fail(c, RGC_ERR_INVALID, "bad stride");
  HIPCHK(c, hipSetDevice(c->device));
  const int n = c->src.n;
  int rc = join_source(c);
  if (rc) return rc;
  if ((rc = ensure(c, c->scratch, sizeof(float) * 3 * (size_t)n))) return rc;
  rgck::transform_f32(c->stream, c->src.in, c->src.stride_f, n, posef_from(T), (float*)c->scratch.p, 3);
  if (stride_bytes == 12) {
    HIPCHK(c, hipMemcpyAsync(out, c->scratch.p, sizeof(float) * 3 * (size_t)n, hipMemcpyDeviceToHost, c->stream));
  } else {
    HIPCHK(c, hipMemcpy2DAsync(out, stride_bytes, c->scratch.p, 12, 12, n, hipMemcpyDeviceToHost, c->stream));
  }
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return RGC_OK;
}

static int get_covs(rgc_ctx* c, Cloud& cl, double* cov9, double* normals) {
  if (!cl.ready) return fail(c, RGC_ERR_NO_INPUT, "cloud not set");
  HIPCHK(c, hipSetDevice(c->device));
  int rc = validate_clouds(c);
  if (rc) return rc;
  if (!cl.ready) return fail(c, RGC_ERR_NO_INPUT, "cloud not set");
  const int n = cl.n;
  if ((rc = join_source(c))) return rc;
  if (cl.general) {  // the general route keeps a 3x3 per point, no normal
    if (normals) return fail(c, RGC_ERR_UNSUPPORTED, "normals exist only under RegularizationMethod PLANE with an additive voxel mode");
    if (!cov9) return RGC_OK;
    if ((rc = ensure(c, c->scratch, sizeof(double) * 9 * (size_t)n))) return rc;
    rgck::unsort6(c->stream, (const double*)cl.c6.p, (const float4*)cl.P.p, n, (double*)c->scratch.p);
    HIPCHK(c, hipMemcpyAsync(cov9, c->scratch.p, sizeof(double) * 9 * (size_t)n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return RGC_OK;
  }
  if ((rc = ensure(c, c->scratch, sizeof(double) * 3 * (size_t)n))) return rc;
  rgck::unsort3(c->stream, (const double*)cl.nx.p, (const double*)cl.ny.p, (const double*)cl.nz.p, (const float4*)cl.P.p, n, (double*)c->scratch.p);
  std::vector<double> tmp;
  double* dst = normals;
  if (!dst) { tmp.resize((size_t)n * 3); dst = tmp.data(); }
  HIPCHK(c, hipMemcpyAsync(dst, c->scratch.p, sizeof(double) * 3 * (size_t)n, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (cov9) {
    for (int i = 0; i < n; i++) {  // C = I - 0.999 n n^T  (fast_gicp_impl.hpp:281,293)
      const double* v = dst + (size_t)i * 3;
      double* C = cov9 + (size_t)i * 9;
      for (int a = 0; a < 3; a++)
        for (int b = 0; b < 3; b++) C[a * 3 + b] = (a == b ? 1.0 : 0.0) - 0.999 * v[a] * v[b];
    }
  }
  return RGC_OK;
}

int rgc_get_source_covariances(rgc_ctx* c, double* cov9, double* normals) { return c ? get_covs(c, c->src, cov9, normals) : RGC_ERR_INVALID; }
int rgc_get_target_covariances(rgc_ctx* c, double* cov9, double* normals) { return c ? get_covs(c, c->tgt, cov9, normals) : RGC_ERR_INVALID; }

// FastGICP::setSourceCovariances / setTargetCovariances (fast_gicp_impl.hpp:93-100): covariances given by the caller replace the ones
// computed from the 20 nearest neighbours.  This path keeps a covariance as its unit normal (C = I - 0.999 n n^T, the PLANE
// regularisation, fast_gicp_impl.hpp:280-293 -- the only form the reference's odometer produces): matrices of that form are
// accepted (to 1e-9), anything else is RGC_ERR_INVALID.  The target's voxel map is rebuilt from the new covariances.
static int set_covs(rgc_ctx* c, Cloud& cl, bool is_target, const double* cov9, int n) {
  if (!cov9) return RGC_ERR_INVALID;
  if (!cl.ready) return fail(c, RGC_ERR_NO_INPUT, "cloud not set");
  HIPCHK(c, hipSetDevice(c->device));
  int rc = validate_clouds(c);
  if (rc) return rc;
  if (!cl.ready) return fail(c, RGC_ERR_NO_INPUT, "cloud not set");
  if (is_target && c->tgt_owner) return fail(c, RGC_ERR_INVALID, "the target is borrowed (rgc_share_target): its covariances belong to the owner");
  if (n != cl.n) return fail(c, RGC_ERR_INVALID, "%d covariances for a cloud of %d points", n, cl.n);
  if (cl.general) {  // the general route takes any symmetric 3x3 as it comes (fast_gicp_impl.hpp:93-100 does not look at them either)
    if ((rc = join_source(c))) return rc;
    if ((rc = ensure(c, c->scratch, sizeof(double) * 9 * (size_t)n))) return rc;
    HIPCHK(c, hipMemcpyAsync(c->scratch.p, cov9, sizeof(double) * 9 * (size_t)n, hipMemcpyHostToDevice, c->stream));
    rgck::sort6(c->stream, (const double*)c->scratch.p, (const float4*)cl.P.p, n, (double*)cl.c6.p);
    if (is_target) {
      rgck::voxel_build_general(c->stream, (const float4*)cl.P.p, (const double*)cl.c6.p, (const int*)cl.start.p, cl.grid, n, (const int*)cl.cell_voxel.p,
                                (double*)cl.vox.p, (int*)cl.vox_cell.p, c->voxel_mode == RGC_VOXEL_MULTIPLICATIVE ? 1 : 0, nullptr);
      c->tgt_generation++;
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));   // (the caller's array is read until here)
    HIPCHK(c, hipGetLastError());
    c->corr_valid = false;
    cl.covs_user = true;
    return RGC_OK;
  }
  std::vector<double> nrm((size_t)n * 3);
  for (int i = 0; i < n; i++) {
    const double* C9 = cov9 + (size_t)i * 9;
    double M[9];
    for (int a = 0; a < 9; a++) M[a] = ((a % 4 == 0 ? 1.0 : 0.0) - C9[a]) / 0.999;   // n n^T
    const int d = (M[0] >= M[4] && M[0] >= M[8]) ? 0 : (M[4] >= M[8] ? 1 : 2);
    const double len = std::sqrt(M[4 * d]);
    double v[3] = {0, 0, 0};
    bool ok = len > 0.0 && std::isfinite(len);
    if (ok) {
      for (int a = 0; a < 3; a++) v[a] = M[3 * a + d] / len;
      for (int a = 0; a < 3 && ok; a++)
        for (int b = 0; b < 3; b++)
          if (!(std::fabs(M[3 * a + b] - v[a] * v[b]) <= 1.0e-9)) { ok = false; break; }
    }
    if (!ok) return fail(c, RGC_ERR_INVALID, "covariance %d is not of the plane-regularised form I - 0.999 n n^T: not supported", i);
    nrm[(size_t)i * 3] = v[0]; nrm[(size_t)i * 3 + 1] = v[1]; nrm[(size_t)i * 3 + 2] = v[2];
  }
  if ((rc = join_source(c))) return rc;
  if ((rc = ensure(c, c->scratch, sizeof(double) * 3 * (size_t)n))) return rc;
  HIPCHK(c, hipMemcpyAsync(c->scratch.p, nrm.data(), sizeof(double) * 3 * (size_t)n, hipMemcpyHostToDevice, c->stream));
  rgck::sort3(c->stream, (const double*)c->scratch.p, (const float4*)cl.P.p, n, (double*)cl.nx.p, (double*)cl.ny.p, (double*)cl.nz.p);
  if (is_target) {
    rgck::voxel_build(c->stream, (const float4*)cl.P.p, (const double*)cl.nx.p, (const double*)cl.ny.p, (const double*)cl.nz.p, (const int*)cl.start.p,
                      cl.grid, n, (const int*)cl.cell_voxel.p, (double*)cl.vox.p, (int*)cl.vox_cell.p);
    c->tgt_generation++;   // borrowers of this target must share again
  }
  HIPCHK(c, hipStreamSynchronize(c->stream));   // (the host vector goes out of scope)
  HIPCHK(c, hipGetLastError());
  c->corr_valid = false;
  cl.covs_user = true;
  return RGC_OK;
}
int rgc_set_source_covariances(rgc_ctx* c, const double* cov9, int n) { return c ? set_covs(c, c->src, false, cov9, n) : RGC_ERR_INVALID; }
int rgc_set_target_covariances(rgc_ctx* c, const double* cov9, int n) { return c ? set_covs(c, c->tgt, true, cov9, n) : RGC_ERR_INVALID; }

// pcl::Registration / FastGICP::clearSource, clearTarget (fast_gicp_impl.hpp:60-69): the cloud and its covariances are dropped
int rgc_clear_source(rgc_ctx* c) {
  if (!c) return RGC_ERR_INVALID;
  if (solve_in_flight(c)) return fail(c, RGC_ERR_INVALID, "a solve is in flight on this context: call rgc_align_end first");
  c->src.ready = false; c->src.n = 0; c->src.spec_used = false; c->corr_valid = false; c->deferred_known = false;
  return RGC_OK;
}
int rgc_clear_target(rgc_ctx* c) {
  if (!c) return RGC_ERR_INVALID;
  if (solve_in_flight(c)) return fail(c, RGC_ERR_INVALID, "a solve is in flight on this context: call rgc_align_end first");
  c->tgt.ready = false; c->tgt.n = 0; c->tgt.spec_used = false; c->corr_valid = false; c->deferred_known = false; c->map_bound = false;
  return RGC_OK;
}

// FastVGICP::swapSourceAndTarget (fast_vgicp_impl.hpp:46-53): the clouds change roles; the voxel map is rebuilt.  Each cloud is
// prepared again in its new role (the source's search grid is not the voxel grid): the covariances are the same function of the cloud.
int rgc_swap_source_and_target(rgc_ctx* c) {
  if (!c) return RGC_ERR_INVALID;
  if (!c->src.ready || !c->tgt.ready) return fail(c, RGC_ERR_NO_INPUT, "source and target must be set first");
  if (c->tgt_owner) return fail(c, RGC_ERR_INVALID, "the target is borrowed (rgc_share_target): it cannot become the source");
  HIPCHK(c, hipSetDevice(c->device));
  int rc = validate_clouds(c);
  if (rc) return rc;
  HIPCHK(c, hipStreamSynchronize(c->stream2));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  // Covariances the CALLER set travel with their cloud (the reference swaps source_covs_ and target_covs_, fast_vgicp_impl.hpp:46-53):
  // taken out in the caller's point order here, put back into the new ordering behind the preparation (computed ones are simply
  // computed again: the same function of the cloud).
  struct Kept { bool on = false; void* p = nullptr; } kept[2];  // [0]: the old source's (-> new target), [1]: the old target's (-> new source)
  auto drop_kept = [&]() { for (auto& k : kept) if (k.p) { (void)hipFree(k.p); k.p = nullptr; } };
  {
    Cloud* from[2] = {&c->src, &c->tgt};
    for (int a = 0; a < 2; a++) {
      if (!from[a]->covs_user) continue;
      kept[a].on = true;
      if (hipMalloc(&kept[a].p, sizeof(double) * (from[a]->general ? 9 : 3) * (size_t)from[a]->n) != hipSuccess) { drop_kept(); return fail(c, RGC_ERR_HIP, "hipMalloc failed (swap)"); }
      if (from[a]->general) rgck::unsort6(c->stream, (const double*)from[a]->c6.p, (const float4*)from[a]->P.p, from[a]->n, (double*)kept[a].p);
      else
      rgck::unsort3(c->stream, (const double*)from[a]->nx.p, (const double*)from[a]->ny.p, (const double*)from[a]->nz.p, (const float4*)from[a]->P.p,
                    from[a]->n, (double*)kept[a].p);
    }
    if (hipStreamSynchronize(c->stream) != hipSuccess) { drop_kept(); return fail(c, RGC_ERR_HIP, "hipStreamSynchronize failed (swap)"); }
  }
  // A target bound to the resident map (rgc_map_commit) was set from the map's own filter output buffer, which the NEXT commit overwrites:
  // as the scan it would keep pointing there (tests/fuzz/fuzz_api.py: swapped back later, it was prepared from another cloud's
  // points).  It takes a copy of its own with it.
  if (c->map_target.p && c->tgt.in == (const float*)c->map_target.p) {
    const size_t bytes = (size_t)c->tgt.n * 16;
    if ((rc = ensure(c, c->tgt.in_copy, bytes))) { drop_kept(); return rc; }
    if (hipMemcpyAsync(c->tgt.in_copy.p, c->map_target.p, bytes, hipMemcpyDeviceToDevice, c->stream) != hipSuccess) { (void)hipGetLastError(); drop_kept(); return fail(c, RGC_ERR_HIP, "swap: copy of the map's target failed"); }
    c->tgt.in = (const float*)c->tgt.in_copy.p;
    c->tgt.stride_f = 4;
  }
  std::swap(c->src.in_copy, c->tgt.in_copy);
  std::swap(c->src.in, c->tgt.in);
  std::swap(c->src.stride_f, c->tgt.stride_f);
  std::swap(c->src.n, c->tgt.n);
  c->src.ready = c->tgt.ready = false;
  c->corr_valid = false; c->deferred_known = false; c->map_bound = false;
  if ((rc = prepare_cloud(c, c->tgt, true, /*force_bbox=*/true))) { drop_kept(); return rc; }
  if ((rc = prepare_cloud(c, c->src, false, /*force_bbox=*/true))) { drop_kept(); return rc; }
  if (kept[0].on || kept[1].on) {
    hipError_t e = hipStreamSynchronize(c->stream2);
    if (e == hipSuccess && kept[0].on) {  // the old source's covariances on the new target: normals, then its voxel map from them
      Cloud& cl = c->tgt;
      if (cl.general) {
        rgck::sort6(c->stream, (const double*)kept[0].p, (const float4*)cl.P.p, cl.n, (double*)cl.c6.p);
        rgck::voxel_build_general(c->stream, (const float4*)cl.P.p, (const double*)cl.c6.p, (const int*)cl.start.p, cl.grid, cl.n, (const int*)cl.cell_voxel.p,
                                  (double*)cl.vox.p, (int*)cl.vox_cell.p, c->voxel_mode == RGC_VOXEL_MULTIPLICATIVE ? 1 : 0, nullptr);
      } else {
      rgck::sort3(c->stream, (const double*)kept[0].p, (const float4*)cl.P.p, cl.n, (double*)cl.nx.p, (double*)cl.ny.p, (double*)cl.nz.p);
      rgck::voxel_build(c->stream, (const float4*)cl.P.p, (const double*)cl.nx.p, (const double*)cl.ny.p, (const double*)cl.nz.p, (const int*)cl.start.p,
                        cl.grid, cl.n, (const int*)cl.cell_voxel.p, (double*)cl.vox.p, (int*)cl.vox_cell.p);
      }
      cl.covs_user = true;
    }
    if (e == hipSuccess && kept[1].on) {
      Cloud& cl = c->src;
      if (cl.general) rgck::sort6(c->stream, (const double*)kept[1].p, (const float4*)cl.P.p, cl.n, (double*)cl.c6.p);
      else rgck::sort3(c->stream, (const double*)kept[1].p, (const float4*)cl.P.p, cl.n, (double*)cl.nx.p, (double*)cl.ny.p, (double*)cl.nz.p);
      cl.covs_user = true;
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    drop_kept();
    if (e != hipSuccess) return fail(c, RGC_ERR_HIP, "swap: %s", hipGetErrorString(e));
    c->src_pending = false;  // (both streams have drained)
  }
  c->stats.n_target = c->tgt.n; c->stats.target_cells = c->tgt.grid.ncell;
  c->stats.n_source = c->src.n; c->stats.source_cells = c->src.grid.ncell;
  return RGC_OK;
}

static int fetch_nvox(rgc_ctx* c) {
  if (c->tgt.nvox >= 0) return RGC_OK;
  HIPCHK(c, hipMemcpyAsync(c->h_small + 7, c->d_small + 7, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->tgt.nvox = c->h_small[7];
  c->stats.n_voxels = c->tgt.nvox;
  return RGC_OK;
}

int rgc_get_voxels(rgc_ctx* c, int cap, int* coords, int* num, double* mean, double* cov9, int* count) {
  if (!c || !count) return RGC_ERR_INVALID;
  if (!c->tgt.ready) return fail(c, RGC_ERR_NO_INPUT, "target not set");
  HIPCHK(c, hipSetDevice(c->device));
  int rc = validate_clouds(c);
  if (rc) return rc;
  if (!c->tgt.ready) return fail(c, RGC_ERR_NO_INPUT, "target not set");
  if ((rc = fetch_nvox(c))) return rc;
  const int V = c->tgt.nvox;
  *count = V;
  const int m = V < cap ? V : cap;
  if (m <= 0) return RGC_OK;
  std::vector<double> rec((size_t)m * rgck::kVoxRec);
  std::vector<int> cell((size_t)m);
  HIPCHK(c, hipMemcpyAsync(rec.data(), c->tgt.vox.p, sizeof(double) * rec.size(), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(cell.data(), c->tgt.vox_cell.p, sizeof(int) * cell.size(), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  const rgck::Grid& g = c->tgt.grid;
  for (int v = 0; v < m; v++) {
    const double* r = &rec[(size_t)v * rgck::kVoxRec];
    const int ci = cell[v];
    if (coords) {
      coords[v * 3 + 0] = ci % g.dim[0] + g.minc[0];
#if defined(RGC_Y_SLOWEST) && RGC_Y_SLOWEST  // the cell order of rgck::cell_index
      coords[v * 3 + 2] = (ci / g.dim[0]) % g.dim[2] + g.minc[2];
      coords[v * 3 + 1] = ci / (g.dim[0] * g.dim[2]) + g.minc[1];
#else
      coords[v * 3 + 1] = (ci / g.dim[0]) % g.dim[1] + g.minc[1];
      coords[v * 3 + 2] = ci / (g.dim[0] * g.dim[1]) + g.minc[2];
#endif
    }
    if (num) num[v] = (int)r[9];
    if (mean) { mean[v * 3] = r[0]; mean[v * 3 + 1] = r[1]; mean[v * 3 + 2] = r[2]; }
    if (cov9) {
      double* C = cov9 + (size_t)v * 9;
      C[0] = r[3]; C[1] = r[4]; C[2] = r[5];
      C[3] = r[4]; C[4] = r[6]; C[5] = r[7];
      C[6] = r[5]; C[7] = r[7]; C[8] = r[8];
    }
  }
  return RGC_OK;
}

// ---- B2 / B3 / B9: the stages either side of the operator in the odometer's frame body ----
static int stage_in(rgc_ctx* c, const float* p, int n, int stride_bytes, int on_device, const float** d_in) {
  if (n > (1 << 27)) return fail(c, RGC_ERR_INVALID, "cloud has %d points, the limit is 2^27", n);  // (every entry point that takes a cloud: one limit)
  if (on_device) {
    if (n > 0) { const int rk = check_device_range(c, p, (size_t)n * stride_bytes - (stride_bytes - 12), "input cloud"); if (rk) return rk; }
    *d_in = p;
    return RGC_OK;
  }
  const size_t bytes = (size_t)n * stride_bytes;
  int rc = ensure(c, c->pre_in, bytes);
  if (rc) return rc;
  HIPCHK(c, hipMemcpyAsync(c->pre_in.p, p, bytes, hipMemcpyHostToDevice, c->stream));
  *d_in = (const float*)c->pre_in.p;
  return RGC_OK;
}

int rgc_deskew(rgc_ctx* c, float* xyzi, int n, int stride_bytes, const double q[4], const double t[3], int on_device) {
  if (!c || !xyzi || !q || !t || n < 0) return RGC_ERR_INVALID;
  if (stride_bytes < 16 || (stride_bytes & 3) || stride_bytes > 4096) return fail(c, RGC_ERR_INVALID, "de-skew needs x,y,z,intensity: stride_bytes >= 16");
  if (n == 0) return RGC_OK;
  HIPCHK(c, hipSetDevice(c->device));
  const float* d_in;
  int rc = stage_in(c, xyzi, n, stride_bytes, on_device, &d_in);
  if (rc) return rc;
  // q_last_curr.inverse() = conjugate / squaredNorm (Eigen), RGC_odometer.cpp:1444
  const double n2 = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
  if (!(n2 > 0)) return fail(c, RGC_ERR_INVALID, "zero quaternion");
  rgck::Quat qi{-q[0] / n2, -q[1] / n2, -q[2] / n2, q[3] / n2};
  if (c->main_has_target_prep && map_prep_finished(c)) c->main_has_target_prep = false;  // it has drained
  rgck::deskew(c->stream, (float*)d_in, stride_bytes / 4, n, qi, t);
  if (!on_device) HIPCHK(c, hipMemcpyAsync(xyzi, d_in, (size_t)n * stride_bytes, hipMemcpyDeviceToHost, c->stream));
  // device memory: in place and stream-ordered, whatever reads the sweep next on the main stream is enqueued behind it -- unless a map
  // preparation is still pending there: rgc_set_source_device orders the scan's stream after a mark recorded BEFORE that preparation
  // (see prepare_cloud), which this kernel would then lie behind
  if (!on_device || c->main_has_target_prep) HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipGetLastError());
  return RGC_OK;
}

int rgc_transform_cloud(rgc_ctx* c, const float* xyzi, int n, int stride_bytes, const double q[4], const double t[3], float* out_xyzi,
                        int on_device) {
  if (!c || !xyzi || !q || !t || !out_xyzi || n < 0) return RGC_ERR_INVALID;
  if (stride_bytes < 12 || (stride_bytes & 3) || stride_bytes > 4096) return fail(c, RGC_ERR_INVALID, "bad stride");
  if (n == 0) return RGC_OK;
  HIPCHK(c, hipSetDevice(c->device));
  const float* d_in;
  int rc = stage_in(c, xyzi, n, stride_bytes, on_device, &d_in);
  if (rc) return rc;
  float* d_out = out_xyzi;
  if (!on_device) {
    if ((rc = ensure(c, c->pre_out, sizeof(float) * 4 * (size_t)n))) return rc;
    d_out = (float*)c->pre_out.p;
  }
  rgck::transform_q(c->stream, d_in, stride_bytes / 4, n, rgck::Quat{q[0], q[1], q[2], q[3]}, t, d_out, 4);
  if (!on_device) HIPCHK(c, hipMemcpyAsync(out_xyzi, d_out, sizeof(float) * 4 * (size_t)n, hipMemcpyDeviceToHost, c->stream));
  // device memory: stream-ordered like rgc_deskew; the one exception is the same as there (a pending map preparation, see rgc_deskew)
  if (!on_device || c->main_has_target_prep) HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipGetLastError());
  return RGC_OK;
}

// B9 followed by setInputTarget, in one call and without a host round trip: the sub-map (device memory, fixed between calls) re-expressed
// by (q, t) into d_scratch and handed to the registration as its new target (RGC_odometer.cpp:1248-1256, 998, 1007).  The output's
// bounding box follows from the input's -- measured once per input buffer (whole 1 m cells, k_bbox) -- and the transform: its eight
// corners through q * p + t in fp64, a millimetre added for the fp32 rounding of the stored points.  rgc_set_target_device takes its grid
// from that box: no bounding-box kernel, no read-back, and no speculative-grid miss when the re-framed map's box swings with the yaw.
// the argument checks of rgc_set_target_reframed (also made by rgc_align_end_reframe BEFORE it consumes the solve)
static int reframe_args_ok(rgc_ctx* c, const float* d_xyzi, int n, int stride_bytes, const float* d_scratch) {
  if (!d_xyzi || !d_scratch || n <= 0) return fail(c, RGC_ERR_INVALID, "rgc_set_target_reframed: null buffer or no points");
  if (stride_bytes < 12 || (stride_bytes & 3) || stride_bytes > 4096) return fail(c, RGC_ERR_INVALID, "bad stride");
  if (n > (1 << 27)) return fail(c, RGC_ERR_INVALID, "cloud has %d points, the limit is 2^27 (32-bit byte offsets into the sorted array)", n);
  if (n < c->prm.k_correspondences) return fail(c, RGC_ERR_TOO_FEW_POINTS, "target cloud has %d points, need >= k = %d", n, c->prm.k_correspondences);
  // the re-framed cloud is WRITTEN to d_scratch while d_xyzi is read: they must not overlap (and one buffer has one bounding-box hint)
  const char* a0 = (const char*)d_xyzi; const char* a1 = a0 + (size_t)n * stride_bytes;
  const char* b0 = (const char*)d_scratch; const char* b1 = b0 + (size_t)n * 16;
  if (a0 < b1 && b0 < a1) return fail(c, RGC_ERR_INVALID, "rgc_set_target_reframed: d_scratch overlaps d_xyzi");
  { int rk = check_device_range(c, d_xyzi, (size_t)n * stride_bytes - (stride_bytes - 12), "rgc_set_target_reframed: d_xyzi"); if (rk) return rk; }
  { int rk = check_device_range(c, d_scratch, (size_t)n * 16, "rgc_set_target_reframed: d_scratch"); if (rk) return rk; }
  return RGC_OK;
}

int rgc_set_target_reframed(rgc_ctx* c, const float* d_xyzi, int n, int stride_bytes, const double q[4], const double t[3], float* d_scratch) {
  if (!c || !q || !t) return RGC_ERR_INVALID;
  if (solve_in_flight(c)) return fail(c, RGC_ERR_INVALID, "a solve is in flight on this context: call rgc_align_end first");
  {
    const int rc = reframe_args_ok(c, d_xyzi, n, stride_bytes, d_scratch);
    if (rc) return rc;
  }
  // a pose that is not a pose -- the NaN a diverged solve hands on through rgc_align_end_reframe, a zero quaternion -- has no box to derive a
  // grid from (the float -> int conversions behind it are undefined: tests/fuzz/fuzz_api.py saw a 40-petabyte allocation request)
  {
    const double qq = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
    if (!(std::isfinite(qq) && qq > 1.0e-12 && qq < 1.0e12 && std::isfinite(t[0]) && std::isfinite(t[1]) && std::isfinite(t[2]) &&
          std::fabs(t[0]) <= 1.0e8 && std::fabs(t[1]) <= 1.0e8 && std::fabs(t[2]) <= 1.0e8))
      return fail(c, RGC_ERR_NONFINITE, "rgc_set_target_reframed: the pose (q, t) is not finite");
  }
  HIPCHK(c, hipSetDevice(c->device));
  const float* xyzi = d_xyzi;
  const float* d_in = d_xyzi;
  float* out_xyzi = d_scratch;
  const int on_device = 1;
  if (on_device && c->spec_on) {
    // the output's bounding box from the input's: measured once per input buffer (whole cells of 1 m, k_bbox), then the eight
    // corners through q * p + t in fp64, a millimetre added for the fp32 rounding of the stored points
    const rgc_ctx::BoxHint* hin = find_hint(c, xyzi, n);
    if (!hin) {
      int* dsm = c->d_small + 32;
      int* hsm = c->h_small + 32;
      const int init[8] = {INT_MAX, INT_MAX, INT_MAX, INT_MIN, INT_MIN, INT_MIN, 0, 0};
      memcpy(hsm, init, sizeof(init));
      HIPCHK(c, hipMemcpyAsync(dsm, hsm, sizeof(init), hipMemcpyHostToDevice, c->stream));
      rgck::bbox(c->stream, d_in, stride_bytes / 4, n, 1.0, dsm, dsm + 6);
      HIPCHK(c, hipMemcpyAsync(hsm, dsm, 7 * sizeof(int), hipMemcpyDeviceToHost, c->stream));
      HIPCHK(c, hipStreamSynchronize(c->stream));
      if (!hsm[6]) {
        const double lo[3] = {hsm[0] + 0.5, hsm[1] + 0.5, hsm[2] + 0.5}, hi[3] = {hsm[3] + 1.5, hsm[4] + 1.5, hsm[5] + 1.5};
        put_hint(c, xyzi, n, lo, hi);
        hin = find_hint(c, xyzi, n);
      }
    }
    if (hin) {
      const double x = q[0], y = q[1], z = q[2], w = q[3];
      const double R[9] = {1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w), 2 * (x * y + z * w), 1 - 2 * (x * x + z * z),
                           2 * (y * z - x * w), 2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)};
      double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
      for (int k = 0; k < 8; k++) {
        const double p[3] = {(k & 1) ? hin->hi[0] : hin->lo[0], (k & 2) ? hin->hi[1] : hin->lo[1], (k & 4) ? hin->hi[2] : hin->lo[2]};
        for (int a = 0; a < 3; a++) {
          const double v = R[3 * a] * p[0] + R[3 * a + 1] * p[1] + R[3 * a + 2] * p[2] + t[a];
          lo[a] = std::min(lo[a], v - 1.0e-3);
          hi[a] = std::max(hi[a], v + 1.0e-3);
        }
      }
      // how large the re-framed box can get as the vehicle turns: under any yaw its x / y extents stay within the horizontal diagonal of
      // the map's own box; pitch and roll of a ground vehicle tilt it by a few cells.  The cell arrays are sized for that ONCE.
      const double dxy = std::hypot(hin->hi[0] - hin->lo[0], hin->hi[1] - hin->lo[1]);
      put_hint(c, out_xyzi, n, lo, hi, dxy, (hi[2] - lo[2]) + 0.08 * dxy);
    }
  }
  // (the re-framing itself is left to the preparation: its counting pass writes d_scratch on the way, prepare_cloud)
  const rgck::Reframe rf{d_in, stride_bytes / 4, rgck::Quat{q[0], q[1], q[2], q[3]}, {t[0], t[1], t[2]}};
  HIPCHK(c, hipGetLastError());
  return set_cloud(c, c->tgt, true, d_scratch, n, 16, true, &rf);
}

int rgc_align_end_reframe(rgc_ctx* c, rgc_ctx* next, double Tw[16], const float* d_map, int n, int stride_bytes, float* d_scratch,
                          float final_T[16], double final_H[36], double* fitness, int* iterations, int* converged, int* lm_failed) {
  if (!c || !next || !Tw) return RGC_ERR_INVALID;
  // Everything that could make the second half (the next frame's target) fail for the caller's arguments is checked BEFORE the solve is
  // consumed: a non-OK return then means "nothing happened" (the solve is still pending, Tw untouched) -- or, past this point, a HIP /
  // allocation failure inside the preparation, with the solve's outputs and Tw already valid (the message says which call failed).
  if (next != c) {
    if (!ctx_alive(next)) return fail(c, RGC_ERR_INVALID, "rgc_align_end_reframe: the next context is not alive");
    if (solve_in_flight(next)) return fail(c, RGC_ERR_INVALID, "rgc_align_end_reframe: a solve is in flight on the next context");
  }
  if (!c->pend.active && !c->gen_res.on) return fail(c, RGC_ERR_INVALID, "rgc_align_end without rgc_align_begin");
  {
    const int rc0 = reframe_args_ok(next, d_map, n, stride_bytes, d_scratch);
    if (rc0) { if (next != c) fail(c, rc0, "rgc_align_end_reframe: %s", next->err); return rc0; }
  }
  // world_T * T in fp64, rows in ascending k (the composition of :1201-1203 on matrices), and world -> body of the new pose: R^T and
  // -R^T t; the unit quaternion of R^T by Shepperd's branches (:1250-1255)
  auto compose = [](const double* Tw_in, const float* T, double* W, double* q, double* t) {
    for (int a = 0; a < 4; a++)
      for (int b = 0; b < 4; b++) {
        double v = 0.0;
        for (int k = 0; k < 4; k++) v += Tw_in[a * 4 + k] * (double)T[k * 4 + b];
        W[a * 4 + b] = v;
      }
    const double Rt[3][3] = {{W[0], W[4], W[8]}, {W[1], W[5], W[9]}, {W[2], W[6], W[10]}};
    const double tr = Rt[0][0] + Rt[1][1] + Rt[2][2];
    if (tr > 0) {
      const double s4 = 2.0 * std::sqrt(tr + 1.0);
      q[0] = (Rt[2][1] - Rt[1][2]) / s4; q[1] = (Rt[0][2] - Rt[2][0]) / s4; q[2] = (Rt[1][0] - Rt[0][1]) / s4; q[3] = 0.25 * s4;
    } else {
      const int i = (Rt[0][0] >= Rt[1][1] && Rt[0][0] >= Rt[2][2]) ? 0 : (Rt[1][1] >= Rt[2][2] ? 1 : 2);
      const int j = (i + 1) % 3, k = (i + 2) % 3;
      const double s4 = 2.0 * std::sqrt(1.0 + Rt[i][i] - Rt[j][j] - Rt[k][k]);
      q[3] = (Rt[k][j] - Rt[j][k]) / s4;
      q[i] = 0.25 * s4;
      q[j] = (Rt[j][i] + Rt[i][j]) / s4;
      q[k] = (Rt[k][i] + Rt[i][k]) / s4;
    }
    const double nrm = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    for (int a = 0; a < 4; a++) q[a] /= nrm;
    const double tx = W[3], ty = W[7], tz = W[11];
    t[0] = -(Rt[0][0] * tx + Rt[0][1] * ty + Rt[0][2] * tz);
    t[1] = -(Rt[1][0] * tx + Rt[1][1] * ty + Rt[1][2] * tz);
    t[2] = -(Rt[2][0] * tx + Rt[2][1] * ty + Rt[2][2] * tz);
  };
  // The POSE of a solve whose score is chained to it arrives before the score does (LmEarly: the deciding launch posts it, then scores it,
  // ~25 us at the headline size).  On two contexts taking turns the next frame's target needs nothing else: it is enqueued on `next` while
  // this context's last launch still computes the score, whose arrival the call then waits for like rgc_align_end.  Not on one context
  // (the score reads the buffers the next preparation writes), not when a guard tripped or a lazy target missed (the solve is repeated).
  float Te[16];
  double We[16], qe[4], te[3];
  bool early_done = false;
  int rc_next = RGC_OK;
  if (next != c && RGC_EARLY_POSE && c->post_on && c->d_post && c->d_early && c->pend.active && c->pend.want_fitness && !c->lm_host && !c->gen_res.on) {
    volatile int* eg = &c->h_early->gen;
    volatile int* fg = &c->h_post->gen;
    bool early = false;
    for (unsigned spin = 0;; spin++) {
      if (*eg == c->lm_seq) { early = true; break; }
      if (*fg == c->lm_seq) break;
      if (spin & 31u) continue;
      const hipError_t qy = hipEventQuery(c->lm_tail);
      if (qy == hipSuccess) { early = *eg == c->lm_seq; break; }
      if (qy != hipErrorNotReady) break;   // (rgc_align_end below reports it)
      (void)hipGetLastError();
    }
    if (early) {
      std::atomic_thread_fence(std::memory_order_acquire);
      rgck::LmEarly E;
      memcpy(&E, c->h_early, sizeof(E));
      if (E.pad == 0 && E.pad2 == 0) {
        for (int i = 0; i < 16; i++) Te[i] = (float)E.x0[i];  // final_transformation_ = x0.cast<float>(), :77
        compose(Tw, Te, We, qe, te);
        rc_next = rgc_set_target_reframed(next, d_map, n, stride_bytes, qe, te, d_scratch);
        if (rc_next) fail(c, rc_next, "rgc_align_end_reframe: %s", next->err);
        early_done = true;
      }
    }
  }
  float T[16];
  char next_err[sizeof(c->err)];
  if (rc_next) memcpy(next_err, c->err, sizeof(next_err));
  int rc = rgc_align_end(c, T, final_H, fitness, iterations, converged, lm_failed);
  if (rc) return rc;
  if (final_T) memcpy(final_T, T, sizeof(T));
  if (early_done && memcmp(T, Te, sizeof(T)) == 0) {  // (always, unless the solve had to be repeated behind the early pose's back)
    memcpy(Tw, We, sizeof(We));
    if (rc_next) memcpy(c->err, next_err, sizeof(next_err));
    return rc_next;
  }
  double W[16], q[4], t[3];
  compose(Tw, T, W, q, t);
  memcpy(Tw, W, sizeof(W));
  rc = rgc_set_target_reframed(next, d_map, n, stride_bytes, q, t, d_scratch);
  if (rc && next != c) fail(c, rc, "rgc_align_end_reframe: %s", next->err);
  return rc;
}

// The rows chain of the leaf filter on box g (rgc_pre.hip); one read-back: *flags (bits as rgck::vg_rows documents) and *n_out.
// h_result (nullable): the chain is only ENQUEUED -- its three result ints go to h_result (pinned), c->vg_done is recorded behind the
// copy, and the caller picks them up later (rgc_voxelgrid_begin / _end); flags / n_out are not written then.
static int voxelgrid_rows(rgc_ctx* c, const float* d_in, int stride_f, int n, float inv, const rgck::LeafGrid& g, int edge, bool dense, float* d_out,
                          int* flags, int* n_out, int* h_result = nullptr) {
  hipStream_t s = c->stream;
  int* dsm = c->d_small + 24;
  int* hsm = c->h_small + 24;
  Cloud& cl = c->aux;
  int rc;
  // the sort's buckets: leaves for a dense cloud, whole grid rows for a sweep, and for a LARGE cloud in a box too big for leaf buckets
  // segments of a row, as many as keep the table near 4 n entries (the ranking pass is quadratic in a bucket's population: a ground-level
  // row of a 1.3 M-point keyframe store holds thousands of points)
  int seg_shift = dense ? 0 : 31;
  if (!dense) {
    const double rows = (double)g.div[1] * (double)g.div[2], nseg_max = 4.0 * (double)n / rows;
    if (nseg_max >= 2.0) {
      seg_shift = 3;
      while (seg_shift < 30 && (double)(((long long)g.div[0] + (1ll << seg_shift) - 1) >> seg_shift) > nseg_max) seg_shift++;
    }
  }
  const size_t nr1 = (size_t)g.div[1] * (size_t)g.div[2] * (size_t)rgck::vg_segments(g, seg_shift) + 1;   // buckets
  if ((rc = ensure(c, cl.cell_of, sizeof(int) * n))) return rc;                                                    // row of every point
  if ((rc = ensure(c, cl.slot_of, sizeof(int) * n))) return rc;                                                    // leaf x of every point
  if ((rc = ensure(c, c->vg_pos, sizeof(int) * n))) return rc;                                                     // arrival slot, then output number
  if ((rc = ensure(c, c->vg_order, sizeof(int) * n))) return rc;
  if ((rc = ensure(c, c->vg_tmp, sizeof(long long) * n))) return rc;
  if ((rc = ensure(c, c->vg_leaf, sizeof(long long) * n))) return rc;
  if ((rc = ensure(c, cl.cnt, sizeof(int) * nr1))) return rc;
  if ((rc = ensure(c, cl.start, sizeof(int) * nr1))) return rc;
  const size_t row_bs = sizeof(long long) * (nr1 / 2048 + 2);
  if ((rc = ensure(c, cl.block_sums, row_bs + sizeof(int) * ((size_t)n / 2048 + 2)))) return rc;
  if (cl.cnt.p != cl.cnt_seen) { cl.cnt_clean = 0; cl.cnt_seen = cl.cnt.p; }
  if (cl.cnt_clean < nr1) {  // afterwards the scan leaves the counters it consumed at zero: no fill per call
    const size_t fill = std::min(cl.cnt.cap, (sizeof(int) * nr1 + 255) & ~(size_t)255);
    HIPCHK(c, hipMemsetAsync(cl.cnt.p, 0, fill, s));
  }
  cl.cnt_clean = std::max(cl.cnt_clean, nr1);  // (what lies beyond this call's rows was not touched: the scan's and the map's filter take turns)
  if (!c->vg_flags_clean) HIPCHK(c, hipMemsetAsync(dsm + 6, 0, sizeof(int), s));
  c->vg_flags_clean = false;
  rgck::vg_rows(s, d_in, stride_f, n, inv, g, edge, seg_shift, (int*)cl.cell_of.p, (int*)cl.slot_of.p, (int*)c->vg_pos.p, (int*)cl.cnt.p, (int*)cl.start.p,
                cl.block_sums.p, (unsigned long long*)c->vg_tmp.p, (int*)c->vg_order.p, (unsigned long long*)c->vg_leaf.p,
                (int*)((char*)cl.block_sums.p + row_bs), d_out, dsm + 5);
  if (h_result) {  // (the finished chain leaves the flag word zeroed: the next chain on this stream finds it so)
    HIPCHK(c, hipMemcpyAsync(h_result, dsm + 5, 3 * sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipEventRecord(c->vg_done, s));
    c->vg_flags_clean = true;
    return RGC_OK;
  }
  HIPCHK(c, hipMemcpyAsync(hsm + 5, dsm + 5, 3 * sizeof(int), hipMemcpyDeviceToHost, s));
  HIPCHK(c, hipStreamSynchronize(s));
  c->vg_flags_clean = true;
  *flags = hsm[5];
  *n_out = hsm[7];
  return RGC_OK;
}
static bool vg_rows_fit(const rgck::LeafGrid& g, int n) {  // sparse enough for the sort over rows (else: over the leaves)
  const double ncell = (double)g.div[0] * (double)g.div[1] * (double)g.div[2], nrows = (double)g.div[1] * (double)g.div[2];
  return ncell <= 2147483647.0 && ncell > 64.0 * (double)n && nrows <= 64.0e6;
}

int rgc_voxelgrid(rgc_ctx* c, const float* xyzi, int n, int stride_bytes, float leaf, float* out_xyzi, int* n_out, int on_device) {
  if (!c || !xyzi || !out_xyzi || !n_out || n < 0) return RGC_ERR_INVALID;
  if (stride_bytes < 12 || (stride_bytes & 3) || stride_bytes > 4096 || !(leaf > 0.f) || !std::isfinite(leaf)) return fail(c, RGC_ERR_INVALID, "bad stride or leaf size");
  *n_out = 0;
  if (n == 0) return RGC_OK;
  HIPCHK(c, hipSetDevice(c->device));
  hipStream_t s = c->stream;
  const float* d_in;
  int rc = stage_in(c, xyzi, n, stride_bytes, on_device, &d_in);
  if (rc) return rc;
  const int stride_f = stride_bytes / 4;
  const float inv = 1.0f / leaf;  // inverse_leaf_size_
  int* dsm = c->d_small + 24;
  int* hsm = c->h_small + 24;
  float* d_out = out_xyzi;
  if (!on_device) {
    if ((rc = ensure(c, c->pre_out, sizeof(float) * 4 * (size_t)n))) return rc;
    d_out = (float*)c->pre_out.p;
  }
  // leaves added on every side of a measured box when the next cloud of this leaf size is filtered on it: 32 for a sweep (its rows are
  // what is counted and scanned), 8 for a dense cloud (its leaves are: a wider box is a longer scan)
  constexpr int kPadSparse = 32, kPadDense = 8;
  auto padded = [](const rgck::LeafGrid& g, int pad) {
    rgck::LeafGrid p = g;
    for (int a = 0; a < 3; a++) { p.minb[a] -= pad; p.div[a] += 2 * pad; }
    return p;
  };
  rgc_ctx::VgBox* box = nullptr;
  for (auto& b : c->vg_box) if (b.leaf == leaf) box = &b;
  bool done = false;
  if (box && box->valid) {
    // the box of an earlier cloud: no bounding-box pass, no read-back before the filter (the frames of a sequence span the same volume)
    const rgck::LeafGrid ps = padded(box->g, kPadSparse), pd = padded(box->g, kPadDense);
    const bool sparse = vg_rows_fit(ps, n);
    const double dcell = (double)pd.div[0] * (double)pd.div[1] * (double)pd.div[2];
    if (sparse || dcell <= (double)c->prm.max_cells) {
      int flags = 0, no = 0;
      if ((rc = voxelgrid_rows(c, d_in, stride_f, n, inv, sparse ? ps : pd, (sparse ? kPadSparse : kPadDense) / 2, !sparse, d_out, &flags, &no))) return rc;
      if (flags & 1) return fail(c, RGC_ERR_NONFINITE, "cloud contains non-finite coordinates (PCL skips them; remove NaNs first)");
      if (flags & 6) box->valid = false;  // outside: measure and repeat now; near a face: measure at the next call
      if (!(flags & 2)) { *n_out = no; done = true; hint_from_leaf_grid(c, out_xyzi, no, sparse ? ps : pd, leaf); }
    }
  }
  if (!done) {
    int init[8] = {INT_MAX, INT_MAX, INT_MAX, INT_MIN, INT_MIN, INT_MIN, 0, 0};
    memcpy(hsm, init, sizeof(init));
    c->vg_flags_clean = false;
    HIPCHK(c, hipMemcpyAsync(dsm, hsm, sizeof(init), hipMemcpyHostToDevice, s));
    rgck::vg_bbox(s, d_in, stride_f, n, inv, dsm, dsm + 6);
    HIPCHK(c, hipMemcpyAsync(hsm, dsm, 7 * sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    if (hsm[6]) return fail(c, RGC_ERR_NONFINITE, "cloud contains non-finite coordinates (PCL skips them; remove NaNs first)");
    c->vg_flags_clean = true;
    rgck::LeafGrid g{};
    double ncell = 1.0;
    for (int a = 0; a < 3; a++) { g.minb[a] = hsm[a]; g.div[a] = hsm[3 + a] - hsm[a] + 1; ncell *= (double)g.div[a]; }
    {  // keep the measured box for the next cloud of this leaf size
      if (!box) { box = &c->vg_box[c->vg_box_next]; c->vg_box_next = (c->vg_box_next + 1) % 4; box->leaf = leaf; }
      bool ok = ncell <= 2.0e9;
      for (int a = 0; a < 3; a++) if (g.minb[a] < -1000000000 || g.div[a] > 1000000000) ok = false;
      box->g = g;
      box->valid = ok;
    }
    if (ncell > 2147483647.0) {
      // PCL: "Leaf size is too small for the input dataset. Integer indices would overflow." -> output = input
      rgck::transform_q(s, d_in, stride_f, n, rgck::Quat{0, 0, 0, 1}, (const double[3]){0, 0, 0}, d_out, 4);
      *n_out = n;
      HIPCHK(c, hipStreamSynchronize(s));
    } else if (vg_rows_fit(g, n)) {
      int flags = 0;
      if ((rc = voxelgrid_rows(c, d_in, stride_f, n, inv, g, 0, false, d_out, &flags, n_out))) return rc;
      hint_from_leaf_grid(c, out_xyzi, *n_out, g, leaf);
    } else {
      // a dense cloud: the same chain with the leaves themselves as the sort's buckets
      if (ncell > (double)c->prm.max_cells) return fail(c, RGC_ERR_GRID_TOO_LARGE, "leaf grid %d x %d x %d exceeds max_cells", g.div[0], g.div[1], g.div[2]);
      int flags = 0;
      if ((rc = voxelgrid_rows(c, d_in, stride_f, n, inv, g, 0, true, d_out, &flags, n_out))) return rc;
      hint_from_leaf_grid(c, out_xyzi, *n_out, g, leaf);
    }
  }
  if (!on_device) {
    HIPCHK(c, hipMemcpyAsync(out_xyzi, d_out, sizeof(float) * 4 * (size_t)*n_out, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
  }
  HIPCHK(c, hipGetLastError());
  return RGC_OK;
}

// rgc_voxelgrid for a DEVICE cloud in two halves.  begin enqueues the filter on the box kept from the previous cloud of this leaf size and
// returns; end waits for it, looks at its flags and returns the point count -- repeating the filter through rgc_voxelgrid when the kept box
// did not hold the cloud (the input must therefore stay untouched in between).  Without a kept box begin is the whole rgc_voxelgrid.
// What it is for: the odometer's sub-map filter (RGC_odometer.cpp:985-991) depends on the pose of the PREVIOUS frame only, so a caller
// can start it when that frame ends and collect the result after the next sweep's own filter -- its 55 us of kernels and the read-back
// of its count are off the frame's critical path.  Other rgc_voxelgrid calls may run in between (they come later in stream order and
// use other result words); only ONE begin may be open per context.
int rgc_voxelgrid_begin(rgc_ctx* c, const float* d_xyzi, int n, int stride_bytes, float leaf, float* d_out) {
  if (!c || !d_xyzi || !d_out || n < 0 || n > (1 << 27)) return RGC_ERR_INVALID;
  if (stride_bytes < 12 || (stride_bytes & 3) || stride_bytes > 4096 || !(leaf > 0.f) || !std::isfinite(leaf)) return fail(c, RGC_ERR_INVALID, "bad stride or leaf size");
  if (c->vg_pend.active) return fail(c, RGC_ERR_INVALID, "rgc_voxelgrid_begin: the previous one has not been ended");
  if (n > 0) { int rk = check_device_range(c, d_xyzi, (size_t)n * stride_bytes - (stride_bytes - 12), "rgc_voxelgrid_begin: d_xyzi"); if (rk) return rk; rk = check_device_range(c, d_out, (size_t)n * 16, "rgc_voxelgrid_begin: d_out"); if (rk) return rk; }
  HIPCHK(c, hipSetDevice(c->device));
  rgc_ctx::VgPending& pd_ = c->vg_pend;
  pd_ = rgc_ctx::VgPending{};
  pd_.d_in = d_xyzi; pd_.n = n; pd_.stride_bytes = stride_bytes; pd_.leaf = leaf; pd_.d_out = d_out;
  rgc_ctx::VgBox* box = nullptr;
  for (auto& b : c->vg_box) if (b.leaf == leaf) box = &b;
  bool enqueued = false;
  if (n > 0 && box && box->valid) {
    constexpr int kPadSparse = 32, kPadDense = 8;   // as in rgc_voxelgrid
    rgck::LeafGrid ps = box->g, pdg = box->g;
    for (int a = 0; a < 3; a++) { ps.minb[a] -= kPadSparse; ps.div[a] += 2 * kPadSparse; pdg.minb[a] -= kPadDense; pdg.div[a] += 2 * kPadDense; }
    const bool sparse = vg_rows_fit(ps, n);
    const double dcell = (double)pdg.div[0] * (double)pdg.div[1] * (double)pdg.div[2];
    if (sparse || dcell <= (double)c->prm.max_cells) {
      int rc = voxelgrid_rows(c, d_xyzi, stride_bytes / 4, n, 1.0f / leaf, sparse ? ps : pdg, (sparse ? kPadSparse : kPadDense) / 2, !sparse, d_out, nullptr,
                              nullptr, c->h_vg);
      if (rc) return rc;
      pd_.g = sparse ? ps : pdg;
      enqueued = true;
    }
  }
  if (!enqueued) {  // no box to trust yet: the whole filter now
    int rc = rgc_voxelgrid(c, d_xyzi, n, stride_bytes, leaf, d_out, &pd_.n_out, 1);
    if (rc) return rc;
    pd_.ready = true;
  }
  pd_.active = true;
  return RGC_OK;
}

int rgc_voxelgrid_end(rgc_ctx* c, int* n_out) {
  if (!c || !n_out) return RGC_ERR_INVALID;
  rgc_ctx::VgPending& pd_ = c->vg_pend;
  if (!pd_.active) return fail(c, RGC_ERR_INVALID, "rgc_voxelgrid_end without rgc_voxelgrid_begin");
  pd_.active = false;
  if (pd_.ready) { *n_out = pd_.n_out; return RGC_OK; }
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipEventSynchronize(c->vg_done));
  const int flags = c->h_vg[0], no = c->h_vg[2];
  if (flags & 1) return fail(c, RGC_ERR_NONFINITE, "cloud contains non-finite coordinates (PCL skips them; remove NaNs first)");
  rgc_ctx::VgBox* box = nullptr;
  for (auto& b : c->vg_box) if (b.leaf == pd_.leaf) box = &b;
  if (box && (flags & 6)) box->valid = false;  // outside: measure and repeat now; near a face: measure at the next call
  if (!(flags & 2)) { *n_out = no; hint_from_leaf_grid(c, pd_.d_out, no, pd_.g, pd_.leaf); return RGC_OK; }
  return rgc_voxelgrid(c, pd_.d_in, pd_.n, pd_.stride_bytes, pd_.leaf, pd_.d_out, n_out, 1);
}


// ---- A1-A8: ScanRegistration::laserCloudHandler on the device (src/scanRegistration.cpp:89-730) ----
static void host_eig3_sym(const double S[6], double ev[3], double V[9]) {  // Jacobi; eigenvalues ASCENDING, columns of V
  double A[3][3] = {{S[0], S[1], S[2]}, {S[1], S[3], S[4]}, {S[2], S[4], S[5]}}, U[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
  for (int sweep = 0; sweep < 60; sweep++) {
    const double off = A[0][1] * A[0][1] + A[0][2] * A[0][2] + A[1][2] * A[1][2];
    const double dg = A[0][0] * A[0][0] + A[1][1] * A[1][1] + A[2][2] * A[2][2];
    if (off <= 1e-40 * dg || off == 0.0) break;
    for (int p = 0; p < 2; p++)
      for (int q = p + 1; q < 3; q++) {
        if (A[p][q] == 0.0) continue;
        const double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
        const double cc = 1.0 / std::sqrt(t * t + 1.0), ss = t * cc;
        for (int k = 0; k < 3; k++) { const double a = A[k][p], b = A[k][q]; A[k][p] = cc * a - ss * b; A[k][q] = ss * a + cc * b; }
        for (int k = 0; k < 3; k++) { const double a = A[p][k], b = A[q][k]; A[p][k] = cc * a - ss * b; A[q][k] = ss * a + cc * b; }
        for (int k = 0; k < 3; k++) { const double a = U[k][p], b = U[k][q]; U[k][p] = cc * a - ss * b; U[k][q] = ss * a + cc * b; }
      }
  }
  int o[3] = {0, 1, 2};
  const double e[3] = {A[0][0], A[1][1], A[2][2]};
  for (int i = 0; i < 2; i++) for (int j = i + 1; j < 3; j++) if (e[o[j]] < e[o[i]]) std::swap(o[i], o[j]);
  for (int j = 0; j < 3; j++) { ev[j] = e[o[j]]; for (int i = 0; i < 3; i++) V[i * 3 + j] = U[i][o[j]]; }
}

void rgc_default_fe_params(rgc_fe_params* p) {
  if (!p) return;
  p->n_scans = 16; p->min_range = 0.5; p->max_range = 80.0; p->use_intensity = 1;  // launch/run.launch:6,12-13,18
}

static int frontend_impl(rgc_ctx* c, const float* xyzi, int n, int stride_bytes, const rgc_fe_params* prm, rgc_fe_out* out, int on_device, bool allow_spec = true);
int rgc_frontend(rgc_ctx* c, const float* xyzi, int n, int stride_bytes, const rgc_fe_params* prm, rgc_fe_out* out) {
  return frontend_impl(c, xyzi, n, stride_bytes, prm, out, 0);
}
// the same with the sweep already on the device (e.g. rgc_pc2_unpack(..., out_on_device = 1)): no host copy of the input
int rgc_frontend_device(rgc_ctx* c, const float* d_xyzi, int n, int stride_bytes, const rgc_fe_params* prm, rgc_fe_out* out) {
  return frontend_impl(c, d_xyzi, n, stride_bytes, prm, out, 1);
}
static int frontend_impl(rgc_ctx* c, const float* xyzi, int n, int stride_bytes, const rgc_fe_params* prm, rgc_fe_out* out, int on_device, bool allow_spec) {
  if (!c || !xyzi || !prm || !out || n < 0 || n > (1 << 27)) return RGC_ERR_INVALID;
  if (n > (1 << 24)) return fail(c, RGC_ERR_INVALID, "sweep has %d points, the front-end's limit is 2^24", n);  // 32-bit sizes and candidate lists below
  if (stride_bytes < 16 || (stride_bytes & 3) || stride_bytes > 4096) return fail(c, RGC_ERR_INVALID, "front-end needs x,y,z,intensity: stride_bytes >= 16");
  const int NS = prm->n_scans;
  if (NS != 16 && NS != 32 && NS != 64) return fail(c, RGC_ERR_INVALID, "only 16, 32 or 64 scan lines (scanRegistration.cpp:69-72)");
  out->n_cloud = out->n_sharp = out->n_sharp_own = out->n_flat = out->n_inten = out->n_ground = 0;
  out->ground_valid = 0;
  c->fe_n_cloud = 0;
  memset(out->ring_count, 0, sizeof(out->ring_count));
  if (n == 0) return RGC_OK;
  HIPCHK(c, hipSetDevice(c->device));
  hipStream_t s = c->stream;
  const int stride_f = stride_bytes / 4;
  const float* d_in;
  int rc = stage_in(c, xyzi, n, stride_bytes, on_device, &d_in);
  if (rc) return rc;
  const int nb = rgck::fe_blocks(n);
  enum { RING, RANK, HIST, META, ST, CL, INUM2, INUM, RANGE, ANGLE, CURV, CURV2, ICURV, DSRC, OSRC, PICK, IPICK, LAB, ILAB, GMARK, MULT, SCNT,
         SPOS, PART, OUTD, SLOTS, FLAGS, SHARP, FLAT, INTEN, GLIST, BSUM, SORTC, SORTI };
  const int nu = NS * 6, fcap = nu * 41;
  constexpr size_t kTailFlags = 304, kTailSt = 336, kTailFeat = 496;  // see OUTD below
  const size_t n4 = (size_t)4 * n;
  const size_t sizes[34] = {n4, n4, (size_t)4 * 64 * nb, 4u * 132, 4u * 8, 4 * n4, n4, n4, n4, n4, n4, n4, n4, n4,
                            n4, n4, n4, n4, n4, n4, n4, n4, n4, (size_t)8 * 11 * nb, kTailFeat + 60u * (size_t)fcap,
                            4u * (size_t)nu * rgck::fe_slot_ints(), 4u * 8, 64u, 64u, 64u, 16u * 10 * (size_t)n, 4u * ((size_t)n / 2048 + 4), n4, n4};
  // OUTD is the sweep's "tail": ground sums / fit / distance sums (doubles 0..33), the flags, the filter's start-end state and the three
  // feature clouds in ONE buffer laid out like the pinned staging area behind the meta block, so that everything the host needs at the
  // end of the sweep comes down in ONE copy and the two small blocks are initialised by ONE
  for (int b = 0; b < 34; b++) if ((rc = ensure(c, c->fe[b], sizes[b] + 64))) return rc;
#define FE(i, T) ((T*)c->fe[i].p)
  unsigned char* const tail = (unsigned char*)c->fe[OUTD].p;
  int* const d_flags = (int*)(tail + kTailFlags);
  int* const d_st = (int*)(tail + kTailSt);
  float* const d_sharp = (float*)(tail + kTailFeat);
  float* const d_flat = d_sharp + 5 * (size_t)fcap;
  float* const d_inten = d_flat + 5 * (size_t)fcap;
  const int init16[16] = {0, 0, 0, 0, 0, 0, 0, 0, INT_MAX, -1, INT_MAX, 0, 0, 0, 0, 0};  // flags (zero) + the filter's state
  memcpy(c->h_small + 32, init16, sizeof(init16));
  HIPCHK(c, hipMemcpyAsync(d_flags, c->h_small + 32, sizeof(init16), hipMemcpyHostToDevice, s));
  rgck::FeParams fp{NS, prm->min_range, prm->max_range};
  rgck::fe_filter(s, d_in, stride_f, n, fp, FE(RING, int), d_st, FE(RANK, int), FE(HIST, int));
  rgck::fe_half(s, d_in, stride_f, n, FE(RING, int), d_st);
  rgck::fe_bucket(s, d_in, stride_f, n, NS, FE(RING, int), FE(RANK, int), FE(HIST, int), FE(META, int), d_st, FE(CL, float4), FE(INUM2, int),
                  FE(PICK, int), FE(IPICK, int), FE(LAB, int), FE(ILAB, int));
  // pinned staging: [0, 1024) meta + ground sums + flags, then the three feature clouds
  const size_t stage_need = 1024 + 3 * 20u * (size_t)fcap;
  if (c->h_stage_cap < stage_need) {
    if (c->h_stage) (void)hipHostFree(c->h_stage);
    c->h_stage = nullptr; c->h_stage_cap = 0;
    HIPCHK(c, hipHostMalloc((void**)&c->h_stage, stage_need, hipHostMallocDefault));
    c->h_stage_cap = stage_need;
  }
  int* meta = (int*)c->h_stage;                       // 129 ints
  int* fl = (int*)(c->h_stage + 832);                 // 8 ints ([528, 800): the ground sums, fit and distance sums)
  unsigned char* h_feat = c->h_stage + 1024;
  // The sweep's size after the range filter and its ring sizes are known on the device (k_fe_hist_scan); the host needs them only to
  // size launches and the selection kernel's LDS.  From the second sweep of a sequence on it does not wait for them: launches are sized
  // by the raw point count, the kernels read the size themselves (csp), the selection kernel's window by the largest ring of the
  // PREVIOUS sweep plus a quarter -- if a ring outgrows that (flag bit 1), the sweep is done again the slow way.
  const bool spec = allow_spec && c->fe_spec_on && c->fe_last_ns == NS && c->fe_last_max_ring > 0 && !out->cloud;
  int cs = n, max_ring = 0;
  const int* csp = nullptr;
  if (spec) {
    csp = FE(META, int) + 128;
    max_ring = std::min(n, c->fe_last_max_ring + c->fe_last_max_ring / 4 + 64);
  } else {
    HIPCHK(c, hipMemcpyAsync(meta, FE(META, int), sizeof(int) * 129, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    cs = meta[128];
    out->n_cloud = cs;
    for (int r = 0; r < NS; r++) { out->ring_count[r] = meta[r]; max_ring = std::max(max_ring, meta[r]); }
    if (cs == 0) return RGC_OK;
    if (out->cloud && out->cloud_cap < cs) return fail(c, RGC_ERR_INVALID, "cloud_cap %d < %d points", out->cloud_cap, cs);
  }
  rgck::fe_stencils(s, FE(CL, float4), cs, csp, FE(RANGE, float), FE(ANGLE, float), FE(INUM2, int), FE(INUM, int), FE(CURV, float), FE(CURV2, float),
                    FE(ICURV, float), FE(DSRC, float), FE(OSRC, float), FE(PICK, int));
  // A5: ground set (with multiplicities) -> weighted centroid / covariance -> plane (scanRegistration.cpp:308-431)
  rgck::fe_ground(s, FE(CL, float4), cs, csp, NS, FE(RANGE, float), FE(META, int), FE(GMARK, int), FE(MULT, int), FE(SCNT, int), FE(PART, double), FE(OUTD, double),
                  FE(OUTD, double) + 16);
  // OUTD: [0..10] the ground sums, [16..31] the plane fit, [32..33] the distance sums -- fitted on the device, read back with the features
  rgck::fe_ground_dist(s, FE(CL, float4), cs, csp, FE(MULT, int), FE(OUTD, double) + 16, FE(PART, double), FE(OUTD, double) + 32);
  // /laser_cloud_ground: pushes in reference order (with duplicates); empty when no ground seed was found
  // (only when the caller takes the list: the chained frame body does not, and these are four launches)
  if (out->ground_pts && out->ground_cap > 0) {
    rgck::exclusive_scan(s, FE(SCNT, int), FE(SPOS, int), cs, FE(BSUM, int));
    const int gcap_dev = 10 * n;
    rgck::fe_ground_list(s, FE(CL, float4), cs, csp, NS, FE(RANGE, float), FE(META, int), FE(SCNT, int), FE(SPOS, int), FE(GLIST, float4), gcap_dev);
  }
  // A7 + A8
  rgck::fe_select(s, FE(CL, float4), NS, FE(META, int), FE(CURV, float), FE(CURV2, float), FE(ICURV, float), FE(INUM, int), FE(GMARK, int),
                  FE(PICK, int), FE(IPICK, int), FE(LAB, int), FE(ILAB, int), FE(SLOTS, int), d_flags, max_ring, FE(SORTC, int), FE(SORTI, int));
  rgck::fe_emit(s, FE(CL, float4), NS, FE(SLOTS, int), FE(DSRC, float), FE(OSRC, float), d_sharp, d_flat, d_inten, fcap,
                d_flags + 4);
  // flags and the three feature clouds (at their capacity: ~80 kB each for 16 rings) come down together into pinned memory, one
  // synchronisation; the counts decide how much of each is handed to the caller
  double* gd = (double*)(c->h_stage + 528);            // 34 doubles behind the 129 meta ints
  static_assert(528 + kTailFlags == 832 && 528 + kTailFeat == 1024, "the device tail mirrors the staging area from gd on");
  HIPCHK(c, hipMemcpyAsync(gd, tail, kTailFeat + 60u * (size_t)fcap, hipMemcpyDeviceToHost, s));
  if (spec) HIPCHK(c, hipMemcpyAsync(meta, FE(META, int), sizeof(int) * 129, hipMemcpyDeviceToHost, s));
  if (out->cloud) HIPCHK(c, hipMemcpyAsync(out->cloud, FE(CL, float4), sizeof(float) * 4 * (size_t)cs, hipMemcpyDeviceToHost, s));
  HIPCHK(c, hipStreamSynchronize(s));
  if (spec) {
    if (fl[0] & 2) {  // a ring outgrew the window sized from the previous sweep (or really holds an oversize sector): the slow way decides
      c->fe_last_max_ring = 0;
      return frontend_impl(c, xyzi, n, stride_bytes, prm, out, on_device, false);
    }
    cs = meta[128];
    out->n_cloud = cs;
    max_ring = 0;
    for (int r = 0; r < NS; r++) { out->ring_count[r] = meta[r]; max_ring = std::max(max_ring, meta[r]); }
    if (cs == 0) return RGC_OK;
  }
  if (fl[0] & 2) return fail(c, RGC_ERR_INVALID, "a ring sector holds more than 2048 points");
  c->fe_n_cloud = cs;
  c->fe_last_ns = NS; c->fe_last_max_ring = max_ring;
  {  // ground message (:403-430) from the sums, the fit and the distance sums that just came down
    const long long gsize = (long long)(gd[10] + 0.5);
    if (gsize > 0) {
      const double* nrm = gd + 19;
      const double* V = gd + 22;
      const double* d2 = gd + 32;
      const double laderH = 0.56;  // :39
      double distance = d2[1] / d2[0], src1 = d2[0] / (double)gsize;  // :403-404
      if ((distance / laderH) > 1.1 || (distance / laderH) < 0.9) distance = laderH;  // :405-409
      if (src1 < 0.9) distance = 0.9 * laderH + 0.1 * distance;                       // :410-413
      double* g = out->groundparam;  // groundparam.msg order, :420-430
      g[0] = nrm[0]; g[1] = nrm[1]; g[2] = nrm[2];
      g[3] = V[1]; g[4] = V[4]; g[5] = V[7];
      g[6] = V[2]; g[7] = V[5]; g[8] = V[8];
      g[9] = distance; g[10] = 1 - src1;
      out->ground_valid = 1;
      out->n_ground = (int)gsize;
      if (out->ground_pts && out->ground_cap > 0) {
        const long long m = gsize < out->ground_cap ? gsize : out->ground_cap;
        HIPCHK(c, hipMemcpyAsync(out->ground_pts, FE(GLIST, float4), sizeof(float) * 4 * (size_t)m, hipMemcpyDeviceToHost, s));
      }
    }
  }
  const int ns = fl[4], nf = fl[5], ni = fl[6];
  out->n_sharp_own = ns; out->n_flat = nf; out->n_inten = ni;
  const bool add_inten = prm->use_intensity && ((double)ns / (double)nf < 0.3);  // :645-656
  out->n_sharp = ns + (add_inten ? ni : 0);
  if (out->feat_cap < out->n_sharp || out->feat_cap < nf || out->feat_cap < ni) return fail(c, RGC_ERR_INVALID, "feat_cap too small");
  if (ns) memcpy(out->sharp, h_feat, 20u * (size_t)ns);
  if (nf) memcpy(out->flat, h_feat + 20u * (size_t)fcap, 20u * (size_t)nf);
  if (ni) memcpy(out->inten, h_feat + 40u * (size_t)fcap, 20u * (size_t)ni);
  if (add_inten && ni) memcpy(out->sharp + 5 * (size_t)ns, h_feat + 40u * (size_t)fcap, 20u * (size_t)ni);
  const struct { void* dst; int src; } diag[7] = {{out->curvature, CURV}, {out->curvature2, CURV2}, {out->inten_curvature, ICURV}, {out->label, LAB},
                                                   {out->inten_label, ILAB}, {out->picked, PICK}, {out->ground_marked, GMARK}};
  for (auto& d : diag) if (d.dst) HIPCHK(c, hipMemcpyAsync(d.dst, c->fe[d.src].p, 4u * (size_t)cs, hipMemcpyDeviceToHost, s));
  HIPCHK(c, hipStreamSynchronize(s));
  HIPCHK(c, hipGetLastError());
#undef FE
  return RGC_OK;
}

int rgc_frontend_cloud_device(rgc_ctx* c, float** d_cloud, int* n) {
  if (!c || !d_cloud || !n) return RGC_ERR_INVALID;
  *d_cloud = c->fe_n_cloud > 0 ? (float*)c->fe[5].p : nullptr;  // CL: float4 {x, y, z, ring + 0.1 relTime}, ring-major
  *n = c->fe_n_cloud;
  return RGC_OK;
}

int rgc_get_stats(rgc_ctx* c, rgc_stats* out) {
  if (!c || !out) return RGC_ERR_INVALID;
  HIPCHK(c, hipSetDevice(c->device));  // (a tripped guard or a lazy target is resolved below: kernels)
  { int rc = validate_clouds(c, /*whole_target=*/false); if (rc) return rc; }
  if (c->tgt.ready) { int rc = fetch_nvox(c); if (rc) return rc; }
  // queries the bulk kNN kernel handed to the cooperative kernel (first int of the deferred-list buffer)
  if (!c->deferred_known) c->stats.deferred_target = c->stats.deferred_source = 0;
  if (!c->deferred_known) {
    HIPCHK(c, hipStreamSynchronize(c->stream2));
    if (c->tgt.ready && c->tgt.segs.p) HIPCHK(c, hipMemcpyAsync(&c->stats.deferred_target, c->tgt.segs.p, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    if (c->src.ready && c->src.segs.p) HIPCHK(c, hipMemcpyAsync(&c->stats.deferred_source, c->src.segs.p, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
  }
  c->stats.searched_target = c->tgt.ready ? c->tgt.n : 0;
  if (c->tgt.ready && c->tgt.searched_known >= 0 && !c->trace_cache) {
    c->stats.searched_target = c->tgt.searched_known;  // (fetched once per preparation)
  } else if (c->tgt.ready && c->tgt.cache_on && c->tgt.seed_warm && c->tgt.cache_small.p) {
    // the neighbour-list cache's list lengths and its epoch word (== the frame: everything was searched)
    int h[rgck::kTodoLists + 3];
    HIPCHK(c, hipMemcpyAsync(h, c->tgt.cache_small.p, sizeof(h), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    const bool redo = h[rgck::kTodoLists] == c->tgt.cache_frame || h[rgck::kTodoLists + 1 + ((c->tgt.cache_frame - 1) & 1)] == c->tgt.cache_frame - 1;
    if (c->trace_cache) {
      int mx = 0; long long sm = 0;
      for (int l = 0; l < rgck::kTodoLists; l++) { mx = std::max(mx, h[l]); sm += h[l]; }
      fprintf(stderr, "[rgc] cache: frame %d epoch %d overflow %d %d lists sum %lld max %d cap %d e2 %d slack %g searched_lists %d\n", c->tgt.cache_frame, h[rgck::kTodoLists],
              h[rgck::kTodoLists + 1], h[rgck::kTodoLists + 2], sm, mx, c->tgt.todo_cap, c->tgt.cache_e2, 4.0 * 1.7320508 * 1.1 * std::ldexp(1.0, c->tgt.cache_e2 - 24), (int)c->tgt.cache_searched_lists);
    }
    if (!redo && c->tgt.cache_searched_lists) {
      int sum = 0;
      for (int l = 0; l < rgck::kTodoLists; l++) sum += std::min(h[l], c->tgt.todo_cap);
      c->stats.searched_target = sum;
    }
    c->tgt.searched_known = c->stats.searched_target;
  }
  *out = c->stats;
  return RGC_OK;
}

#ifdef RGC_LAB_TURN
extern "C" RGC_API int rgc_lab_turn(rgc_ctx*, unsigned long long* out8) { rgck::lab_turn(out8); return RGC_OK; }
#endif
#if defined(RGC_LAB) || defined(RGC_LAB_BLK)
RGC_API int rgc_lab_blocks(rgc_ctx*, long long* out65536) { rgck::lab_blocks(out65536); return RGC_OK; }
#endif
#ifdef RGC_LAB
RGC_API int rgc_lab_lm_ts(rgc_ctx* c, unsigned long long* out16) { rgck::lab_lm_ts(out16, c->stream); return RGC_OK; }
RGC_API int rgc_lab_why(rgc_ctx*, int* out8) { rgck::lab_why(out8); return RGC_OK; }
RGC_API int rgc_lab_declines(rgc_ctx*, int* out16) { rgck::lab_declines(out16); return RGC_OK; }
RGC_API int rgc_lab_iters(rgc_ctx*, unsigned long long* out8) { rgck::lab_iters(out8); return RGC_OK; }
RGC_API int rgc_lab_wave_ts(rgc_ctx* c, long long* out16384) { (void)hipStreamSynchronize(c->stream); rgck::lab_wave_ts(out16384, c->stream2); return RGC_OK; }
// developer build only (-DRGC_LAB): the deferred-query list of a cloud as the bulk kNN kernel left it
RGC_API int rgc_lab_deferred(rgc_ctx* c, int is_target, int* idx, float* thr, int cap, int* count) {
  Cloud& cl = is_target ? c->tgt : c->src;
  HIPCHK(c, hipStreamSynchronize(c->stream2));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  int cnt = 0;
  HIPCHK(c, hipMemcpy(&cnt, cl.segs.p, sizeof(int), hipMemcpyDeviceToHost));
  *count = cnt;
  const int m = cnt < cap ? cnt : cap;
  HIPCHK(c, hipMemcpy(idx, (const int*)cl.segs.p + 16, sizeof(int) * m, hipMemcpyDeviceToHost));
  HIPCHK(c, hipMemcpy(thr, (const int*)cl.segs.p + 16 + cl.n, sizeof(float) * m, hipMemcpyDeviceToHost));
  return RGC_OK;
}
#endif

int rgc_device_alloc(rgc_ctx* c, size_t bytes, void** p) {
  if (!c || !p) return RGC_ERR_INVALID;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipMalloc(p, bytes));
  return RGC_OK;
}
int rgc_device_free(rgc_ctx* c, void* p) {
  if (!c) return RGC_ERR_INVALID;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamSynchronize(c->stream2));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipFree(p));
  return RGC_OK;
}
int rgc_host_alloc(size_t bytes, void** p) {
  if (!p) return RGC_ERR_INVALID;
  *p = nullptr;
  return hipHostMalloc(p, bytes ? bytes : 1, hipHostMallocPortable) == hipSuccess ? RGC_OK : RGC_ERR_HIP;
}
int rgc_host_free(void* p) {
  if (!p) return RGC_OK;
  return hipHostFree(p) == hipSuccess ? RGC_OK : RGC_ERR_HIP;
}

int rgc_upload(rgc_ctx* c, void* d, const void* h, size_t bytes) {
  if (!c || !d || !h) return RGC_ERR_INVALID;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, c->stream));
  // behind a pending map preparation this copy is NOT covered by the mark the scan's stream waits for (rgc_set_source_device in rgc_hip.h):
  // remembered, so that a source set next is ordered behind it -- at the price of that one frame's overlap -- instead of racing it
  if (c->main_has_target_prep) c->main_late_producer = true;
  return RGC_OK;
}
int rgc_download(rgc_ctx* c, void* h, const void* d, size_t bytes) {
  if (!c || !d || !h) return RGC_ERR_INVALID;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return RGC_OK;
}
int rgc_synchronize(rgc_ctx* c) {
  if (!c) return RGC_ERR_INVALID;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamSynchronize(c->stream2));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->main_has_target_prep = false;
  return RGC_OK;
}
void* rgc_stream(rgc_ctx* c) { return c ? (void*)c->stream : nullptr; }

int rgc_profile_enable(rgc_ctx* c, int on) {
  if (!c) return RGC_ERR_INVALID;
  c->prof_on = on != 0;
  return RGC_OK;
}
int rgc_profile_select(rgc_ctx* c, unsigned kind_mask) {
  if (!c) return RGC_ERR_INVALID;
  c->prof_mask = kind_mask;
  return RGC_OK;
}
int rgc_profile_reset(rgc_ctx* c) {
  if (!c) return RGC_ERR_INVALID;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamSynchronize(c->stream2));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  prof_collect(c);
  for (int i = 0; i < kProfKinds; i++) { c->prof_launches[i] = 0; c->prof_ms[i] = 0; c->prof_points[i] = 0; }
  return RGC_OK;
}
int rgc_profile_get(rgc_ctx* c, int kind, long long* launches, double* total_ms, long long* total_points) {
  if (!c || kind < 0 || kind >= kProfKinds) return RGC_ERR_INVALID;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamSynchronize(c->stream2));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  prof_collect(c);
  if (launches) *launches = c->prof_launches[kind];
  if (total_ms) *total_ms = c->prof_ms[kind];
  if (total_points) *total_points = c->prof_points[kind];
  return RGC_OK;
}
const char* rgc_profile_name(int kind) {
  static const char* names[kProfKinds] = {"grid_build", "knn_cov_target", "voxel_build", "linearize", "compute_error", "fitness",
                                          "knn_cov_source", "knn_coop_target", "knn_coop_source"};
  return (kind >= 0 && kind < kProfKinds) ? names[kind] : "?";
}

// ---- f1: scan-to-map FEATURE registration of the mapping node (RGC_mapping.cpp:1069-1358) --------------------------------
int rgc_mapreg_set_maps(rgc_ctx* c, const float* corner_map, int n_corner, const float* surf_map, int n_surf, int stride_bytes) {
  if (!c || !corner_map || !surf_map) return RGC_ERR_INVALID;
  if (stride_bytes < 12 || (stride_bytes & 3) || stride_bytes > 4096) return fail(c, RGC_ERR_INVALID, "stride_bytes must be a multiple of 4 and >= 12");
  if (n_corner < 5 || n_surf < 5) return fail(c, RGC_ERR_TOO_FEW_POINTS, "feature maps need at least 5 points each (5-NN)");
  if (n_corner > (1 << 27) || n_surf > (1 << 27)) return fail(c, RGC_ERR_INVALID, "feature map larger than 2^27 points");
  HIPCHK(c, hipSetDevice(c->device));
  const float* src[2] = {corner_map, surf_map};
  const int n[2] = {n_corner, n_surf};
  for (int m = 0; m < 2; m++) {
    Cloud& cl = c->mr_map[m];
    cl.ready = false;
    const size_t bytes = (size_t)n[m] * stride_bytes;
    int rc = ensure(c, cl.in_copy, bytes);
    if (rc) return rc;
    HIPCHK(c, hipMemcpyAsync(cl.in_copy.p, src[m], bytes - (stride_bytes - 12), hipMemcpyHostToDevice, c->stream));
    cl.in = (const float*)cl.in_copy.p;
    cl.stride_f = stride_bytes / 4;
    cl.n = n[m];
    if ((rc = prepare_map_grid(c, cl, kMapregCell[m]))) return rc;
  }
  return RGC_OK;
}

static int mapreg_upload_features(rgc_ctx* c, int slot, const float* feat, int n) {
  int rc;
  if ((rc = ensure(c, c->mr_feat[slot], sizeof(float) * 4 * (size_t)(n > 0 ? n : 1)))) return rc;
  if ((rc = ensure(c, c->mr_fac[slot], sizeof(double) * 8 * (size_t)(n > 0 ? n : 1)))) return rc;
  if (n > 0) HIPCHK(c, hipMemcpyAsync(c->mr_feat[slot].p, feat, sizeof(float) * 4 * (size_t)n, hipMemcpyHostToDevice, c->stream));
  return RGC_OK;
}

int rgc_mapreg_associate(rgc_ctx* c, int kind, const float* feat_xyzw, int n, const double q_xyzw[4], const double t[3], double* factors8,
                         int* n_valid) {
  if (!c || !feat_xyzw || !q_xyzw || !t || n < 0 || n > (1 << 27) || (kind != 0 && kind != 1)) return RGC_ERR_INVALID;
  if (!c->mr_map[kind].ready) return fail(c, RGC_ERR_NO_INPUT, "rgc_mapreg_set_maps first");
  HIPCHK(c, hipSetDevice(c->device));
  int rc = mapreg_upload_features(c, kind, feat_xyzw, n);
  if (rc) return rc;
  const Cloud& m = c->mr_map[kind];
  const rgck::MapregAssoc one{(const float*)c->mr_feat[kind].p, n, kind == 0 ? 1 : 0, rgck::Quat{q_xyzw[0], q_xyzw[1], q_xyzw[2], q_xyzw[3]},
                              {t[0], t[1], t[2]}, (const float4*)m.P.p, (const int*)m.start.p, m.grid, (double*)c->mr_fac[kind].p, nullptr};
  rgck::mapreg_associate(c->stream, &one, 1);
  std::vector<double> tmp;
  double* dst = factors8;
  if (!dst) { tmp.resize((size_t)8 * (n > 0 ? n : 1)); dst = tmp.data(); }
  if (n > 0) HIPCHK(c, hipMemcpyAsync(dst, c->mr_fac[kind].p, sizeof(double) * 8 * (size_t)n, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipGetLastError());
  if (n_valid) {
    int cnt = 0;
    for (int i = 0; i < n; i++) cnt += dst[(size_t)8 * i + 7] != 0.0;
    *n_valid = cnt;
  }
  return RGC_OK;
}

int rgc_mapreg_optimize(rgc_ctx* c, const float* corner_cur, int n_ccur, const float* surf_cur, int n_scur, const float* corner_last,
                        int n_clast, const float* surf_last, int n_slast, const rgc_mapreg_ground* ground_cur, const rgc_mapreg_ground* ground_last,
                        const rgc_mapreg_imu* imu, double poses[14], rgc_mapreg_report report[2], int* gate_failed) {
  if (!c || !poses || n_ccur < 0 || n_scur < 0 || n_clast < 0 || n_slast < 0) return RGC_ERR_INVALID;
  if (n_ccur > (1 << 27) || n_scur > (1 << 27) || n_clast > (1 << 27) || n_slast > (1 << 27)) return fail(c, RGC_ERR_INVALID, "feature cloud larger than 2^27 points");
  if ((n_ccur && !corner_cur) || (n_scur && !surf_cur) || (n_clast && !corner_last) || (n_slast && !surf_last)) return RGC_ERR_INVALID;
  if (!c->mr_map[0].ready || !c->mr_map[1].ready) return fail(c, RGC_ERR_NO_INPUT, "rgc_mapreg_set_maps first");
  if (report) memset(report, 0, sizeof(rgc_mapreg_report) * 2);
  // the gate of :1069 (laserCloudCornerDSNum > 10 && laserCloudSurfDSNum > 50 && map sizes likewise)
  const bool gate = n_ccur > 10 && n_scur > 50 && c->mr_map[0].n > 10 && c->mr_map[1].n > 50;
  if (gate_failed) *gate_failed = gate ? 0 : 1;
  if (!gate) return RGC_OK;
  HIPCHK(c, hipSetDevice(c->device));
  const float* feat[4] = {corner_cur, surf_cur, corner_last, surf_last};
  const int nfeat[4] = {n_ccur, n_scur, n_clast, n_slast};
  const rgc_mapreg_ground* const ground[2] = {ground_cur, ground_last};
  int rc;
  for (int s = 0; s < 4; s++)
    if ((rc = mapreg_upload_features(c, s, feat[s], nfeat[s]))) return rc;
  const int nb = std::max(rgck::mapreg_blocks(n_ccur, n_scur), rgck::mapreg_blocks(n_clast, n_slast));
  if ((rc = ensure(c, c->mr_partials, sizeof(double) * 2 * rgck::kAccum * (size_t)(nb > 0 ? nb : 1)))) return rc;
  for (int iter = 0; iter < 2; iter++) {  // :1076
    // association at the current estimate of both poses (frozen during the solve); the factor counts (the reference's
    // corner_num / surf_num ...) ride home with the first evaluation's synchronisation
    int* dcnt = (int*)c->mr_small.p + 8;
    HIPCHK(c, hipMemsetAsync(dcnt, 0, 4 * sizeof(int), c->stream));
    rgck::MapregAssoc sets[4];
    for (int s = 0; s < 4; s++) {
      const double* q = poses + 7 * (s / 2);
      const Cloud& m = c->mr_map[s & 1];
      sets[s] = rgck::MapregAssoc{(const float*)c->mr_feat[s].p, nfeat[s], (s & 1) == 0 ? 1 : 0, rgck::Quat{q[0], q[1], q[2], q[3]}, {q[4], q[5], q[6]},
                                  (const float4*)m.P.p, (const int*)m.start.p, m.grid, (double*)c->mr_fac[s].p, dcnt + s};
    }
    rgck::mapreg_associate(c->stream, sets, 4);  // the four loops of :1092-1282 side by side
    HIPCHK(c, hipMemcpyAsync(c->h_small + 40, dcnt, 4 * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    // ceres::Solve restated: trust-region LM, <= 6 iterations (:1333-1341), Ceres 1.14 defaults: initial radius 1e4, damping diag(H)/radius clamped to [1e-6, 1e32], step accepted above a relative decrease of 1e-3
    double radius = 1e4, decrease_factor = 2.0;
    MapregSystem S, Sn;
    if ((rc = mapreg_eval(c, nfeat, poses, true, ground, imu, &S))) return rc;
    if (report) {
      report[iter].n_edge_cur = c->h_small[40]; report[iter].n_plane_cur = c->h_small[41];
      report[iter].n_edge_last = c->h_small[42]; report[iter].n_plane_last = c->h_small[43];
    }
    int it = 0, n_success = 0;
    const double initial_cost = S.cost;
    for (it = 0; it < 6; it++) {
      double gmax = 0;
      for (int a = 0; a < 12; a++) gmax = std::fmax(gmax, std::fabs(S.g[a]));
      if (gmax <= 1e-10) break;
      double A[144], rhs[12], d[12], model = 0;
      memcpy(A, S.H, sizeof(A));
      for (int a = 0; a < 12; a++) {
        A[a * 13] += std::fmin(std::fmax(S.H[a * 13], 1e-6), 1e32) / radius;  // min / max_lm_diagonal
        rhs[a] = -S.g[a];
      }
      const bool ok = chol_solve(A, rhs, d, 12);
      for (int a = 0; a < 12 && ok; a++) {  // model cost change = -d^T (g + H d / 2)
        double Hd = 0;
        for (int e = 0; e < 12; e++) Hd += S.H[a * 12 + e] * d[e];
        model -= d[a] * (S.g[a] + 0.5 * Hd);
      }
      double rho = -1.0, xn[14];
      memcpy(xn, poses, sizeof(xn));
      if (ok && model > 0) {
        for (int b = 0; b < 2; b++) {
          quat_plus(poses + 7 * b, d + 6 * b, xn + 7 * b);
          for (int a = 0; a < 3; a++) xn[7 * b + 4 + a] = poses[7 * b + 4 + a] + d[6 * b + 3 + a];
        }
        // the candidate's cost AND its normal equations in one launch: nearly every step is accepted, and an accepted step
        // needs them next (a rejected one just drops them)
        if ((rc = mapreg_eval(c, nfeat, xn, true, ground, imu, &Sn))) return rc;
        rho = (S.cost - Sn.cost) / model;
      }
      if (rho > 1e-3) {
        const double old_cost = S.cost;
        memcpy(poses, xn, sizeof(xn));
        radius = std::fmin(radius / std::fmax(1.0 / 3.0, 1.0 - std::pow(2.0 * rho - 1.0, 3)), 1e16);
        decrease_factor = 2.0;
        n_success++;
        S = Sn;
        double step2 = 0, x2 = 0;
        for (int a = 0; a < 12; a++) step2 += d[a] * d[a];
        for (int a = 0; a < 14; a++) x2 += poses[a] * poses[a];
        if (std::fabs(old_cost - S.cost) <= 1e-6 * old_cost) { it++; break; }
        if (std::sqrt(step2) <= 1e-8 * (std::sqrt(x2) + 1e-8)) { it++; break; }
      } else {
        radius /= decrease_factor;
        decrease_factor *= 2.0;
        if (radius < 1e-32) { it++; break; }
      }
    }
    const double cost = S.cost;
    if (report) { report[iter].initial_cost = initial_cost; report[iter].final_cost = cost; report[iter].iterations = it; report[iter].successful = n_success; }
  }
  for (int b = 0; b < 2; b++) {  // q_w_last.normalize(); q_w_curr.normalize(); (:1375-1376)
    double* q = poses + 7 * b;
    const double nn = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    if (nn > 0) for (int a = 0; a < 4; a++) q[a] /= nn;
  }
  return RGC_OK;
}

// ---- f2: rolling local map resident on the device (replaces the keyframe deque + per-frame re-framing + re-upload of
// src/RGC_odometer.cpp:1218-1256, 985-991, 1007) ----------------------------------------------------------------------------
static int map_reserve(rgc_ctx* c, int which, size_t points, bool preserve) {
  DevBuf& b = c->map_store[which];
  const size_t bytes = points * 16;
  if (bytes <= b.cap && b.p) return RGC_OK;
  void* np = nullptr;
  const size_t want = std::max(bytes + bytes / 2, (size_t)1 << 20);
  HIPCHK(c, hipMalloc(&np, want));
  if (b.p) {
    if (preserve && c->map_n) HIPCHK(c, hipMemcpyAsync(np, b.p, c->map_n * 16, hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipFree(b.p));
  }
  b.p = np;
  b.cap = want;
  return RGC_OK;
}

int rgc_map_reset(rgc_ctx* c, const double origin[3]) {
  if (!c) return RGC_ERR_INVALID;
  c->map_kf.clear();
  c->map_n = 0;
  c->map_dirty = true;
  c->map_ntarget = 0;
  if (c->map_bound) { c->tgt.ready = false; c->tgt.n = 0; c->corr_valid = false; c->map_bound = false; }
  for (int a = 0; a < 3; a++) c->map_origin[a] = origin ? origin[a] : 0.0;
  c->map_rev++;
  return RGC_OK;
}

int rgc_map_insert(rgc_ctx* c, const float* xyzi, int n, int stride_bytes, const double q[4], const double t[3], int on_device, int* keyframe_id) {
  if (!c || !xyzi || !q || !t || n <= 0) return RGC_ERR_INVALID;
  if (stride_bytes < 16 || (stride_bytes & 3) || stride_bytes > 4096) return fail(c, RGC_ERR_INVALID, "a keyframe is x,y,z,intensity: stride_bytes >= 16");
  if (c->map_n + (size_t)n > ((size_t)1 << 27)) return fail(c, RGC_ERR_INVALID, "the map would exceed 2^27 points");
  HIPCHK(c, hipSetDevice(c->device));
  const float* d_in;
  int rc = stage_in(c, xyzi, n, stride_bytes, on_device, &d_in);
  if (rc) return rc;
  if ((rc = map_reserve(c, c->map_cur, c->map_n + n, true))) return rc;
  // (the other buffer -- where the first eviction or re-basing compacts to -- grows with it: a first hipMalloc of that size is 9 ms in
  // whichever frame it falls)
  if ((rc = map_reserve(c, c->map_cur ^ 1, c->map_n + n, false))) return rc;
  // surroundingCloud.push_back(transformPointCloud(FullPointsLessFlat, q_w_curr, t_w_curr)) (:1237), relative to the origin
  const double tr[3] = {t[0] - c->map_origin[0], t[1] - c->map_origin[1], t[2] - c->map_origin[2]};
  rgck::transform_q(c->stream, d_in, stride_bytes / 4, n, rgck::Quat{q[0], q[1], q[2], q[3]}, tr, (float*)c->map_store[c->map_cur].p + 4 * c->map_n, 4);
  if (!on_device) HIPCHK(c, hipStreamSynchronize(c->stream));  // pre_in is re-used by the next staged call
  HIPCHK(c, hipGetLastError());
  rgc_ctx::MapKf kf{c->map_next_id++, c->map_n, n, {t[0], t[1], t[2]}};
  c->map_kf.push_back(kf);
  c->map_n += n;
  c->map_dirty = true;
  c->map_rev++;
  if (keyframe_id) *keyframe_id = kf.id;
  return RGC_OK;
}

int rgc_map_evict(rgc_ctx* c, int max_keyframes, const double center[3], double radius, int* n_evicted) {
  if (!c) return RGC_ERR_INVALID;
  if (n_evicted) *n_evicted = 0;
  std::vector<rgc_ctx::MapKf> keep;
  for (const auto& k : c->map_kf) {
    bool far = false;
    if (center && radius > 0 && &k != &c->map_kf.back()) {  // the newest keyframe always stays: an empty map cannot be committed
      const double dx = k.t[0] - center[0], dy = k.t[1] - center[1], dz = k.t[2] - center[2];
      far = std::sqrt(dx * dx + dy * dy + dz * dz) > radius;
    }
    if (!far) keep.push_back(k);
  }
  if (max_keyframes > 0 && (int)keep.size() > max_keyframes) keep.erase(keep.begin(), keep.end() - max_keyframes);  // pop_front, :1242-1247
  const int gone = (int)c->map_kf.size() - (int)keep.size();
  if (!gone) return RGC_OK;
  HIPCHK(c, hipSetDevice(c->device));
  size_t total = 0;
  for (const auto& k : keep) total += k.n;
  const int other = c->map_cur ^ 1;
  int rc = map_reserve(c, other, std::max(total, (size_t)1), false);
  if (rc) return rc;
  size_t off = 0;
  for (size_t i = 0; i < keep.size();) {  // runs of surviving neighbours move with one copy
    size_t j = i, run = 0;
    const size_t base = keep[i].off;
    while (j < keep.size() && keep[j].off == base + run) { run += keep[j].n; j++; }
    HIPCHK(c, hipMemcpyAsync((char*)c->map_store[other].p + off * 16, (const char*)c->map_store[c->map_cur].p + base * 16, run * 16,
                             hipMemcpyDeviceToDevice, c->stream));
    for (size_t k = i; k < j; k++) keep[k].off = off + (keep[k].off - base);
    off += run;
    i = j;
  }
  c->map_cur = other;
  c->map_kf.swap(keep);
  c->map_n = total;
  c->map_dirty = true;
  c->map_rev++;
  if (n_evicted) *n_evicted = gone;
  return RGC_OK;
}

int rgc_map_rebase(rgc_ctx* c, const double new_origin[3]) {
  if (!c || !new_origin) return RGC_ERR_INVALID;
  const double d[3] = {c->map_origin[0] - new_origin[0], c->map_origin[1] - new_origin[1], c->map_origin[2] - new_origin[2]};
  if (c->map_n) {
    HIPCHK(c, hipSetDevice(c->device));
    const int other = c->map_cur ^ 1;
    int rc = map_reserve(c, other, c->map_n, false);
    if (rc) return rc;
    rgck::transform_q(c->stream, (const float*)c->map_store[c->map_cur].p, 4, (int)c->map_n, rgck::Quat{0, 0, 0, 1}, d, (float*)c->map_store[other].p, 4);
    HIPCHK(c, hipGetLastError());
    c->map_cur = other;
  }
  for (int a = 0; a < 3; a++) c->map_origin[a] = new_origin[a];
  c->map_dirty = true;
  c->map_rev++;
  if (c->map_bound) {  // the committed target is in the OLD origin's coordinates: an align before the next rgc_map_commit must fail, not drift
    c->tgt.ready = false;
    c->corr_valid = false;
  }
  return RGC_OK;
}

int rgc_map_commit(rgc_ctx* c, float leaf, int* n_target) {
  if (!c || !(leaf > 0.f)) return RGC_ERR_INVALID;
  if (c->map_bound && !c->map_dirty && leaf == c->map_leaf && c->tgt.ready) {  // nothing changed: the resident target stands
    if (n_target) *n_target = c->map_ntarget;
    return RGC_OK;
  }
  // (before anything is written: the filter below writes the buffer the resident target was set from -- a commit refused behind it would
  // leave a set target whose input has been overwritten, and rgc_map_download(1) returning another cloud; tests/fuzz/fuzz_api.py)
  if (solve_in_flight(c)) return fail(c, RGC_ERR_INVALID, "a solve is in flight on this context: call rgc_align_end first");
  if (!c->map_n) return fail(c, RGC_ERR_NO_INPUT, "the map holds no keyframe");
  int rc = ensure(c, c->map_target, c->map_n * 16);
  if (rc) return rc;
  int nt = 0;
  // downSizeFilter2.setInputCloud(laserCloudsubmap); filter (:985-991) -- on the resident store, nothing crosses PCIe
  if (c->map_bound) {  // from here on the buffer no longer holds the cloud the bound target was set from: whatever fails below, that target goes
    c->map_bound = false;
    c->tgt.ready = false; c->tgt.n = 0; c->corr_valid = false;
  }
  if ((rc = rgc_voxelgrid(c, (const float*)c->map_store[c->map_cur].p, (int)c->map_n, 16, leaf, (float*)c->map_target.p, &nt, 1))) return rc;
  // setInputTarget (:1007): grid, exact-kNN covariances, Gaussian voxel map
  if ((rc = set_cloud(c, c->tgt, true, (const float*)c->map_target.p, nt, 16, true))) return rc;
  c->map_bound = true;
  c->map_dirty = false;
  c->map_leaf = leaf;
  c->map_ntarget = nt;
  if (n_target) *n_target = nt;
  return RGC_OK;
}

int rgc_map_get_info(rgc_ctx* c, rgc_map_info* out) {
  if (!c || !out) return RGC_ERR_INVALID;
  out->n_keyframes = (int)c->map_kf.size();
  out->n_points = (long long)c->map_n;
  out->n_target = c->map_bound && !c->map_dirty ? c->map_ntarget : -1;
  out->revision = c->map_rev;
  out->oldest_id = c->map_kf.empty() ? -1 : c->map_kf.front().id;
  out->newest_id = c->map_kf.empty() ? -1 : c->map_kf.back().id;
  for (int a = 0; a < 3; a++) out->origin[a] = c->map_origin[a];
  return RGC_OK;
}

int rgc_map_download(rgc_ctx* c, int which, float* out_xyzi, int cap, int* n) {
  if (!c || !n || cap < 0 || (cap && !out_xyzi)) return RGC_ERR_INVALID;
  const void* src = nullptr;
  int have = 0;
  if (which == 0) { src = c->map_store[c->map_cur].p; have = (int)c->map_n; }
  else if (which == 1) {
    if (!c->map_bound || c->map_dirty) return fail(c, RGC_ERR_NO_INPUT, "rgc_map_commit first");
    src = c->map_target.p; have = c->map_ntarget;
  } else return RGC_ERR_INVALID;
  *n = have;
  const int m = std::min(have, cap);
  if (m > 0) {
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(out_xyzi, src, (size_t)m * 16, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
  }
  return RGC_OK;
}

// ---- f3: PointCloud2 <-> device arrays ---------------------------------------------------------------------------------
int rgc_pc2_unpack(rgc_ctx* c, const void* data, int n, const rgc_pc2_layout* L, float* xyzi_out, int* ring_out, float* time_out,
                   int out_on_device) {
  if (!c || !data || !L || !xyzi_out || n < 0) return RGC_ERR_INVALID;
  if (L->point_step <= 0) return fail(c, RGC_ERR_INVALID, "point_step must be positive");
  if (n > (1 << 24)) return fail(c, RGC_ERR_INVALID, "message has %d points, the limit is 2^24", n);
  rgck::Pc2Layout K{};
  K.point_step = L->point_step;
  K.big_endian = L->is_bigendian ? 1 : 0;
  for (int f = 0; f < 6; f++) {
    int off = L->offset[f], ty = L->datatype[f];
    if (off >= 0) {
      if (ty < 1 || ty > 8) return fail(c, RGC_ERR_INVALID, "field %d: unknown PointField datatype %d", f, ty);
      const int size = (ty == 1 || ty == 2) ? 1 : (ty == 3 || ty == 4) ? 2 : (ty == 8 ? 8 : 4);
      if (off + size > L->point_step) return fail(c, RGC_ERR_INVALID, "field %d runs past point_step", f);
      // fromROSMsg<PointXYZI> maps x, y, z, intensity only from FLOAT32 fields (a mismatching datatype leaves the default)
      if (L->strict && f < 4 && ty != 7) off = -1;
    }
    K.off[f] = off;
    K.type[f] = ty;
  }
  if (n == 0) return RGC_OK;
  HIPCHK(c, hipSetDevice(c->device));
  hipStream_t s = c->stream;
  const size_t bytes = (size_t)n * L->point_step;
  int rc;
  if ((rc = ensure(c, c->pre_in, bytes))) return rc;
  HIPCHK(c, hipMemcpyAsync(c->pre_in.p, data, bytes, hipMemcpyHostToDevice, s));
  float4* d_xyzi;
  int* d_ring = nullptr;
  float* d_time = nullptr;
  if (out_on_device) {
    d_xyzi = (float4*)xyzi_out; d_ring = ring_out; d_time = time_out;
  } else {
    if ((rc = ensure(c, c->pre_out, (size_t)n * 24))) return rc;
    d_xyzi = (float4*)c->pre_out.p;
    if (ring_out) d_ring = (int*)((char*)c->pre_out.p + (size_t)n * 16);
    if (time_out) d_time = (float*)((char*)c->pre_out.p + (size_t)n * 20);
  }
  rgck::pc2_unpack(s, (const unsigned char*)c->pre_in.p, n, K, d_xyzi, d_ring, d_time);
  if (!out_on_device) {
    HIPCHK(c, hipMemcpyAsync(xyzi_out, d_xyzi, (size_t)n * 16, hipMemcpyDeviceToHost, s));
    if (ring_out) HIPCHK(c, hipMemcpyAsync(ring_out, d_ring, (size_t)n * 4, hipMemcpyDeviceToHost, s));
    if (time_out) HIPCHK(c, hipMemcpyAsync(time_out, d_time, (size_t)n * 4, hipMemcpyDeviceToHost, s));
  }
  HIPCHK(c, hipStreamSynchronize(s));
  HIPCHK(c, hipGetLastError());
  return RGC_OK;
}

int rgc_pc2_pack(rgc_ctx* c, int kind, const float* in, int n, int in_on_device, void* data_out) {
  if (!c || !in || !data_out || n < 0 || n > (1 << 24) || (kind != 0 && kind != 1)) return RGC_ERR_INVALID;   // (a message of more than 2^24 points: rgc_pc2_unpack's limit)
  if (n == 0) return RGC_OK;
  HIPCHK(c, hipSetDevice(c->device));
  const int cols = kind == 0 ? 4 : 5, step = kind == 0 ? 32 : 48;
  const float* d_in;
  int rc = stage_in(c, in, n, cols * 4, in_on_device, &d_in);
  if (rc) return rc;
  if ((rc = ensure(c, c->pre_out, (size_t)n * step))) return rc;
  rgck::pc2_pack(c->stream, d_in, cols, n, kind, (unsigned char*)c->pre_out.p);
  HIPCHK(c, hipMemcpyAsync(data_out, c->pre_out.p, (size_t)n * step, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipGetLastError());
  return RGC_OK;
}

// ---- f4: loop-closure ICP (pcl::IterativeClosestPoint as configured at RGC_mapping.cpp:2050-2069) -----------------------------
// R, t minimising sum |R p + t - q|^2 from n, sum p, sum q, sum p q^T (TransformationEstimationSVD = Umeyama without scale): SVD of
// the centred correlation through the eigen decomposition of H^T H
static void rigid_from_sums(double n, const double sp[3], const double sq[3], const double spq[9], double R[9], double t[3]) {
  double cp[3], cq[3], H[9];
  for (int a = 0; a < 3; a++) { cp[a] = sp[a] / n; cq[a] = sq[a] / n; }
  for (int a = 0; a < 3; a++)
    for (int b = 0; b < 3; b++) H[a * 3 + b] = spq[a * 3 + b] - n * cp[a] * cq[b];
  double HtH[9] = {0};
  for (int a = 0; a < 3; a++)
    for (int b = 0; b < 3; b++)
      for (int k = 0; k < 3; k++) HtH[a * 3 + b] += H[k * 3 + a] * H[k * 3 + b];
  const double S6[6] = {HtH[0], 0.5 * (HtH[1] + HtH[3]), 0.5 * (HtH[2] + HtH[6]), HtH[4], 0.5 * (HtH[5] + HtH[7]), HtH[8]};
  double ev[3], Va[9], V[9];
  host_eig3_sym(S6, ev, Va);  // ascending
  for (int a = 0; a < 3; a++) { V[a * 3 + 0] = Va[a * 3 + 2]; V[a * 3 + 1] = Va[a * 3 + 1]; V[a * 3 + 2] = Va[a * 3 + 0]; }  // descending
  double U[9];
  for (int j = 0; j < 2; j++) {
    double w[3] = {0, 0, 0};
    for (int a = 0; a < 3; a++)
      for (int k = 0; k < 3; k++) w[a] += H[a * 3 + k] * V[k * 3 + j];
    double nn = std::sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
    if (!(nn > 1e-300)) {  // rank deficient: any unit vector orthogonal to the previous column
      if (j == 0) { w[0] = 1; w[1] = 0; w[2] = 0; }
      else {
        const double a0 = std::fabs(U[0]), a1 = std::fabs(U[3]), a2 = std::fabs(U[6]);
        double e[3] = {a0 <= a1 && a0 <= a2 ? 1.0 : 0.0, a1 < a0 && a1 <= a2 ? 1.0 : 0.0, 0.0};
        if (e[0] == 0.0 && e[1] == 0.0) e[2] = 1.0;
        w[0] = U[3] * e[2] - U[6] * e[1]; w[1] = U[6] * e[0] - U[0] * e[2]; w[2] = U[0] * e[1] - U[3] * e[0];
      }
      nn = std::sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
    }
    for (int a = 0; a < 3; a++) U[a * 3 + j] = w[a] / nn;
  }
  // right-handed completions: R = [v0 v1 v0xv1] [u0 u1 u0xu1]^T is the proper rotation V diag(1, 1, det) U^T
  U[2] = U[3] * U[7] - U[6] * U[4]; U[5] = U[6] * U[1] - U[0] * U[7]; U[8] = U[0] * U[4] - U[3] * U[1];
  V[2] = V[3] * V[7] - V[6] * V[4]; V[5] = V[6] * V[1] - V[0] * V[7]; V[8] = V[0] * V[4] - V[3] * V[1];
  for (int a = 0; a < 3; a++)
    for (int b = 0; b < 3; b++) {
      double v = 0;
      for (int k = 0; k < 3; k++) v += V[a * 3 + k] * U[b * 3 + k];
      R[a * 3 + b] = v;
    }
  for (int a = 0; a < 3; a++) t[a] = cq[a] - (R[a * 3] * cp[0] + R[a * 3 + 1] * cp[1] + R[a * 3 + 2] * cp[2]);
}

void rgc_default_icp_params(rgc_icp_params* p) {
  if (!p) return;
  p->max_iterations = 100;                    // :2053
  p->max_correspondence_distance = 10.0;      // poseGraphSearchRadius * 2 with historyKeyframeSearchRadius = 5 (:155, :2052)
  p->transformation_epsilon = 1e-6;           // :2054
  p->euclidean_fitness_epsilon = 1e-6;        // :2055
}

int rgc_icp_align(rgc_ctx* c, const float* source, int ns, const float* target, int nt, int stride_bytes, const rgc_icp_params* prm,
                  float final_T[16], rgc_icp_result* res) {
  if (!c || !source || !target || !prm || !final_T || !res) return RGC_ERR_INVALID;
  if (stride_bytes < 12 || (stride_bytes & 3) || stride_bytes > 4096) return fail(c, RGC_ERR_INVALID, "stride_bytes must be a multiple of 4 and >= 12");
  if (ns < 1 || nt < 1) return fail(c, RGC_ERR_TOO_FEW_POINTS, "ICP needs a non-empty source and target");
  if (ns > (1 << 27) || nt > (1 << 27)) return fail(c, RGC_ERR_INVALID, "cloud larger than 2^27 points");
  if (!(prm->max_correspondence_distance > 0) || prm->max_iterations < 1) return fail(c, RGC_ERR_INVALID, "bad ICP parameters");
  HIPCHK(c, hipSetDevice(c->device));
  hipStream_t s = c->stream;
  memset(res, 0, sizeof(*res));
  int rc;
  // target grid (icp.setInputTarget builds a kd-tree, :2066)
  Cloud& tg = c->aux;
  tg.ready = false;
  {
    const size_t bytes = (size_t)nt * stride_bytes;
    if ((rc = ensure(c, tg.in_copy, bytes))) return rc;
    HIPCHK(c, hipMemcpyAsync(tg.in_copy.p, target, bytes - (stride_bytes - 12), hipMemcpyHostToDevice, s));
    tg.in = (const float*)tg.in_copy.p;
    tg.stride_f = stride_bytes / 4;
    tg.n = nt;
    if ((rc = prepare_map_grid(c, tg, 1.0))) return rc;
  }
  // the source as float4, transformed in place every iteration (pcl::transformPointCloud, fp32)
  const float* d_src;
  if ((rc = stage_in(c, source, ns, stride_bytes, 0, &d_src))) return rc;  // the raw source, kept for the fitness score
  if ((rc = ensure(c, c->pre_out, sizeof(float4) * (size_t)ns))) return rc;
  float4* cur = (float4*)c->pre_out.p;
  const rgck::PoseF I{{1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0}};
  rgck::transform_f32(s, d_src, stride_bytes / 4, ns, I, (float*)cur, 4);  // identity guess: a plain copy to 16-byte points
  const int nb = rgck::linearize_blocks(ns);
  if ((rc = ensure(c, c->partials, sizeof(double) * (rgck::kAccum + 2) * (size_t)(nb > 0 ? nb : 1)))) return rc;
  float fin[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  const double rot_thr = 1.0 - prm->transformation_epsilon, trans_thr = prm->transformation_epsilon;
  double prev_mse = DBL_MAX;
  for (;;) {
    rgck::icp_accumulate(s, cur, ns, (const float4*)tg.P.p, (const int*)tg.start.p, tg.grid, prm->max_correspondence_distance,
                         (double*)c->partials.p, c->d_out);
    HIPCHK(c, hipMemcpyAsync(c->h_out, c->d_out, sizeof(double) * 17, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    HIPCHK(c, hipGetLastError());
    const double* S = c->h_out;
    const double cnt = S[0];
    res->n_correspondences = (int)cnt;
    if (cnt < 3) { res->converged = 0; res->state = RGC_ICP_NO_CORRESPONDENCES; break; }  // min_number_correspondences_
    double R[9], t[3];
    rigid_from_sums(cnt, S + 1, S + 4, S + 7, R, t);
    float T[16] = {(float)R[0], (float)R[1], (float)R[2], (float)t[0], (float)R[3], (float)R[4], (float)R[5], (float)t[1],
                   (float)R[6], (float)R[7], (float)R[8], (float)t[2], 0, 0, 0, 1};
    rgck::transform_f32(s, (const float*)cur, 4, ns, posef_from(T), (float*)cur, 4);
    float nf[16];
    for (int i = 0; i < 4; i++)
      for (int j = 0; j < 4; j++) {
        float v = 0.0f;
        for (int k = 0; k < 4; k++) v += T[i * 4 + k] * fin[k * 4 + j];
        nf[i * 4 + j] = v;
      }
    memcpy(fin, nf, sizeof(fin));
    res->iterations++;
    // DefaultConvergenceCriteria::hasConverged [3P-memory]
    if (res->iterations >= prm->max_iterations) { res->converged = 1; res->state = RGC_ICP_ITERATIONS; break; }
    const double cos_angle = 0.5 * ((double)T[0] + (double)T[5] + (double)T[10] - 1.0);
    const double tr2 = (double)T[3] * T[3] + (double)T[7] * T[7] + (double)T[11] * T[11];
    if (cos_angle >= rot_thr && tr2 <= trans_thr) { res->converged = 1; res->state = RGC_ICP_TRANSFORM; break; }
    const double mse = S[16] / cnt;
    if (std::fabs(mse - prev_mse) < 1e-12) { res->converged = 1; res->state = RGC_ICP_ABS_MSE; break; }
    if (std::fabs(mse - prev_mse) / prev_mse < prm->euclidean_fitness_epsilon) { res->converged = 1; res->state = RGC_ICP_REL_MSE; break; }
    prev_mse = mse;
  }
  // getFitnessScore(): the ORIGINAL source through the final transformation (fp32), mean squared 1-NN distance
  rgck::transform_f32(s, d_src, stride_bytes / 4, ns, posef_from(fin), (float*)cur, 4);
  if ((rc = ensure(c, c->fit_partials, sizeof(double) * (size_t)rgck::fitness_blocks(ns) + 64))) return rc;
  rgck::fitness(s, cur, ns, I, (const float4*)tg.P.p, (const int*)tg.start.p, tg.grid, (double*)c->fit_partials.p, c->d_out);
  HIPCHK(c, hipMemcpyAsync(c->h_out, c->d_out, sizeof(double), hipMemcpyDeviceToHost, s));
  HIPCHK(c, hipStreamSynchronize(s));
  HIPCHK(c, hipGetLastError());
  res->fitness = c->h_out[0] / (double)ns;
  memcpy(final_T, fin, sizeof(fin));
  return RGC_OK;
}

}  // extern "C"
