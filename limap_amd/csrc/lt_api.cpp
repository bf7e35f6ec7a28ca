// lt_api.cpp -- C ABI (include/limap_amd.h) of the MI355X line-triangulation backend:
// context, device buffers, the device pipeline, and the host tail (ComputeLineTracks).
//
// The tail follows global_line_triangulator.cc:168-351 (filterNodeByNumOuterEdges, run_clustering,
// build_tracks_from_clusters), base/graph.cc:57-87,156-165, merging/merging.cc:18-103
// (ComputeLineTrackLabelsGreedy) and merging/aggregator.cc:8-101; it runs on the host in C++
// because it is a serial union-find over a few 10^4..10^6 edges (SURVEY.md 8e "Tail").
// There is no CPU fallback for the kernels: without a GPU lt_create fails.

#include "lt_ctx.h"
#include "lt_tail.h"
#include "lt_rows.h"
#include "lt_pool.h"

#include <algorithm>
#include <atomic>
#include <parallel/algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <dlfcn.h>
#include <map>
#include <mutex>
#include <queue>
#include <set>
#include <string>
#include <unordered_map>
#include <vector>

using namespace lt;

namespace lt {
void launch_fn_query(hipStream_t st, const double *in30, int by_endpoints, double *out32);
// lt_kernels_v2.hip
int gen_slots(long long max_rows);
int gen_groups(long long max_rows);
size_t seg_gate_bytes();
void launch_build_blk(hipStream_t st, int n_blk, const long long *m_off, const int *blk_img, const int *blk_nb,
                      const int *blk_slot, const long long *seg_off, const long long *blk_line_base, void *blkrec);
size_t blk_rec_bytes();
void launch_gen_split(hipStream_t st, int n_blk, long long max_rows, const GenCfg &cfg, const long long *m_off,
                      const int *m_pairs, const int *blk_img, const int *blk_nb, const int *blk_slot,
                      const long long *seg_off, const Cam *cams, const Seg *segs, const PairRec *pairs,
                      const long long *blk_line_base, CRec *st_r, double *st_unc, unsigned *st_key,
                      unsigned *wave_count, unsigned *cnt_bl, int lds_segs, int lds_segs1, void *st_row,
                      unsigned *surv_count, long long n_segs, void *gates, void *blkrec, hipEvent_t *ev3,
                      const double *seg_vp, const unsigned char *seg_has_vp, const long long *seg_pt_off,
                      const void *seg_pts, const double *sfm_xyz, int *err_flag, int many_on, int one_on,
                      int mult);
size_t seg_point_bytes();
void launch_node_prefix(hipStream_t st, long long G, const int *node_img, const long long *seg_off,
                        const long long *nb_off, const long long *blk_line_base, unsigned *cnt_bl,
                        unsigned *base_bl, unsigned *n_tris, long long *tri_off, unsigned long long *status,
                        int *err_flag);
void launch_place(hipStream_t st, int n_blk, long long max_rows, const long long *m_off, const int *blk_img,
                  const long long *seg_off, const long long *blk_line_base, const unsigned *base_bl,
                  const unsigned *wave_count, const long long *tri_off, const CRec *st_r, const double *st_unc,
                  const unsigned *st_key, CRec *cand, double *cand_unc, unsigned *cand_node, int mult, unsigned *perm);
void launch_pack_keys(hipStream_t st, int n_blk, long long max_rows, const long long *m_off,
                      const unsigned *wave_count, const long long *wave_pos, const unsigned *st_key,
                      unsigned *keys_c, unsigned *src_c, int mult);
void launch_permute(hipStream_t st, long long C, const unsigned *skeys, const unsigned *ssrc, const CRec *st_r,
                    const double *st_unc, CRec *cand, double *cand_unc, unsigned *cand_node);
void launch_host_view(hipStream_t st, long long C, const unsigned *perm, const CRec *rec, const double *unc, Cand *out_c,
                      CandLite *out_l);
void launch_cand_node(hipStream_t st, long long G, const long long *tri_off, unsigned *cand_node);
size_t score3_lds_bytes(int max_nb, bool f32);
size_t cand_meta_bytes();
void launch_score3(hipStream_t st, long long C, long long G, const long long *tri_off, const unsigned *cand_node,
                   void *meta, const CRec *cand, const int *node_img, const long long *nb_off,
                   const int *blk_order, const Cam *cams, double *score, unsigned long long *pair_counter,
                   int max_nb, const ScoreCfg &cfg, double scaleinv_guard2, hipEvent_t ev_before, unsigned *draw,
                   bool f32, unsigned *perm, void *rng, bool perm_is_placement, const unsigned *tile_order,
                   unsigned *bucket_cnt, unsigned *bucket_list, unsigned bucket_cap, const unsigned *place,
                   unsigned *rec, const float *st_z, int *err_flag, void *split_pairs, unsigned *split_tile_head,
                   unsigned *split_counters, unsigned split_region_cap, void *split_tile_lohi);
int score3_tile_buckets();
}

// ---- roctx ranges (tracing) around the host-visible stages, so that a `rocprofv3 --marker-trace` timeline
// shows upload / run / download / tail next to the kernels.  The marker library is looked up at run time (no link
// dependency); without it, or with LT_ROCTX=0, the ranges cost one branch.
namespace lt_trace {
typedef int (*push_fn)(const char *);
typedef int (*pop_fn)(void);
static push_fn g_push = nullptr;
static pop_fn g_pop = nullptr;
static void init_once() {
  static bool done = false;
  if (done) return;
  done = true;
  const char *sw = getenv("LT_ROCTX");
  if (sw && sw[0] == '0') return;
  const char *libs[] = {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so"};
  for (const char *l : libs) {
    void *h = dlopen(l, RTLD_LAZY | RTLD_GLOBAL);
    if (!h) continue;
    g_push = (push_fn)dlsym(h, "roctxRangePushA");
    g_pop = (pop_fn)dlsym(h, "roctxRangePop");
    if (g_push && g_pop) return;
    g_push = nullptr;
    g_pop = nullptr;
  }
}
struct Range {
  bool on;
  explicit Range(const char *name) {
    init_once();
    on = g_push != nullptr;
    if (on) g_push(name);
  }
  ~Range() {
    if (on) g_pop();
  }
};
}  // namespace lt_trace
#define LT_CONCAT2(a, b) a##b
#define LT_CONCAT(a, b) LT_CONCAT2(a, b)
#define LT_RANGE(name) lt_trace::Range LT_CONCAT(lt_range_, __LINE__)(name)

// ---- pooled page-locked host blocks (see lt_ctx.h) ----
namespace lt_host {
namespace {
// never destroyed: contexts may be released during process teardown, after static destructors ran
std::mutex &g_pool_mu = *new std::mutex;
std::vector<HostBlock> &g_pool = *new std::vector<HostBlock>;  // released blocks, at most kPoolBlocks / kPoolBytes
constexpr size_t kPoolBlocks = 8;
constexpr size_t kPoolBytes = 4ull << 30;
}  // namespace
HostBlock host_block_acquire(size_t bytes) {
  bytes = std::max<size_t>(bytes, 4096);
  {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    int best = -1;
    for (size_t i = 0; i < g_pool.size(); ++i)
      if (g_pool[i].bytes >= bytes && g_pool[i].bytes <= 8 * bytes + (1u << 20) &&  // a 4 KB scratch must not eat the staging block
          (best < 0 || g_pool[i].bytes < g_pool[best].bytes))
        best = (int)i;
    if (best >= 0) {
      HostBlock b = g_pool[best];
      g_pool.erase(g_pool.begin() + best);
      return b;
    }
  }
  HostBlock b;
  // size classes (powers of two up to 64 MB) so that the blocks a context needs -- per-node results, the
  // download buffer, scratch -- are interchangeable between contexts; pinning a new block costs ~1 ms per MB
  if (bytes <= (64u << 20)) {
    size_t c = 4096;
    while (c < bytes) c <<= 1;
    b.bytes = c;
  } else {
    b.bytes = bytes + bytes / 8;
  }
  void *q = nullptr;
  if (hipHostMalloc(&q, b.bytes, hipHostMallocDefault) == hipSuccess) {
    b.p = q;
    b.pinned = true;
  } else {
    (void)hipGetLastError();
    b.p = std::malloc(b.bytes);
    b.pinned = false;
    if (!b.p) b.bytes = 0;
  }
  return b;
}
void host_block_release(HostBlock b) {
  if (!b.p) return;
  {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    size_t total = b.bytes;
    for (auto &x : g_pool) total += x.bytes;
    if (b.pinned && g_pool.size() < kPoolBlocks && total <= kPoolBytes) {
      g_pool.push_back(b);
      return;
    }
  }
  if (b.pinned) (void)hipHostFree(b.p);
  else std::free(b.p);
}
// ---- cached device blocks (see lt_ctx.h) ----
namespace {
struct DevBlock {
  void *p;
  size_t cap;
  int device;
};
std::vector<DevBlock> &g_dev_pool = *new std::vector<DevBlock>;
constexpr size_t kDevPoolBlocks = 512;
constexpr size_t kDevPoolBytes = 64ull << 30;  // of 288 GB HBM
}  // namespace
void *dev_block_acquire(size_t bytes, size_t *cap) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    int best = -1;
    for (size_t i = 0; i < g_dev_pool.size(); ++i) {
      const DevBlock &b = g_dev_pool[i];
      if (b.device != dev || b.cap < bytes || b.cap > 4 * bytes + (1u << 20)) continue;
      if (best < 0 || b.cap < g_dev_pool[best].cap) best = (int)i;
    }
    if (best >= 0) {
      DevBlock b = g_dev_pool[best];
      g_dev_pool.erase(g_dev_pool.begin() + best);
      *cap = b.cap;
      return b.p;
    }
  }
  void *p = nullptr;
  if (hipMalloc(&p, bytes) != hipSuccess) {
    (void)hipGetLastError();
    // out of device memory: give the cache back and retry once
    std::vector<DevBlock> drop;
    {
      std::lock_guard<std::mutex> lk(g_pool_mu);
      drop.swap(g_dev_pool);
    }
    for (auto &b : drop) (void)hipFree(b.p);
    if (hipMalloc(&p, bytes) != hipSuccess) {
      (void)hipGetLastError();
      *cap = 0;
      return nullptr;
    }
  }
  *cap = bytes;
  return p;
}
void dev_block_release(void *p, size_t cap) {
  if (!p) return;
  int dev = 0;
  (void)hipGetDevice(&dev);
  {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    size_t total = cap;
    for (auto &b : g_dev_pool) total += b.cap;
    if (g_dev_pool.size() < kDevPoolBlocks && total <= kDevPoolBytes) {
      g_dev_pool.push_back(DevBlock{p, cap, dev});
      return;
    }
  }
  (void)hipFree(p);
}
// streams + timing events of destroyed contexts (creating them costs ~2 ms per context)
struct StreamSet {
  int device;
  hipStream_t stream;
  hipEvent_t ev[14];
};
std::vector<StreamSet> &g_stream_pool = *new std::vector<StreamSet>;
bool stream_set_acquire(int device, hipStream_t *stream, hipEvent_t ev[14]) {
  std::lock_guard<std::mutex> lk(g_pool_mu);
  for (size_t i = 0; i < g_stream_pool.size(); ++i)
    if (g_stream_pool[i].device == device) {
      *stream = g_stream_pool[i].stream;
      for (int k = 0; k < 14; ++k) ev[k] = g_stream_pool[i].ev[k];
      g_stream_pool.erase(g_stream_pool.begin() + i);
      return true;
    }
  return false;
}
bool stream_set_release(int device, hipStream_t stream, hipEvent_t ev[14]) {
  std::lock_guard<std::mutex> lk(g_pool_mu);
  if (g_stream_pool.size() >= 8) return false;
  StreamSet s;
  s.device = device;
  s.stream = stream;
  for (int k = 0; k < 14; ++k) s.ev[k] = ev[k];
  g_stream_pool.push_back(s);
  return true;
}
void release_cached_memory() {
  std::vector<DevBlock> drop;
  std::vector<HostBlock> hdrop;
  {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    drop.swap(g_dev_pool);
    hdrop.swap(g_pool);
  }
  for (auto &b : drop) (void)hipFree(b.p);
  for (auto &b : hdrop) {
    if (b.pinned) (void)hipHostFree(b.p);
    else std::free(b.p);
  }
}
}  // namespace lt_host

extern "C" void lt_release_cached_memory(void) { lt_host::release_cached_memory(); }

namespace {

// per-kernel HIP events cost a few microseconds of stream bubble each.  LT_FINE_TIMERS (read per run): unset / 1 =
// the event in front of k_score3 only (timer [15]: bench.py prices the dominant kernel with it, inside its timed
// region), 2 = also the events around k_gates and k_tri_rows (timers [13], [14]: +5 us per step), 0 = none
int fine_level() {
  const char *e = getenv("LT_FINE_TIMERS");
  if (!e) return 1;
  return e[0] == '0' ? 0 : (e[0] == '2' ? 2 : 1);
}
bool fine_timers() { return fine_level() >= 1; }
bool fine_gen_timers() { return fine_level() >= 2; }

double now_ms() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

}  // namespace

namespace {

double multiplier(double score_th) { return 1.0 / std::sqrt(-std::log(score_th) * 2.0); }  // line_linker.cc:9-12

LinkCfg2 make_l2(const lt_config &c) {
  LinkCfg2 l;
  l.score_th = c.l2_score_th; l.th_angle = c.l2_th_angle; l.th_overlap = c.l2_th_overlap;
  l.th_smartoverlap = c.l2_th_smartoverlap; l.th_smartangle = c.l2_th_smartangle;
  l.th_perp = c.l2_th_perp; l.th_innerseg = c.l2_th_innerseg;
  l.mult = multiplier(c.l2_score_th);
  l.use_angle = c.l2_use_angle; l.use_overlap = c.l2_use_overlap; l.use_smartangle = c.l2_use_smartangle;
  l.use_perp = c.l2_use_perp; l.use_innerseg = c.l2_use_innerseg; l.pad_ = 0;
  return l;
}
LinkCfg3 make_l3(const lt_config &c) {
  LinkCfg3 l;
  l.score_th = c.l3_score_th; l.th_angle = c.l3_th_angle; l.th_overlap = c.l3_th_overlap;
  l.th_smartoverlap = c.l3_th_smartoverlap; l.th_smartangle = c.l3_th_smartangle;
  l.th_perp = c.l3_th_perp; l.th_innerseg = c.l3_th_innerseg; l.th_scaleinv = c.l3_th_scaleinv;
  l.mult = multiplier(c.l3_score_th);
  l.use_angle = c.l3_use_angle; l.use_overlap = c.l3_use_overlap; l.use_smartangle = c.l3_use_smartangle;
  l.use_perp = c.l3_use_perp; l.use_innerseg = c.l3_use_innerseg; l.use_scaleinv = c.l3_use_scaleinv;
  return l;
}

GenCfg make_gen(const lt_ctx *ctx) {
  const lt_config &c = ctx->cfg;
  GenCfg g;
  g.min_length_2d = c.min_length_2d; g.angle_th = c.line_tri_angle_threshold; g.iou_th = c.IoU_threshold;
  g.sens_th = c.sensitivity_threshold; g.var2d = c.var2d;
  for (int k = 0; k < 3; ++k) { g.lo[k] = ctx->lo[k]; g.hi[k] = ctx->hi[k]; }
  g.use_ranges = ctx->ranges_on; g.use_endpoints = c.use_endpoints_triangulation;
  g.disable_algebraic = c.disable_algebraic_triangulation;
  // LT_TEST_NO_FAST_GATES: the cheap gates never decide, the reference's exact gates do all the work;
  // results must not change (tests/test_gpu_guards.py)
  g.force_undecided = getenv("LT_TEST_NO_FAST_GATES") != nullptr;
  // VP-guided proposals do not depend on the algebraic gates: every row must reach the triangulation kernel
  if (c.use_vp && !c.disable_vp_triangulation) g.force_undecided = 1;
  // The gate `90 - acos(a)*180/pi < th` is equivalent to a < sin(th) up to libm rounding; outside
  // a +-1e-7 relative band around sin(th) the comparison of a alone decides, inside it the exact
  // expression is evaluated.  For thresholds outside (0, 90) the band covers everything.
  double th = c.line_tri_angle_threshold;
  if (th > 1e-3 && th < 89.0) {
    double s = std::sin(th * kPi / 180.0);
    g.sin_lo = s * (1.0 - 1e-7);
    g.sin_hi = s * (1.0 + 1e-7);
  } else {
    g.sin_lo = -1.0;
    g.sin_hi = 1e300;
  }
  // sensitivity = 90 - acos(c) 180/pi > th  <=>  c > sin(th) (acos is decreasing); same +-1e-7 band as above
  {
    const double ths = c.sensitivity_threshold;
    if (ths > 1e-3 && ths < 89.0) {
      const double sn = std::sin(ths * kPi / 180.0);
      g.sens_lo = sn * (1.0 - 1e-7);
      g.sens_hi = sn * (1.0 + 1e-7);
    } else {
      g.sens_lo = -1.0;   // the band covers everything: always the exact expression
      g.sens_hi = 1e300;
    }
  }
  // `length <= min_length` skips the connection (base_line_triangulator.cc:166,177); length = sqrt(q)
  const double L = c.min_length_2d;
  if (L > 0.0) {
    g.len_lo2 = L * L * (1.0 - 1e-12);
    g.len_hi2 = L * L * (1.0 + 1e-12);
  } else if (L == 0.0) {
    g.len_lo2 = g.len_hi2 = 0.0;  // only q == 0 has length <= 0
  } else {
    g.len_lo2 = g.len_hi2 = -1.0;  // nothing has a negative length
  }
  return g;
}

ScoreCfg make_score(const lt_ctx *ctx) {
  ScoreCfg s;
  s.l2 = make_l2(ctx->cfg);
  s.l3 = make_l3(ctx->cfg);
  // set_to_shared_parent_scoring, line_linker.h:115-121
  s.l3.use_angle = 1; s.l3.use_overlap = 0; s.l3.use_perp = 0; s.l3.use_innerseg = 0; s.l3.use_scaleinv = 1;
  // 3D angle gate: score_angle >= score_th  <=>  angle <= th_angle (up to rounding).  Pairs whose
  // |cos| is below cos(th_angle * (1 + 1e-6) + 1e-6 deg) can never pass; all others are evaluated
  // exactly.  th_angle >= 90 disables the early exit.
  double th = s.l3.th_angle * (1.0 + 1e-6) + 1e-6;
  s.cos_guard = (th < 90.0) ? std::cos(th * kPi / 180.0) : -1.0;
  // LT_TEST_NO_SCORE_GUARDS: no conservative early exit in the scoring sweep (every pair of a node is
  // evaluated densely); results must not change (tests/test_gpu_guards.py)
  if (getenv("LT_TEST_NO_SCORE_GUARDS")) s.cos_guard = -1.0;
  s.fullscore_th = ctx->cfg.fullscore_th;
  s.max_valid_conns = ctx->cfg.max_valid_conns;
  s.pad_ = 0;
  return s;
}

int bits_for(long long n) {
  int b = 1;
  while (b < 32 && (1ll << b) < n) ++b;
  return b;
}

// ---------------------------------------------------------------------------------------------
int init_common(lt_ctx *ctx, int n_img, const int32_t *img_ids, const int64_t *seg_off_in,
                const std::vector<int> &perm) {
  // perm: sorted position -> caller position
  ctx->n_img = n_img;
  ctx->img_ids.resize(n_img);
  ctx->id2idx.clear();
  ctx->seg_off.assign(n_img + 1, 0);
  for (int i = 0; i < n_img; ++i) {
    ctx->img_ids[i] = img_ids[perm[i]];
    if (ctx->id2idx.count(ctx->img_ids[i])) return fail(ctx, LT_ERR_ARGUMENT, "duplicate image id in Init");
    ctx->id2idx[ctx->img_ids[i]] = i;
    long long m = seg_off_in[perm[i] + 1] - seg_off_in[perm[i]];
    if (m < 0) return fail(ctx, LT_ERR_ARGUMENT, "seg_off must be non-decreasing");
    if (m > 65535) return fail(ctx, LT_ERR_ARGUMENT, "more than 65535 lines in one image (uint16 line ids, util/types.h:16)");
    ctx->seg_off[i + 1] = ctx->seg_off[i] + m;
  }
  ctx->G = ctx->seg_off[n_img];
  if (ctx->G >= (1ll << 32) - 1) return fail(ctx, LT_ERR_ARGUMENT, "too many nodes (>= 2^32-1)");
  ctx->h_node_img.resize(ctx->G);
  for (int i = 0; i < n_img; ++i)
    for (long long g = ctx->seg_off[i]; g < ctx->seg_off[i + 1]; ++g) ctx->h_node_img[g] = i;
  ctx->triangulated.assign(n_img, 0);
  ctx->neighbors.assign(n_img, {});
  lt_host::host_block_release(ctx->best_c_blk);
  ctx->best_c_blk = lt_host::host_block_acquire(sizeof(Cand) * (size_t)std::max<long long>(ctx->G, 1));
  ctx->best_c = (Cand *)ctx->best_c_blk.p;
  if (!ctx->best_c) return fail(ctx, LT_ERR_RUNTIME, "out of host memory for the per-node results");
  ctx->best_c_set.assign(n_img, 0);
  ctx->best_score.assign(ctx->G, 0.0);
  ctx->best_src2.assign(2 * ctx->G, 0);
  ctx->n_tris.assign(ctx->G, 0);
  ctx->has_best.assign(ctx->G, 0);
  ctx->valid_edges.reset(ctx->G);
  ctx->dbg_pool.clear(); ctx->dbg_off.assign((size_t)ctx->G, 0); ctx->dbg_cnt.assign((size_t)ctx->G, 0);
  ctx->vp_ready = false;
  ctx->pts_ready = false; ctx->sfm_given = false; ctx->pts_dirty = false;
  ctx->h_seg_pts.clear(); ctx->h_seg_pt_off.clear(); ctx->h_sfm_ids.clear(); ctx->h_sfm_xyz.clear();
  ctx->tracks.clear();
  ctx->tracks_done = false;
  ctx->job_mode = 0;
  ctx->job_imgs.clear(); ctx->job_nbs.clear(); ctx->job_order.clear();
  ctx->h_m_off.assign(1, 0); ctx->h_m_pairs.clear(); ctx->streamed_ints = 0;
  ctx->rows_sorted = true;
  ctx->uploaded = ctx->ran = ctx->downloaded = false;
  return LT_OK;
}

// hk / hq / ht / hs: the scene in host memory (ascending id order) if the caller has it, else it is read back
int build_invariants(lt_ctx *ctx, const double *hk = nullptr, const double *hq = nullptr, const double *ht = nullptr,
                     const double *hs = nullptr) {
  hipStream_t st = ctx->stream;
  ENSURE(ctx, ctx->d_cams, sizeof(Cam) * (size_t)std::max(ctx->n_img, 1));
  ENSURE(ctx, ctx->d_segs, sizeof(Seg) * (size_t)std::max<long long>(ctx->G, 1));
  ENSURE(ctx, ctx->d_seg_off, sizeof(long long) * (size_t)(ctx->n_img + 1));
  ENSURE(ctx, ctx->d_node_img, sizeof(int) * (size_t)std::max<long long>(ctx->G, 1));
  HIPCHK(ctx, hipMemcpyAsync(ctx->d_seg_off.p, ctx->seg_off.data(), sizeof(long long) * (ctx->n_img + 1),
                             hipMemcpyHostToDevice, st));
  if (ctx->G > 0)
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_node_img.p, ctx->h_node_img.data(), sizeof(int) * ctx->G,
                               hipMemcpyHostToDevice, st));
  launch_build_cams(st, ctx->n_img, ctx->d_kvec.as<double>(), ctx->d_qvec.as<double>(), ctx->d_tvec.as<double>(),
                    ctx->d_cams.as<Cam>());
  // the per-segment gate records of stage A (k_gates) are written by the same kernel as the segment records
  ENSURE(ctx, ctx->d_seg_gates, seg_gate_bytes() * (size_t)std::max<long long>(ctx->G, 1));
  launch_build_segs(st, ctx->G, ctx->n_img, ctx->d_seg_off.as<long long>(), ctx->d_segs_raw.as<double>(),
                    ctx->cfg.add_halfpix ? 0.5 : 0.0, ctx->d_cams.as<Cam>(), ctx->d_segs.as<Seg>(), ctx->d_seg_gates.p);
  HIPCHK(ctx, hipGetLastError());
  HIPCHK(ctx, hipStreamSynchronize(st));
  {  // host copies for the tail-side filters (small: 88 B per image + 32 B per segment)
    const int n = ctx->n_img;
    std::vector<double> k(4 * (size_t)n), q(4 * (size_t)n), t(3 * (size_t)n);
    if (hk && hq && ht && hs) {
      std::memcpy(k.data(), hk, 32 * (size_t)n);
      std::memcpy(q.data(), hq, 32 * (size_t)n);
      std::memcpy(t.data(), ht, 24 * (size_t)n);
      ctx->h_segs.assign(hs, hs + 4 * (size_t)ctx->G);
    } else {
      ctx->h_segs.assign(4 * (size_t)ctx->G, 0.0);
      if (n > 0) {
        HIPCHK(ctx, hipMemcpy(k.data(), ctx->d_kvec.p, 32 * (size_t)n, hipMemcpyDeviceToHost));
        HIPCHK(ctx, hipMemcpy(q.data(), ctx->d_qvec.p, 32 * (size_t)n, hipMemcpyDeviceToHost));
        HIPCHK(ctx, hipMemcpy(t.data(), ctx->d_tvec.p, 24 * (size_t)n, hipMemcpyDeviceToHost));
      }
      if (ctx->G > 0)
        HIPCHK(ctx, hipMemcpy(ctx->h_segs.data(), ctx->d_segs_raw.p, 32 * (size_t)ctx->G, hipMemcpyDeviceToHost));
    }
    if (ctx->cfg.add_halfpix)
      for (double &v : ctx->h_segs) v = v + 0.5;
    ctx->h_cams.resize(n);
    for (int i = 0; i < n; ++i) cam_build(&k[4 * i], &q[4 * i], &t[3 * i], &ctx->h_cams[i]);
  }
  ctx->inited = true;
  return LT_OK;
}

// Assemble the per-job neighbour tables from the buffered calls.
void build_job_tables(lt_ctx *ctx) {
  const int n_img = ctx->n_img;
  ctx->h_nb_off.assign(n_img + 1, 0);
  std::vector<int> job_pos(n_img, -1);
  for (size_t j = 0; j < ctx->job_imgs.size(); ++j) job_pos[ctx->job_imgs[j]] = (int)j;
  ctx->h_blk_img.clear(); ctx->h_blk_nb.clear(); ctx->h_blk_slot.clear(); ctx->h_blk_order.clear();
  ctx->max_nb = 1;
  for (int i = 0; i < n_img; ++i) {
    ctx->h_nb_off[i] = (long long)ctx->h_blk_img.size();
    int j = job_pos[i];
    if (j < 0) continue;
    const auto &nbs = ctx->job_nbs[j];
    ctx->max_nb = std::max(ctx->max_nb, (int)nbs.size());
    for (size_t k = 0; k < nbs.size(); ++k) {
      ctx->h_blk_img.push_back(i);
      ctx->h_blk_nb.push_back(nbs[k]);
      ctx->h_blk_slot.push_back((int)k);
      ctx->h_blk_order.push_back(ctx->job_order[j][k]);
    }
  }
  ctx->h_nb_off[n_img] = (long long)ctx->h_blk_img.size();
  ctx->n_blk = (int)ctx->h_blk_img.size();
  ctx->h_blk_line_base.assign(ctx->n_blk + 1, 0);
  ctx->max_nb_segs = 0;
  ctx->max_own_segs = 0;
  for (int b = 0; b < ctx->n_blk; ++b) {
    int i1 = ctx->h_blk_img[b];
    int i2 = ctx->h_blk_nb[b];
    ctx->max_nb_segs = std::max(ctx->max_nb_segs, (int)(ctx->seg_off[i2 + 1] - ctx->seg_off[i2]));
    ctx->max_own_segs = std::max(ctx->max_own_segs, (int)(ctx->seg_off[i1 + 1] - ctx->seg_off[i1]));
    ctx->h_blk_line_base[b + 1] = ctx->h_blk_line_base[b] + (ctx->seg_off[i1 + 1] - ctx->seg_off[i1]) + 1;
  }
}

template <class T>
int upload_vec(lt_ctx *ctx, DevBuf &buf, const std::vector<T> &v) {
  ENSURE(ctx, buf, sizeof(T) * std::max<size_t>(v.size(), 1));
  if (!v.empty())
    HIPCHK(ctx, hipMemcpyAsync(buf.p, v.data(), sizeof(T) * v.size(), hipMemcpyHostToDevice, ctx->stream));
  return LT_OK;
}

}  // namespace

// =============================================================================================
extern "C" {

void lt_config_default(lt_config *c) {
  std::memset(c, 0, sizeof(*c));
  c->min_length_2d = 20.0; c->line_tri_angle_threshold = 5.0; c->IoU_threshold = 0.1;
  c->sensitivity_threshold = 70.0; c->var2d = 2.0;
  c->fullscore_th = 1.0; c->max_valid_conns = 1000; c->min_num_outer_edges = 1;
  c->merging_strategy = 0; c->num_outliers_aggregator = 2;
  c->l2_score_th = 0.5; c->l2_th_angle = 8.0; c->l2_th_overlap = 0.1; c->l2_th_smartoverlap = 0.2;
  c->l2_th_smartangle = 1.0; c->l2_th_perp = 5.0; c->l2_th_innerseg = 5.0;
  c->l2_use_angle = 1; c->l2_use_overlap = 1; c->l2_use_smartangle = 1; c->l2_use_perp = 1; c->l2_use_innerseg = 0;
  c->l3_score_th = 0.5; c->l3_th_angle = 10.0; c->l3_th_overlap = 0.01; c->l3_th_smartoverlap = 0.1;
  c->l3_th_smartangle = 1.0; c->l3_th_perp = 0.02; c->l3_th_innerseg = 0.02; c->l3_th_scaleinv = 0.01;
  c->l3_use_angle = 1; c->l3_use_overlap = 1; c->l3_use_smartangle = 1; c->l3_use_perp = 0;
  c->l3_use_innerseg = 1; c->l3_use_scaleinv = 0;
}

int lt_abi_version(void) { return 1; }
uint64_t lt_sizeof_config(void) { return (uint64_t)sizeof(lt_config); }

lt_ctx *lt_create(const lt_config *cfg, int device) {
  int n_dev = 0;
  hipError_t e = hipGetDeviceCount(&n_dev);
  if (e != hipSuccess || n_dev <= 0) {
    std::fprintf(stderr, "limap_amd: no HIP device available (%s); this backend has no CPU fallback\n",
                 hipGetErrorString(e));
    return nullptr;
  }
  if (device < 0 || device >= n_dev) {
    std::fprintf(stderr, "limap_amd: device %d out of range (%d devices)\n", device, n_dev);
    return nullptr;
  }
  if (hipSetDevice(device) != hipSuccess) return nullptr;
  lt_ctx *ctx = new lt_ctx();
  ctx->cfg = *cfg;
  ctx->device = device;
  if (!lt_host::stream_set_acquire(device, &ctx->stream, ctx->ev)) {
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
      delete ctx;
      return nullptr;
    }
    for (auto &ev : ctx->ev) (void)hipEventCreate(&ev);
  }
  ctx->pool_stream = ctx->stream;
  ctx->h_pinned_blk = lt_host::host_block_acquire(4096);
  ctx->h_pinned = ctx->h_pinned_blk.pinned ? (long long *)ctx->h_pinned_blk.p : nullptr;
  return ctx;
}

int finish_run(lt_ctx *ctx);
#define LT_FINISH(ctx)              \
  do {                              \
    int rc_fin_ = finish_run(ctx);  \
    if (rc_fin_) return rc_fin_;    \
  } while (0)

void lt_destroy(lt_ctx *ctx) {
  if (!ctx) return;
  (void)finish_run(ctx);
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  DevBuf *bufs[] = {&ctx->d_kvec, &ctx->d_qvec, &ctx->d_tvec, &ctx->d_segs_raw, &ctx->d_cams, &ctx->d_segs,
                    &ctx->d_seg_off, &ctx->d_node_img, &ctx->d_nb_off, &ctx->d_blk_img, &ctx->d_blk_nb,
                    &ctx->d_blk_slot, &ctx->d_blk_order, &ctx->d_m_off, &ctx->d_m_pairs, &ctx->d_pairs,
                    &ctx->d_keys, &ctx->d_rows, &ctx->d_row_blk, &ctx->d_skeys, &ctx->d_srows, &ctx->d_sort_tmp,
                    &ctx->d_conn_off, &ctx->d_st_c, &ctx->d_st_l, &ctx->d_flags, &ctx->d_pos, &ctx->d_scan_tmp,
                    &ctx->d_item_off, &ctx->d_masks, &ctx->d_mask_cnt, &ctx->d_mask_pos, &ctx->d_cand, &ctx->d_hcand, &ctx->d_hlite,
                    &ctx->d_split_pairs, &ctx->d_split_segs, &ctx->d_split_head, &ctx->d_split_tot,
                    &ctx->d_rm_line, &ctx->d_rm_act, &ctx->d_rm_edges, &ctx->d_rm_cnt,
                    &ctx->d_lite, &ctx->d_tri_off, &ctx->d_score, &ctx->d_best_idx, &ctx->d_edge_flag,
                    &ctx->d_nvalid, &ctx->d_edge_off, &ctx->d_edges, &ctx->d_best_c, &ctx->d_best_score,
                    &ctx->d_best_src, &ctx->d_ntris, &ctx->d_err, &ctx->d_blk_line_base, &ctx->d_cnt_bl,
                    &ctx->d_st_key, &ctx->d_wave_count, &ctx->d_wave_pos, &ctx->d_ntris_u, &ctx->d_cand_node,
                    &ctx->d_pair_counter, &ctx->d_result3, &ctx->d_tile_order, &ctx->d_scan_status, &ctx->d_perm, &ctx->d_rng, &ctx->d_chunks, &ctx->d_cand_meta, &ctx->d_st_row, &ctx->d_surv_count, &ctx->d_seg_gates, &ctx->d_blkrec, &ctx->d_seg_vp, &ctx->d_seg_has_vp, &ctx->d_base_bl, &ctx->d_blk_chunk_off, &ctx->d_needed, &ctx->d_seg_pts, &ctx->d_seg_pt_off, &ctx->d_sfm_xyz,
                    &ctx->d_place_perm, &ctx->d_ex_rec, &ctx->d_ex_ent, &ctx->d_ex_z, &ctx->d_tile_list, &ctx->d_exp_tile_order, &ctx->d_tail_keys, &ctx->d_tail_skeys, &ctx->d_tail_sims, &ctx->d_tail_mark,
                    &ctx->d_tail_pos, &ctx->d_tail_recs, &ctx->d_tail_nodes, &ctx->d_tail_tmp, &ctx->d_tail_keep,
                    &ctx->d_tail_kpos};
  lt_host::host_block_release(ctx->h_pinned_blk);
  lt_host::host_block_release(ctx->best_c_blk);
  for (DevBuf *b : bufs) b->release();
  for (auto &e : ctx->ev_b)
    if (e) (void)hipEventDestroy(e);
  // a context that still owns its stream hands stream + events to the next context
  if (ctx->pool_stream) (void)hipStreamSynchronize(ctx->pool_stream);
  if (!(ctx->pool_stream && lt_host::stream_set_release(ctx->device, ctx->pool_stream, ctx->ev))) {
    for (auto &ev : ctx->ev)
      if (ev) (void)hipEventDestroy(ev);
    if (ctx->pool_stream) (void)hipStreamDestroy(ctx->pool_stream);
  }
  delete ctx;
}

const char *lt_last_error(lt_ctx *ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int lt_set_stream(lt_ctx *ctx, void *hip_stream) {
  LT_FINISH(ctx);
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));  // nothing of this context stays in flight on the old stream
  if (hip_stream) {
    ctx->stream = reinterpret_cast<hipStream_t>(hip_stream);
    ctx->own_stream = false;
  } else {
    if (!ctx->pool_stream) HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->pool_stream, hipStreamNonBlocking));
    ctx->stream = ctx->pool_stream;
    ctx->own_stream = true;
  }
  return LT_OK;
}

int lt_set_ranges(lt_ctx *ctx, const double lo[3], const double hi[3]) {
  ctx->ranges_on = true;
  for (int k = 0; k < 3; ++k) { ctx->lo[k] = lo[k]; ctx->hi[k] = hi[k]; }
  return LT_OK;
}
int lt_unset_ranges(lt_ctx *ctx) {
  ctx->ranges_on = false;
  return LT_OK;
}

int lt_init(lt_ctx *ctx, int n_img, const int32_t *img_ids, const double *kvec, const double *qvec,
            const double *tvec, const int64_t *seg_off, const double *segs) {
  LT_FINISH(ctx);
  HIPCHK(ctx, hipSetDevice(ctx->device));
  if (n_img < 0) return fail(ctx, LT_ERR_ARGUMENT, "n_img < 0");
  // the TriangulateImage calls follow Init: start waking the team of the row pass now (lt_pool.h; returns at once)
  lt_host::SpinPool::get(lt_host::row_workers()).wake();
  std::vector<int> perm(n_img);
  for (int i = 0; i < n_img; ++i) perm[i] = i;
  std::stable_sort(perm.begin(), perm.end(), [&](int a, int b) { return img_ids[a] < img_ids[b]; });
  static const bool init_trace = getenv("LT_TAIL_TRACE") != nullptr;
  double tl = now_ms();
  auto lap = [&](const char *what) {
    if (!init_trace) return;
    double t = now_ms();
    std::fprintf(stderr, "[init] %-18s %.3f ms\n", what, t - tl);
    tl = t;
  };
  int rc = init_common(ctx, n_img, img_ids, seg_off, perm);
  if (rc) return rc;
  lap("init_common");
  // gather into ascending-id order
  std::vector<double> k(4 * (size_t)n_img), q(4 * (size_t)n_img), t(3 * (size_t)n_img), s(4 * (size_t)ctx->G);
  for (int i = 0; i < n_img; ++i) {
    int p = perm[i];
    std::memcpy(&k[4 * i], kvec + 4 * p, 32);
    std::memcpy(&q[4 * i], qvec + 4 * p, 32);
    std::memcpy(&t[3 * i], tvec + 3 * p, 24);
    long long m = seg_off[p + 1] - seg_off[p];
    if (m > 0) std::memcpy(&s[4 * ctx->seg_off[i]], segs + 4 * seg_off[p], (size_t)m * 32);
  }
  lap("gather");
  if ((rc = upload_vec(ctx, ctx->d_kvec, k))) return rc;
  if ((rc = upload_vec(ctx, ctx->d_qvec, q))) return rc;
  if ((rc = upload_vec(ctx, ctx->d_tvec, t))) return rc;
  if ((rc = upload_vec(ctx, ctx->d_segs_raw, s))) return rc;
  lap("upload");
  rc = build_invariants(ctx, k.data(), q.data(), t.data(), s.data());
  lap("build_invariants");
  return rc;
}

int lt_init_vp(lt_ctx *ctx, int n_img, const int32_t *img_ids, const int64_t *label_off, const int32_t *labels,
               const int64_t *vp_off, const double *vps) {
  LT_FINISH(ctx);
  if (!ctx->inited) return fail(ctx, LT_ERR_STATE, "InitVPResults before Init");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  std::vector<double> vp(3 * (size_t)std::max<long long>(ctx->G, 1), 0.0);
  std::vector<unsigned char> has((size_t)std::max<long long>(ctx->G, 1), 0);
  for (int i = 0; i < n_img; ++i) {
    auto it = ctx->id2idx.find(img_ids[i]);
    if (it == ctx->id2idx.end()) return fail(ctx, LT_ERR_ARGUMENT, "InitVPResults: unknown image id " + std::to_string(img_ids[i]));
    const int idx = it->second;
    const long long M = ctx->seg_off[idx + 1] - ctx->seg_off[idx];
    const long long nl = label_off[i + 1] - label_off[i], nv = vp_off[i + 1] - vp_off[i];
    if (nl != M) return fail(ctx, LT_ERR_ARGUMENT, "InitVPResults: " + std::to_string(nl) + " labels for image " +
                                                       std::to_string(img_ids[i]) + " with " + std::to_string(M) + " lines");
    for (long long l = 0; l < M; ++l) {
      const int lab = labels[label_off[i] + l];
      if (lab < 0) continue;  // VPResult::HasVP (vplib/vpbase.h:42)
      if (lab >= nv) return fail(ctx, LT_ERR_ARGUMENT, "InitVPResults: VP label out of range");
      const long long g = ctx->seg_off[idx] + l;
      has[(size_t)g] = 1;
      for (int k = 0; k < 3; ++k) vp[3 * (size_t)g + k] = vps[3 * (vp_off[i] + lab) + k];
    }
  }
  int rc;
  if ((rc = upload_vec(ctx, ctx->d_seg_vp, vp))) return rc;
  if ((rc = upload_vec(ctx, ctx->d_seg_has_vp, has))) return rc;
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  ctx->vp_ready = true;
  return LT_OK;
}

int lt_set_bipartites(lt_ctx *ctx, int n_img, const int32_t *img_ids, const int64_t *pt_off, const int32_t *pt_ids,
                      const double *pt_xy, const int32_t *pt_p3d, const int64_t *line_off, const int64_t *lp_off,
                      const int32_t *lp_ptids) {
  LT_FINISH(ctx);
  if (!ctx->inited) return fail(ctx, LT_ERR_STATE, "SetBipartites2d before Init");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  struct SegPointH { int p3d_id, sfm; double x, y; };
  static_assert(sizeof(SegPointH) == 24, "SegPoint layout");
  if (seg_point_bytes() != sizeof(SegPointH)) return fail(ctx, LT_ERR_RUNTIME, "SegPoint layout mismatch");
  std::vector<std::vector<SegPointH>> per_seg((size_t)std::max<long long>(ctx->G, 1));
  for (int i = 0; i < n_img; ++i) {
    auto it = ctx->id2idx.find(img_ids[i]);
    if (it == ctx->id2idx.end()) return fail(ctx, LT_ERR_ARGUMENT, "SetBipartites2d: unknown image id " + std::to_string(img_ids[i]));
    const int idx = it->second;
    const long long M = ctx->seg_off[idx + 1] - ctx->seg_off[idx];
    if (line_off[i + 1] - line_off[i] != M)
      return fail(ctx, LT_ERR_ARGUMENT, "SetBipartites2d: image " + std::to_string(img_ids[i]) + " has " + std::to_string(M) +
                                            " lines, the bipartite lists " + std::to_string(line_off[i + 1] - line_off[i]));
    // point id -> row of this image's point arrays
    std::unordered_map<int, long long> row;
    for (long long k = pt_off[i]; k < pt_off[i + 1]; ++k) row[pt_ids[k]] = k;
    for (long long l = 0; l < M; ++l) {
      const long long L = line_off[i] + l;
      // neighbor_points(): ascending point id (std::set); the reference then keys by point3D_id, first wins
      std::vector<int> ids(lp_ptids + lp_off[L], lp_ptids + lp_off[L + 1]);
      std::sort(ids.begin(), ids.end());
      ids.erase(std::unique(ids.begin(), ids.end()), ids.end());
      std::map<int, SegPointH> by3d;
      for (int pid : ids) {
        auto r = row.find(pid);
        if (r == row.end()) return fail(ctx, LT_ERR_ARGUMENT, "SetBipartites2d: a line refers to an unknown point id");
        SegPointH sp{pt_p3d[r->second], -1, pt_xy[2 * r->second], pt_xy[2 * r->second + 1]};
        by3d.insert({sp.p3d_id, sp});
      }
      auto &dst = per_seg[(size_t)(ctx->seg_off[idx] + l)];
      dst.clear();
      for (auto &kv : by3d) dst.push_back(kv.second);
    }
  }
  ctx->h_seg_pt_off.assign((size_t)ctx->G + 1, 0);
  ctx->h_seg_pts.clear();
  for (long long g = 0; g < ctx->G; ++g) {
    ctx->h_seg_pt_off[(size_t)g] = (long long)(ctx->h_seg_pts.size() / 3);
    for (auto &sp : per_seg[(size_t)g]) {
      double packed[3];
      std::memcpy(packed, &sp, 24);
      ctx->h_seg_pts.insert(ctx->h_seg_pts.end(), packed, packed + 3);
    }
  }
  ctx->h_seg_pt_off[(size_t)ctx->G] = (long long)(ctx->h_seg_pts.size() / 3);
  ctx->max_seg_pts = 0;
  for (long long g = 0; g < ctx->G; ++g)
    ctx->max_seg_pts = std::max(ctx->max_seg_pts, ctx->h_seg_pt_off[(size_t)g + 1] - ctx->h_seg_pt_off[(size_t)g]);
  ctx->pts_ready = true;
  ctx->pts_dirty = true;
  ctx->uploaded = ctx->ran = false;
  return LT_OK;
}

int lt_set_sfm_points(lt_ctx *ctx, int64_t n, const int32_t *ids, const double *xyz) {
  LT_FINISH(ctx);
  ctx->h_sfm_ids.assign(ids, ids + n);
  ctx->h_sfm_xyz.assign(xyz, xyz + 3 * n);
  ctx->sfm_given = n > 0;  // sfm_points_.empty() -> the shared points are triangulated from the two views
  ctx->pts_dirty = true;
  ctx->uploaded = ctx->ran = false;
  return LT_OK;
}

// resolve point3D ids to SfM rows and upload the point tables (called from lt_run_device when needed)
static int upload_points(lt_ctx *ctx) {
  if (!ctx->pts_ready || !ctx->pts_dirty) return LT_OK;
  std::vector<double> pts = ctx->h_seg_pts;
  if (ctx->sfm_given) {
    std::unordered_map<int, int> where;
    for (size_t k = 0; k < ctx->h_sfm_ids.size(); ++k) where[ctx->h_sfm_ids[k]] = (int)k;
    for (size_t e = 0; e + 3 <= pts.size(); e += 3) {
      int head[2];
      std::memcpy(head, &pts[e], 8);
      auto it = where.find(head[0]);
      head[1] = it == where.end() ? -1 : it->second;
      std::memcpy(&pts[e], head, 8);
    }
  }
  if (pts.empty()) pts.assign(3, 0.0);
  int rc;
  if ((rc = upload_vec(ctx, ctx->d_seg_pts, pts))) return rc;
  if ((rc = upload_vec(ctx, ctx->d_seg_pt_off, ctx->h_seg_pt_off))) return rc;
  std::vector<double> xyz = ctx->h_sfm_xyz;
  if (xyz.empty()) xyz.assign(3, 0.0);
  if ((rc = upload_vec(ctx, ctx->d_sfm_xyz, xyz))) return rc;
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  ctx->pts_dirty = false;
  return LT_OK;
}

int lt_init_device(lt_ctx *ctx, int n_img, const int32_t *img_ids, const void *d_kvec, const void *d_qvec,
                   const void *d_tvec, const int64_t *seg_off, const void *d_segs) {
  LT_FINISH(ctx);
  HIPCHK(ctx, hipSetDevice(ctx->device));
  std::vector<int> perm(n_img);
  for (int i = 0; i < n_img; ++i) {
    perm[i] = i;
    if (i > 0 && img_ids[i] <= img_ids[i - 1])
      return fail(ctx, LT_ERR_ARGUMENT, "lt_init_device needs strictly ascending image ids");
  }
  int rc = init_common(ctx, n_img, img_ids, seg_off, perm);
  if (rc) return rc;
  hipStream_t st = ctx->stream;
  ENSURE(ctx, ctx->d_kvec, 32 * (size_t)std::max(n_img, 1));
  ENSURE(ctx, ctx->d_qvec, 32 * (size_t)std::max(n_img, 1));
  ENSURE(ctx, ctx->d_tvec, 24 * (size_t)std::max(n_img, 1));
  ENSURE(ctx, ctx->d_segs_raw, 32 * (size_t)std::max<long long>(ctx->G, 1));
  if (n_img > 0) {
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_kvec.p, d_kvec, 32 * (size_t)n_img, hipMemcpyDeviceToDevice, st));
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_qvec.p, d_qvec, 32 * (size_t)n_img, hipMemcpyDeviceToDevice, st));
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_tvec.p, d_tvec, 24 * (size_t)n_img, hipMemcpyDeviceToDevice, st));
  }
  if (ctx->G > 0)
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_segs_raw.p, d_segs, 32 * (size_t)ctx->G, hipMemcpyDeviceToDevice, st));
  return build_invariants(ctx);
}

int lt_refresh_scene_device(lt_ctx *ctx, const void *d_kvec, const void *d_qvec, const void *d_tvec,
                            const void *d_segs) {
  if (!ctx->inited) return fail(ctx, LT_ERR_STATE, "lt_refresh_scene_device before Init");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  const int n_img = ctx->n_img;
  if (n_img > 0) {
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_kvec.p, d_kvec, 32 * (size_t)n_img, hipMemcpyDeviceToDevice, st));
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_qvec.p, d_qvec, 32 * (size_t)n_img, hipMemcpyDeviceToDevice, st));
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_tvec.p, d_tvec, 24 * (size_t)n_img, hipMemcpyDeviceToDevice, st));
  }
  if (ctx->G > 0)
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_segs_raw.p, d_segs, 32 * (size_t)ctx->G, hipMemcpyDeviceToDevice, st));
  launch_build_cams(st, n_img, ctx->d_kvec.as<double>(), ctx->d_qvec.as<double>(), ctx->d_tvec.as<double>(),
                    ctx->d_cams.as<Cam>());
  ENSURE(ctx, ctx->d_seg_gates, seg_gate_bytes() * (size_t)std::max<long long>(ctx->G, 1));
  launch_build_segs(st, ctx->G, n_img, ctx->d_seg_off.as<long long>(), ctx->d_segs_raw.as<double>(),
                    ctx->cfg.add_halfpix ? 0.5 : 0.0, ctx->d_cams.as<Cam>(), ctx->d_segs.as<Seg>(), ctx->d_seg_gates.p);
  HIPCHK(ctx, hipGetLastError());
  ctx->ran = false;
  return LT_OK;
}

int lt_set_scene_chunks(lt_ctx *ctx, int n_chunks, const int32_t *img_begin, const void *const *d_kvec,
                        const void *const *d_qvec, const void *const *d_tvec, const void *const *d_segs) {
  LT_FINISH(ctx);
  if (!ctx->inited) return fail(ctx, LT_ERR_STATE, "lt_set_scene_chunks before Init");
  if (n_chunks <= 0 || img_begin[0] != 0) return fail(ctx, LT_ERR_ARGUMENT, "chunks must start at image 0");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  std::vector<SceneChunk> ch(n_chunks);
  for (int c = 0; c < n_chunks; ++c) {
    if (img_begin[c] < 0 || img_begin[c] > ctx->n_img || (c > 0 && img_begin[c] < img_begin[c - 1]))
      return fail(ctx, LT_ERR_ARGUMENT, "chunk image ranges must be ascending and inside the scene");
    ch[c].k = (const double *)d_kvec[c]; ch[c].q = (const double *)d_qvec[c]; ch[c].t = (const double *)d_tvec[c];
    ch[c].s = (const double *)d_segs[c];
    ch[c].img_begin = img_begin[c];
    ch[c].seg_begin = ctx->seg_off[img_begin[c]];
    ch[c].pad_ = 0;
  }
  ENSURE(ctx, ctx->d_chunks, sizeof(SceneChunk) * (size_t)n_chunks);
  HIPCHK(ctx, hipMemcpy(ctx->d_chunks.p, ch.data(), sizeof(SceneChunk) * (size_t)n_chunks, hipMemcpyHostToDevice));
  ctx->n_chunks = n_chunks;
  return LT_OK;
}

int lt_refresh_scene_chunks(lt_ctx *ctx) {
  if (!ctx->inited || ctx->n_chunks <= 0) return fail(ctx, LT_ERR_STATE, "lt_refresh_scene_chunks before lt_set_scene_chunks");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  // with an uploaded job only the images it references (triangulated here, or a neighbour) need their
  // segment records; without one, everything
  const bool listed = ctx->uploaded && ctx->n_needed > 0;
  launch_build_scene_chunked(ctx->stream, ctx->n_img, ctx->G, ctx->n_chunks, ctx->d_chunks.as<SceneChunk>(),
                             ctx->d_seg_off.as<long long>(), ctx->cfg.add_halfpix ? 0.5 : 0.0, ctx->d_cams.as<Cam>(),
                             ctx->d_segs.as<Seg>(), listed ? ctx->d_needed.as<int>() : nullptr, ctx->n_needed,
                             ctx->max_needed_segs, ctx->d_seg_gates.p);
  HIPCHK(ctx, hipGetLastError());
  ctx->ran = false;
  return LT_OK;
}

// ---- the host pass over the match rows (lt_rows.h), shared out over the persistent team (lt_pool.h) ----
struct RowBlk {  // one (image, neighbour) block of rows
  const int32_t *src; long long n, dst; long long M1, M2; int img_id, nb_id;
};
struct RowJob {
  const RowBlk *blks; int nb_total; const int *chunk_of; std::atomic<int> *chunk_done; int *bad; unsigned *out;
  std::atomic<int> next_blk{0}, bad_any{0}, uns_any{0};
  static void run(void *arg, int, int) {
    RowJob &J = *static_cast<RowJob *>(arg);
    int uns_t = 0;
    for (;;) {
      const int b = J.next_blk.fetch_add(1, std::memory_order_relaxed);
      if (b >= J.nb_total) break;
      const RowBlk &B = J.blks[b];
      const lt::RowStats rs = lt::pack_rows(B.src, B.n, J.out + B.dst);  // packed: line | neighbour line << 16
      int err = 0;
      if (B.n > 0 && (unsigned long long)rs.mx_line >= (unsigned long long)B.M1) err |= 1;
      if (B.n > 0 && (unsigned long long)rs.mx_ng >= (unsigned long long)B.M2) err |= 2;
      J.bad[b] = err;
      uns_t |= rs.unsorted;
      if (err) J.bad_any.store(1, std::memory_order_relaxed);
      if (J.chunk_done) J.chunk_done[J.chunk_of[b]].fetch_add(1, std::memory_order_release);
    }
    if (uns_t) J.uns_any.store(1, std::memory_order_relaxed);
  }
};

static int begin_image(lt_ctx *ctx, int img_id, int mode, int *idx_out) {
  if (!ctx->inited) return fail(ctx, LT_ERR_STATE, "TriangulateImage called before Init");
  auto it = ctx->id2idx.find(img_id);
  if (it == ctx->id2idx.end()) return fail(ctx, LT_ERR_ARGUMENT, "unknown image id " + std::to_string(img_id));
  *idx_out = it->second;
  // already_scored_ guard (global_line_triangulator.cc:73): the call changes nothing -- in particular it does not
  // invalidate the results or tracks of the batch this image belongs to (the callers return right behind this)
  if (ctx->triangulated[(size_t)it->second]) return LT_OK;
  // ComputeLineTracks ended the batch: with the tail on the device the per-node results are still there -- fetch them
  // now, so that this call starts a new batch (like the host tail, which downloads before it runs) instead of
  // appending to the finished one
  if (ctx->tracks_done && ctx->ran && !ctx->downloaded) {
    int rc = lt_download(ctx);
    if (rc) return rc;
  }
  if (ctx->job_mode != 0 && ctx->job_mode != mode && !ctx->downloaded) {
    int rc = lt_flush(ctx);  // switching between matched and exhaustive calls: run what is buffered
    if (rc) return rc;
  }
  if (ctx->downloaded) {  // a new batch after results were read: start a fresh job
    ctx->job_imgs.clear(); ctx->job_nbs.clear(); ctx->job_order.clear();
    ctx->h_m_off.assign(1, 0); ctx->h_m_pairs.clear(); ctx->streamed_ints = 0;
    ctx->rows_sorted = true;
    ctx->uploaded = ctx->ran = ctx->downloaded = false;
  }
  ctx->job_mode = mode;
  ctx->uploaded = ctx->ran = false;
  ctx->tracks_done = false;
  return LT_OK;
}

// TriangulateImage with the rows of every neighbour given by its own pointer (no concatenation on the
// caller's side).  Validation (base_line_triangulator.cc:79,87-94), the sortedness probe for the
// sort-free placement and the single copy into the staging buffer run in one parallel pass.
int lt_triangulate_image_rows(lt_ctx *ctx, int img_id, int n_nb, const int32_t *nb_ids,
                              const int32_t *const *rows, const int64_t *n_rows) {
  LT_FINISH(ctx);
  struct Acc {  // [12] host ms spent buffering match rows (all calls of the batch)
    lt_ctx *c; double t0;
    ~Acc() { c->timers[12] += now_ms() - t0; }
  } acc{ctx, now_ms()};
  int idx;
  int rc = begin_image(ctx, img_id, 1, &idx);
  if (rc) return rc;
  if (ctx->triangulated[idx]) return LT_OK;  // already_scored_ guard (global_line_triangulator.cc:73)
  if (n_nb > 255) return fail(ctx, LT_ERR_ARGUMENT, "more than 255 neighbours (uint8 neighbour index, base_line_triangulator.h:15)");
  // the reference iterates std::map<int, MatrixXi>: ascending neighbour id
  std::vector<int> order(n_nb);
  for (int k = 0; k < n_nb; ++k) order[k] = k;
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return nb_ids[a] < nb_ids[b]; });
  std::vector<int> nbs(n_nb), ord(n_nb);
  std::vector<long long> M2(n_nb), dst(n_nb + 1, 0);
  const long long M1 = ctx->seg_off[idx + 1] - ctx->seg_off[idx];
  for (int k = 0; k < n_nb; ++k) {
    int o = order[k];
    if (k > 0 && nb_ids[o] == nb_ids[order[k - 1]]) return fail(ctx, LT_ERR_ARGUMENT, "duplicate neighbour id in matches");
    auto it = ctx->id2idx.find(nb_ids[o]);
    if (it == ctx->id2idx.end()) return fail(ctx, LT_ERR_ARGUMENT, "unknown neighbour image id " + std::to_string(nb_ids[o]));
    if (n_rows[o] < 0) return fail(ctx, LT_ERR_ARGUMENT, "negative row count");
    nbs[k] = it->second;
    ord[k] = k;  // already ascending id
    M2[k] = ctx->seg_off[it->second + 1] - ctx->seg_off[it->second];
    dst[k + 1] = dst[k] + n_rows[o];
  }
  const size_t base = ctx->h_m_pairs.size();
  {
    // the staging block may move when it grows: no asynchronous copy may still be reading it
    size_t want = base + (size_t)dst[n_nb];
    if (ctx->job_imgs.empty() && dst[n_nb] > 0)  // first image of a batch: one allocation for the usual case
      // (every image of the scene in one batch; capped at 1 GB -- a large scene arrives in batches, and a
      // page-locked allocation costs ~0.1 s per GB)
      want = std::max(want, std::min<size_t>((size_t)dst[n_nb] * (size_t)std::max(1, ctx->n_img) + 1024, (size_t)1 << 28));
    if (want > ctx->h_m_pairs.capacity()) {
      if (ctx->streamed_ints > 0) HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
      if (!ctx->h_m_pairs.reserve(want)) return fail(ctx, LT_ERR_RUNTIME, "out of host memory for the match rows");
    }
  }
  if (!ctx->h_m_pairs.grow_to(base + (size_t)dst[n_nb])) return fail(ctx, LT_ERR_RUNTIME, "out of host memory for the match rows");
  int *out = ctx->h_m_pairs.data() + base;
  // ONE pass over the rows (lt_rows.h): validation as reductions + the staged copy, packed to one word per row; the
  // blocks are shared out between this thread and the workers of the persistent team that are awake (lt_pool.h)
  std::vector<int> bad(n_nb, 0);
  std::vector<RowBlk> blks((size_t)n_nb);
  for (int k = 0; k < n_nb; ++k)
    blks[(size_t)k] = RowBlk{rows[order[k]], n_rows[order[k]], dst[k], M1, M2[k], img_id, nb_ids[order[k]]};
  RowJob job;
  job.blks = blks.data(); job.nb_total = n_nb; job.chunk_of = nullptr; job.chunk_done = nullptr;
  job.bad = bad.data(); job.out = reinterpret_cast<unsigned *>(out);
  if (dst[n_nb] >= (1 << 14)) {
    lt_host::SpinPool &pool = lt_host::SpinPool::get(lt_host::row_workers());
    pool.begin(&RowJob::run, &job);
    RowJob::run(&job, 0, 0);
    pool.end();
  } else {
    RowJob::run(&job, 0, 0);  // a few rows: not worth a notify
  }
  if (job.uns_any.load()) ctx->rows_sorted = false;
  for (int k = 0; k < n_nb; ++k) {
    if (!bad[k]) continue;
    ctx->h_m_pairs.grow_to(base);
    if (bad[k] & 1)  // base_line_triangulator.cc:87-94
      return fail(ctx, LT_ERR_RUNTIME,
                  "IndexError! Out-of-index matches exist between image (img_id = " + std::to_string(img_id) +
                      ") and neighbor image (img_id = " + std::to_string(nb_ids[order[k]]) +
                      "). Please make sure you are reusing the correct descriptors and matches when using the "
                      "--skip_exists option.");
    return fail(ctx, LT_ERR_RUNTIME, "IndexError! neighbour line id out of range in matches of image " + std::to_string(img_id));
  }
  // stream the rows to the device while the caller prepares the next image (they are final: staging
  // is in call order, which is the device order whenever the images arrive in ascending id order)
  // (one copy per ~4 MB of rows: an enqueue costs the host ~5 us, an image brings ~0.8 MB; lt_upload sends the rest)
  if (ctx->h_m_pairs.blk.pinned && ctx->streamed_ints <= base && dst[n_nb] > 0 &&
      base + (size_t)dst[n_nb] - ctx->streamed_ints >= (1u << 20)) {
    const size_t from = ctx->streamed_ints, end = base + (size_t)dst[n_nb];
    if (hipSetDevice(ctx->device) == hipSuccess) {
      bool ok = true;
      if (sizeof(int) * end > ctx->d_m_pairs.cap) {
        // grow the device buffer (first copy: sized for the whole batch), keeping the streamed prefix
        DevBuf nb;
        size_t want = sizeof(int) * std::max(end, ctx->h_m_pairs.capacity());
        ok = nb.ensure(want);
        if (ok && from > 0)
          ok = hipMemcpyAsync(nb.p, ctx->d_m_pairs.p, sizeof(int) * from, hipMemcpyDeviceToDevice, ctx->stream) == hipSuccess &&
               hipStreamSynchronize(ctx->stream) == hipSuccess;
        if (ok) {
          ctx->d_m_pairs.release();
          ctx->d_m_pairs = nb;
        } else {
          nb.release();
          (void)hipGetLastError();
        }
      }
      if (ok && hipMemcpyAsync(ctx->d_m_pairs.as<int>() + from, ctx->h_m_pairs.data() + from, sizeof(int) * (end - from),
                               hipMemcpyHostToDevice, ctx->stream) == hipSuccess)
        ctx->streamed_ints = end;
      else
        (void)hipGetLastError();  // not fatal: lt_upload sends whatever was not streamed
    }
  }
  for (int k = 0; k < n_nb; ++k) ctx->h_m_off.push_back(ctx->h_m_off.back() + n_rows[order[k]]);
  ctx->job_imgs.push_back(idx);
  ctx->job_nbs.push_back(nbs);
  ctx->job_order.push_back(ord);
  ctx->neighbors[idx] = nbs;
  ctx->triangulated[idx] = 1;
  return LT_OK;
}

int lt_triangulate_all_rows(lt_ctx *ctx, int n_images, const int32_t *img_ids, const int64_t *nb_off, const int32_t *nb_ids,
                            const int32_t *const *rows, const int64_t *n_rows) {
  LT_FINISH(ctx);
  struct Acc {  // [12] host ms spent buffering match rows
    lt_ctx *c; double t0;
    ~Acc() { c->timers[12] += now_ms() - t0; }
  } acc{ctx, now_ms()};
  if (n_images < 0 || (n_images > 0 && (!img_ids || !nb_off))) return fail(ctx, LT_ERR_ARGUMENT, "null argument");
  struct Img {
    int idx; std::vector<int> nbs, ord; std::vector<long long> cnt;
  };
  static const bool all_trace = getenv("LT_TAIL_TRACE") != nullptr;
  double tl = acc.t0;
  auto lap = [&](const char *what) {
    if (!all_trace) return;
    double t = now_ms();
    std::fprintf(stderr, "[all] %-18s %.3f ms\n", what, t - tl);
    tl = t;
  };
  std::vector<RowBlk> blks;
  std::vector<Img> imgs;
  const size_t base = ctx->h_m_pairs.size();
  long long total_rows = 0;
  std::vector<char> seen_here((size_t)std::max(ctx->n_img, 1), 0);
  // ---- pass 1 (serial, cheap): the per-image bookkeeping of lt_triangulate_image_rows, block descriptors ----
  for (int k = 0; k < n_images; ++k) {
    int idx;
    int rc = begin_image(ctx, img_ids[k], 1, &idx);
    if (rc) return rc;
    if (ctx->triangulated[idx] || seen_here[(size_t)idx]) continue;  // already_scored_ guard (global_line_triangulator.cc:73)
    seen_here[(size_t)idx] = 1;
    const int n_nb = (int)(nb_off[k + 1] - nb_off[k]);
    const int32_t *nb = nb_ids + nb_off[k];
    if (n_nb > 255) return fail(ctx, LT_ERR_ARGUMENT, "more than 255 neighbours (uint8 neighbour index, base_line_triangulator.h:15)");
    std::vector<int> order(n_nb);
    for (int e = 0; e < n_nb; ++e) order[e] = e;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return nb[a] < nb[b]; });  // std::map order
    Img im;
    im.idx = idx;
    const long long M1 = ctx->seg_off[idx + 1] - ctx->seg_off[idx];
    for (int e = 0; e < n_nb; ++e) {
      const int o = order[e];
      if (e > 0 && nb[o] == nb[order[e - 1]]) return fail(ctx, LT_ERR_ARGUMENT, "duplicate neighbour id in matches");
      auto it = ctx->id2idx.find(nb[o]);
      if (it == ctx->id2idx.end()) return fail(ctx, LT_ERR_ARGUMENT, "unknown neighbour image id " + std::to_string(nb[o]));
      const long long n = n_rows[nb_off[k] + o];
      if (n < 0) return fail(ctx, LT_ERR_ARGUMENT, "negative row count");
      im.nbs.push_back(it->second);
      im.ord.push_back(e);
      im.cnt.push_back(n);
      blks.push_back(RowBlk{rows[nb_off[k] + o], n, total_rows, M1, ctx->seg_off[it->second + 1] - ctx->seg_off[it->second],
                         img_ids[k], nb[o]});
      total_rows += n;
    }
    imgs.push_back(std::move(im));
  }
  if (imgs.empty()) return LT_OK;
  lap("bookkeeping");
  // ---- staging: one allocation for the whole call ----
  {
    const size_t want = base + (size_t)total_rows;
    if (want > ctx->h_m_pairs.capacity()) {
      if (ctx->streamed_ints > 0) HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
      if (!ctx->h_m_pairs.reserve(want)) return fail(ctx, LT_ERR_RUNTIME, "out of host memory for the match rows");
    }
    if (!ctx->h_m_pairs.grow_to(want)) return fail(ctx, LT_ERR_RUNTIME, "out of host memory for the match rows");
  }
  int *out = ctx->h_m_pairs.data() + base;
  lap("staging");
  // ---- pass 2: validation (reductions over the rows) + the single copy, in CHUNKS of >= 8 MB of rows: one parallel
  // region per chunk over its (image, neighbour) blocks, and the chunk's host -> device copy enqueued right behind it, so
  // that the DMA of chunk c runs under the host pass of chunk c + 1 (one copy at the end left 1.5 ms of DMA exposed) ----
  const int nb_total = (int)blks.size();
  std::vector<int> bad((size_t)nb_total, 0);
  int unsorted = 0;
  bool stream_ok = ctx->h_m_pairs.blk.pinned && ctx->streamed_ints <= base && total_rows > 0 &&
                   hipSetDevice(ctx->device) == hipSuccess;
  if (stream_ok && ctx->streamed_ints < base) {
    // rows of earlier calls that were not streamed yet go first (the device buffer is filled in order)
    stream_ok = false;
  }
  if (stream_ok) {
    const size_t end = base + (size_t)total_rows;
    if (sizeof(int) * end > ctx->d_m_pairs.cap) {
      DevBuf nbuf;
      bool ok = nbuf.ensure(sizeof(int) * std::max(end, ctx->h_m_pairs.capacity()));
      if (ok && base > 0)
        ok = hipMemcpyAsync(nbuf.p, ctx->d_m_pairs.p, sizeof(int) * base, hipMemcpyDeviceToDevice, ctx->stream) == hipSuccess &&
             hipStreamSynchronize(ctx->stream) == hipSuccess;
      if (ok) {
        ctx->d_m_pairs.release();
        ctx->d_m_pairs = nbuf;
      } else {
        nbuf.release();
        (void)hipGetLastError();
        stream_ok = false;
      }
    }
  }
  // The workers (lt_pool.h: a persistent team, already awake when Init preceded this call) take blocks in order from
  // a shared counter; this thread does no row work -- it waits for each chunk (>= 2 MB of packed rows) to be complete
  // and enqueues its host -> device copy, so the DMA of chunk c runs under the workers' pass over chunk c + 1 (a copy
  // at the very end left 1.5 ms of DMA exposed; 8 MB chunks delayed the first copy by a fifth of the pass)
  constexpr long long kChunkRows = 512 << 10;
  std::vector<int> chunk_end;  // block index behind every chunk
  {
    long long acc_rows = 0;
    for (int b = 0; b < nb_total; ++b) {
      acc_rows += blks[(size_t)b].n;
      if (acc_rows >= kChunkRows || b == nb_total - 1) {
        chunk_end.push_back(b + 1);
        acc_rows = 0;
      }
    }
  }
  const int n_chunks = (int)chunk_end.size();
  std::vector<int> chunk_of((size_t)nb_total);
  for (int c = 0, b = 0; c < n_chunks; ++c)
    for (; b < chunk_end[(size_t)c]; ++b) chunk_of[(size_t)b] = c;
  std::vector<std::atomic<int>> chunk_done((size_t)n_chunks);
  for (auto &x : chunk_done) x.store(0, std::memory_order_relaxed);
  RowJob job;
  job.blks = blks.data(); job.nb_total = nb_total; job.chunk_of = chunk_of.data(); job.chunk_done = chunk_done.data();
  job.bad = bad.data(); job.out = reinterpret_cast<unsigned *>(out);
  int *const d_rows = stream_ok ? ctx->d_m_pairs.as<int>() : nullptr;
  int *const h_rows = ctx->h_m_pairs.data();
  size_t streamed_to = ctx->streamed_ints;
  lap("device buffer");
  lt_host::SpinPool &pool = lt_host::SpinPool::get(lt_host::row_workers());
  pool.begin(&RowJob::run, &job);
  {
    bool ok = d_rows != nullptr;
    for (int c = 0; c < n_chunks; ++c) {
      const int first = c == 0 ? 0 : chunk_end[(size_t)c - 1], want = chunk_end[(size_t)c] - first;
      while (chunk_done[(size_t)c].load(std::memory_order_acquire) < want) {
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
      }
      if (!ok || job.bad_any.load(std::memory_order_relaxed)) continue;
      const size_t from = base + (size_t)blks[(size_t)first].dst;
      const size_t to = base + (size_t)(blks[(size_t)chunk_end[(size_t)c] - 1].dst + blks[(size_t)chunk_end[(size_t)c] - 1].n);
      if (to > from) {
        if (hipMemcpyAsync(d_rows + from, h_rows + from, sizeof(int) * (to - from), hipMemcpyHostToDevice, ctx->stream) == hipSuccess)
          streamed_to = to;
        else {
          (void)hipGetLastError();  // not fatal: lt_upload sends whatever was not streamed
          ok = false;
        }
      }
    }
  }
  pool.end();
  lap("row pass");
  unsorted = job.uns_any.load();
  const bool any_bad = job.bad_any.load() != 0;
  if (streamed_to > ctx->streamed_ints) ctx->streamed_ints = streamed_to;
  for (int b = 0; b < nb_total && any_bad; ++b) {  // the first offending block in call order raises, like the per-image calls
    if (!bad[(size_t)b]) continue;
    if (ctx->streamed_ints > base) {  // chunks of this call are already on their way: they are void
      (void)hipStreamSynchronize(ctx->stream);
      ctx->streamed_ints = base;
    }
    ctx->h_m_pairs.grow_to(base);
    const RowBlk &B = blks[(size_t)b];
    if (bad[(size_t)b] & 1)  // base_line_triangulator.cc:87-94
      return fail(ctx, LT_ERR_RUNTIME,
                  "IndexError! Out-of-index matches exist between image (img_id = " + std::to_string(B.img_id) +
                      ") and neighbor image (img_id = " + std::to_string(B.nb_id) +
                      "). Please make sure you are reusing the correct descriptors and matches when using the "
                      "--skip_exists option.");
    return fail(ctx, LT_ERR_RUNTIME, "IndexError! neighbour line id out of range in matches of image " + std::to_string(B.img_id));
  }
  if (unsorted) ctx->rows_sorted = false;
  for (Img &im : imgs) {
    for (long long n : im.cnt) ctx->h_m_off.push_back(ctx->h_m_off.back() + n);
    ctx->job_imgs.push_back(im.idx);
    ctx->neighbors[im.idx] = im.nbs;
    ctx->job_nbs.push_back(std::move(im.nbs));
    ctx->job_order.push_back(std::move(im.ord));
    ctx->triangulated[im.idx] = 1;
  }
  return LT_OK;
}

int lt_triangulate_image(lt_ctx *ctx, int img_id, int n_nb, const int32_t *nb_ids, const int64_t *m_off,
                         const int32_t *m_pairs) {
  std::vector<const int32_t *> rows(std::max(n_nb, 1));
  std::vector<int64_t> n_rows(std::max(n_nb, 1));
  for (int k = 0; k < n_nb; ++k) {
    rows[k] = m_pairs + 2 * m_off[k];
    n_rows[k] = m_off[k + 1] - m_off[k];
  }
  return lt_triangulate_image_rows(ctx, img_id, n_nb, nb_ids, rows.data(), n_rows.data());
}

int lt_triangulate_image_exhaustive(lt_ctx *ctx, int img_id, int n_nb, const int32_t *nb_ids) {
  LT_FINISH(ctx);
  int idx;
  int rc = begin_image(ctx, img_id, 2, &idx);
  if (rc) return rc;
  if (ctx->triangulated[idx]) return LT_OK;
  if (n_nb > 255) return fail(ctx, LT_ERR_ARGUMENT, "more than 255 neighbours (uint8 neighbour index, base_line_triangulator.h:15)");
  std::vector<int> nbs;
  for (int k = 0; k < n_nb; ++k) {
    auto it = ctx->id2idx.find(nb_ids[k]);
    if (it == ctx->id2idx.end()) return fail(ctx, LT_ERR_ARGUMENT, "unknown neighbour image id " + std::to_string(nb_ids[k]));
    for (int p = 0; p < k; ++p)
      if (nb_ids[p] == nb_ids[k]) return fail(ctx, LT_ERR_ARGUMENT, "duplicate neighbour id in neighbors list");
    nbs.push_back(it->second);
  }
  // exhaustive mode keeps the caller's neighbour order (:113-114); the per-image support sum
  // still runs over ascending image ids (std::map score_table, global_line_triangulator.cc:83,110)
  std::vector<int> ord(n_nb);
  for (int k = 0; k < n_nb; ++k) ord[k] = k;
  std::stable_sort(ord.begin(), ord.end(), [&](int a, int b) { return nb_ids[a] < nb_ids[b]; });
  ctx->job_imgs.push_back(idx);
  ctx->job_nbs.push_back(nbs);
  ctx->job_order.push_back(ord);
  ctx->neighbors[idx] = nbs;
  ctx->triangulated[idx] = 1;
  return LT_OK;
}

int lt_upload(lt_ctx *ctx) {
  LT_RANGE("lt_upload (match rows + job tables -> HBM)");
  LT_FINISH(ctx);
  if (!ctx->inited) return fail(ctx, LT_ERR_STATE, "upload before Init");
  if (ctx->uploaded) return LT_OK;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  double t0 = now_ms();
  build_job_tables(ctx);
  int rc;
  {
    // images referenced by the job (lt_refresh_scene_chunks rebuilds only their segment records)
    std::vector<char> need((size_t)std::max(ctx->n_img, 1), 0);
    for (size_t j = 0; j < ctx->job_imgs.size(); ++j) {
      need[(size_t)ctx->job_imgs[j]] = 1;
      for (int nb : ctx->job_nbs[j]) need[(size_t)nb] = 1;
    }
    std::vector<int> list;
    ctx->max_needed_segs = 0;
    for (int i = 0; i < ctx->n_img; ++i)
      if (need[(size_t)i]) {
        list.push_back(i);
        ctx->max_needed_segs = std::max(ctx->max_needed_segs, ctx->seg_off[i + 1] - ctx->seg_off[i]);
      }
    ctx->n_needed = (int)list.size();
    if (list.empty()) list.push_back(0);
    if ((rc = upload_vec(ctx, ctx->d_needed, list))) return rc;
  }
  if ((rc = upload_vec(ctx, ctx->d_nb_off, ctx->h_nb_off))) return rc;
  if ((rc = upload_vec(ctx, ctx->d_blk_img, ctx->h_blk_img))) return rc;
  if ((rc = upload_vec(ctx, ctx->d_blk_nb, ctx->h_blk_nb))) return rc;
  if ((rc = upload_vec(ctx, ctx->d_blk_slot, ctx->h_blk_slot))) return rc;
  if ((rc = upload_vec(ctx, ctx->d_blk_order, ctx->h_blk_order))) return rc;
  if ((rc = upload_vec(ctx, ctx->d_blk_line_base, ctx->h_blk_line_base))) return rc;
  if (ctx->job_mode == 1) {
    // block order in the tables is image-index-major; the staging arrays are call-order-major:
    // re-pack rows so that block b of the table owns rows m_off[b]..m_off[b+1]
    std::vector<long long> call_first_blk(ctx->job_imgs.size() + 1, 0);
    for (size_t j = 0; j < ctx->job_imgs.size(); ++j) call_first_blk[j + 1] = call_first_blk[j] + (long long)ctx->job_nbs[j].size();
    std::vector<int> job_pos(ctx->n_img, -1);
    for (size_t j = 0; j < ctx->job_imgs.size(); ++j) job_pos[ctx->job_imgs[j]] = (int)j;
    std::vector<long long> m_off(ctx->n_blk + 1, 0);
    bool in_order = true;
    {
      long long b = 0;
      for (int i = 0; i < ctx->n_img; ++i) {
        int j = job_pos[i];
        if (j < 0) continue;
        if (call_first_blk[j] != b) in_order = false;
        for (size_t k = 0; k < ctx->job_nbs[j].size(); ++k, ++b) {
          long long cb = call_first_blk[j] + (long long)k;
          m_off[b + 1] = m_off[b] + (ctx->h_m_off[cb + 1] - ctx->h_m_off[cb]);
        }
      }
    }
    ctx->P = m_off[ctx->n_blk];
    ctx->n_conn = ctx->P;
    ctx->max_rows = 0;
    for (int bq = 0; bq < ctx->n_blk; ++bq) ctx->max_rows = std::max(ctx->max_rows, m_off[bq + 1] - m_off[bq]);
    if (ctx->P >= (1ll << 32) - 1) return fail(ctx, LT_ERR_ARGUMENT, "too many match rows in one batch (>= 2^32-1)");
    if (sizeof(int) * (size_t)std::max<long long>(ctx->P, 1) > ctx->d_m_pairs.cap) {
      HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
      ctx->streamed_ints = 0;  // the buffer is replaced: everything is sent again
      ENSURE(ctx, ctx->d_m_pairs, sizeof(int) * (size_t)std::max<long long>(ctx->P, 1));
    }
    if (in_order) {
      // call order == device order: only what was not streamed during buffering is still to be sent
      const size_t total = (size_t)ctx->P, sent = std::min(ctx->streamed_ints, total);
      if (total > sent)
        HIPCHK(ctx, hipMemcpyAsync(ctx->d_m_pairs.as<int>() + sent, ctx->h_m_pairs.data() + sent, sizeof(int) * (total - sent),
                                   hipMemcpyHostToDevice, ctx->stream));
    } else {
      long long b = 0;
      for (int i = 0; i < ctx->n_img; ++i) {
        int j = job_pos[i];
        if (j < 0) continue;
        for (size_t k = 0; k < ctx->job_nbs[j].size(); ++k, ++b) {
          long long cb = call_first_blk[j] + (long long)k;
          long long n = ctx->h_m_off[cb + 1] - ctx->h_m_off[cb];
          if (n > 0)
            HIPCHK(ctx, hipMemcpyAsync(ctx->d_m_pairs.as<int>() + m_off[b], ctx->h_m_pairs.data() + ctx->h_m_off[cb],
                                       sizeof(int) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
        }
      }
    }
    if ((rc = upload_vec(ctx, ctx->d_m_off, m_off))) return rc;
    // per-block records of the matched pipeline (row range, images, segment bases): a function of the job
    ENSURE(ctx, ctx->d_blkrec, blk_rec_bytes() * (size_t)std::max(ctx->n_blk, 1));
    launch_build_blk(ctx->stream, ctx->n_blk, ctx->d_m_off.as<long long>(), ctx->d_blk_img.as<int>(),
                     ctx->d_blk_nb.as<int>(), ctx->d_blk_slot.as<int>(), ctx->d_seg_off.as<long long>(),
                     ctx->d_blk_line_base.as<long long>(), ctx->d_blkrec.p);
  } else if (ctx->job_mode == 2) {
    // work items: per node, per neighbour block, chunks of 64 neighbour lines
    ctx->h_item_off.assign(ctx->G + 1, 0);
    // per block: chunks of the earlier neighbour blocks of the same image (the item index of
    // (node, block, chunk) is item_off[node] + blk_chunk_off[block] + chunk -- no search on the device)
    std::vector<int> blk_chunk_off((size_t)std::max(ctx->n_blk, 1), 0);
    ctx->max_chunks = 1;
    long long items = 0, conns = 0;
    for (int i = 0; i < ctx->n_img; ++i) {
      long long per_node = 0, conn_node = 0;
      for (long long b = ctx->h_nb_off[i]; b < ctx->h_nb_off[i + 1]; ++b) {
        int i2 = ctx->h_blk_nb[b];
        long long M2 = ctx->seg_off[i2 + 1] - ctx->seg_off[i2];
        blk_chunk_off[(size_t)b] = (int)per_node;
        ctx->max_chunks = std::max(ctx->max_chunks, (int)((M2 + 63) / 64));
        per_node += (M2 + 63) / 64;
        conn_node += M2;
      }
      for (long long g = ctx->seg_off[i]; g < ctx->seg_off[i + 1]; ++g) {
        ctx->h_item_off[g] = items;
        items += per_node;
        conns += conn_node;
      }
    }
    ctx->h_item_off[ctx->G] = items;
    ctx->P = items;
    ctx->n_conn = conns;
    if ((rc = upload_vec(ctx, ctx->d_item_off, ctx->h_item_off))) return rc;
    if ((rc = upload_vec(ctx, ctx->d_blk_chunk_off, blk_chunk_off))) return rc;
  } else {
    ctx->P = 0;
    ctx->n_conn = 0;
  }
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  ctx->uploaded = true;
  ctx->ran = false;
  ctx->timers[8] = now_ms() - t0;
  return LT_OK;
}

// Completes the run that lt_run_device_async left in flight: waits for its end marker, reads the error flag,
// the candidate count and the pair statistic from the pinned slots of its set, and its event timings.
int finish_run(lt_ctx *ctx) {
  LT_RANGE("lt_sync (end of run: result scalars, event timings)");
  if (!ctx->run_pending) return LT_OK;
  ctx->run_pending = false;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  hipEvent_t *ev = ctx->pend_set ? ctx->ev_b : ctx->ev;
  long long *hp = ctx->h_pinned ? ctx->h_pinned + 8 * ctx->pend_set : nullptr;
  int derr = 0;
  if (hp) {
    HIPCHK(ctx, hipEventSynchronize(ev[12]));
    derr = (int)hp[1];
    ctx->stat_pairs_eval = hp[2];
    ctx->C_last = ctx->pend_count_on_device ? hp[0] : ctx->pend_C;
  } else {
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    HIPCHK(ctx, hipMemcpy(&derr, ctx->d_err.p, sizeof(int), hipMemcpyDeviceToHost));
    unsigned long long pe = 0;
    ctx->C_last = ctx->pend_C;
    if (ctx->pend_count_on_device)
      HIPCHK(ctx, hipMemcpy(&ctx->C_last, ctx->d_tri_off.as<long long>() + ctx->G, 8, hipMemcpyDeviceToHost));
    if (ctx->C_last > 0) HIPCHK(ctx, hipMemcpy(&pe, ctx->d_pair_counter.p, 8, hipMemcpyDeviceToHost));
    ctx->stat_pairs_eval = (long long)pe;
  }
  ctx->stat_survivors = -1;  // summed on demand (lt_get_timers)
  if (derr == 5) {
    // the staging capacity of the one-pass exhaustive mode did not hold: repeat the job in the two-pass form (exact
    // sizes).  When this is the earlier of two runs in flight the later one -- same inputs -- is repeated at its own end.
    if (ctx->ex_retry_depth > 0 || !ctx->ex_staged_set[ctx->pend_set])
      return fail(ctx, LT_ERR_RUNTIME, "internal: candidate staging overflow outside the one-pass exhaustive mode");
    ctx->ex_two_pass = true;
    // the counters kept counting beyond the capacity: the next run gets what this one would have needed
    if (hp && hp[3] > 0 && ctx->n_conn > 0)
      ctx->ex_frac = 1.4 * (double)hp[3] * (double)ex_regions() / (double)ctx->n_conn;
    if (ctx->in_run_async) return LT_OK;
    ctx->ex_retry_depth = 1;
    int rc2 = lt_run_device_async(ctx);
    if (!rc2) rc2 = finish_run(ctx);
    ctx->ex_retry_depth = 0;
    return rc2;
  }
  ctx->timers[17] = ctx->timers[18] = 0.0;
  if (hp && ctx->ex_staged_set[ctx->pend_set]) {
    ctx->timers[17] = (double)hp[3] * (double)ex_regions();
    ctx->timers[18] = (double)ctx->ex_region_cap * (double)ex_regions();
  }
  if (ctx->job_mode == 2 && derr == 0 && ctx->n_conn > 0) {
    // this run's need of staging slots -> capacity of the next one: 1.4 x the fullest region (one-pass form), or an
    // estimate from the candidate count (two-pass form: slots = listed connections + block padding, ~1.5 per candidate)
    if (hp && ctx->ex_staged_set[ctx->pend_set])
      ctx->ex_frac = std::max(1.4 * (double)hp[3] * (double)ex_regions() / (double)ctx->n_conn, 1e-4);
    else if (ctx->ex_frac <= 0.0)
      ctx->ex_frac = std::max(2.2 * (double)ctx->C_last / (double)ctx->n_conn, 1e-4);
    ctx->ex_two_pass = false;
  }
  if (derr == 4) return fail(ctx, LT_ERR_RUNTIME, "internal: the scan of the node counts did not complete");
  if (derr == 7) {
    // the pair list of the three-kernel scoring did not hold: repeat the run with the fused kernel (same results)
    if (ctx->score_split_off) return fail(ctx, LT_ERR_RUNTIME, "internal: pair list overflow with the fused scoring kernel");
    ctx->score_split_off = true;
    if (ctx->in_run_async) return LT_OK;
    int rc2 = lt_run_device_async(ctx);
    if (!rc2) rc2 = finish_run(ctx);
    return rc2;
  }
  if (derr == 3)
    return fail(ctx, LT_ERR_RUNTIME, "the one-point proposal supports at most 250 shared points per connection");
  if (derr == 2)
    return fail(ctx, LT_ERR_RUNTIME, "map::at: a point shared by two lines has a point3D_id that is not among the SfM points");
  if (derr != 0) return fail(ctx, LT_ERR_RUNTIME, "IndexError! Out-of-index matches detected on the device");
  // coarse stages from five events (every hipEventRecord between kernels costs ~1.5 us of device time):
  // [3] generation incl. the pair records = ev0..ev3, [4] placement = ev3..ev4, [5] scoring incl. its per-candidate
  // records = ev4..ev5, [6] selection = ev5..ev7; [1], [2], [7] are no longer separate stages
  float ms;
  ctx->timers[1] = ctx->timers[2] = ctx->timers[7] = 0.0;
  const int eg = ctx->pend_ev_gen_end, ep = ctx->pend_ev_place_end;
  const int ee = hp ? 12 : 7;  // end of the run: the end marker behind the result copies, if there are result slots
  const int kA[4] = {0, eg, ep, 5}, kB[4] = {eg, ep, 5, ee}, kT[4] = {3, 4, 5, 6};
  for (int k = 0; k < 4; ++k) {
    HIPCHK(ctx, hipEventElapsedTime(&ms, ev[kA[k]], ev[kB[k]]));
    ctx->timers[kT[k]] = ms;
  }
  HIPCHK(ctx, hipEventElapsedTime(&ms, ev[0], ev[ee]));
  ctx->timers[0] = ms;
  // single-kernel durations of the matched pipeline: [13] k_gates, [14] k_tri_rows, [15] k_score3
  ctx->timers[13] = ctx->timers[14] = ctx->timers[15] = 0.0;
  if (ctx->pend_fine_gen && ctx->job_mode == 1 && ctx->n_blk > 0 && ctx->max_rows > 0) {
    if (hipEventElapsedTime(&ms, ev[8], ev[9]) == hipSuccess) ctx->timers[13] = ms;
    if (hipEventElapsedTime(&ms, ev[9], ev[10]) == hipSuccess) ctx->timers[14] = ms;
  }
  if (ctx->pend_fine_score && ctx->C_last > 0 && hipEventElapsedTime(&ms, ev[11], ev[5]) == hipSuccess) ctx->timers[15] = ms;
  (void)hipGetLastError();
  ctx->timers[11] = (double)ctx->stat_pairs_eval;
  if (const char *mode = getenv("LT_EXP_TILE_ORDER")) {  // developer experiment: tile order from the node sizes of this run
    if (ctx->exp_tile_order_C != ctx->C_last && ctx->C_last > 0 && ctx->job_mode == 1) {
      const long long G = ctx->G, C = ctx->C_last;
      std::vector<long long> off((size_t)G + 1);
      HIPCHK(ctx, hipMemcpy(off.data(), ctx->d_tri_off.p, 8 * (size_t)(G + 1), hipMemcpyDeviceToHost));
      const long long nt = (C + 63) / 64;
      std::vector<long long> cost((size_t)nt, 0);
      for (long long g = 0; g < G; ++g) {
        const long long n = off[g + 1] - off[g];
        for (long long c = off[g]; c < off[g + 1]; ++c) cost[(size_t)(c >> 6)] += n;
      }
      std::vector<unsigned> order((size_t)nt);
      for (long long t = 0; t < nt; ++t) order[(size_t)t] = (unsigned)t;
      if (mode[0] == 'l') {  // lpt
        std::stable_sort(order.begin(), order.end(), [&](unsigned x, unsigned y) { return cost[x] > cost[y]; });
      } else {  // cheapest 30 % last, natural order inside the classes
        std::vector<long long> sorted(cost);
        std::sort(sorted.begin(), sorted.end());
        const long long thr = sorted[(size_t)(0.3 * (double)nt)];
        std::stable_partition(order.begin(), order.end(), [&](unsigned x) { return cost[x] > thr; });
      }
      ENSURE(ctx, ctx->d_exp_tile_order, 4 * (size_t)nt);
      HIPCHK(ctx, hipMemcpy(ctx->d_exp_tile_order.p, order.data(), 4 * (size_t)nt, hipMemcpyHostToDevice));
      ctx->exp_tile_order_C = C;
    }
  }
  for (int k = 0; k < 24; ++k)  // [16] (survivors) is counted on demand by lt_get_timers, not per run
    if (k != 8 && k != 9 && k != 10 && k != 12 && k != 16) ctx->timer_sums[k] += ctx->timers[k];
  ++ctx->timer_runs;
  return LT_OK;
}
int lt_sync(lt_ctx *ctx) { return finish_run(ctx); }

// Enqueues the whole run and returns.  A run still in flight from the previous call is completed AFTER the
// new one has been enqueued (its errors are the return value), so a caller that streams batches keeps the
// device busy across the host's end-of-run bookkeeping.  Two sets of events / pinned result slots alternate.
int lt_run_device_async(lt_ctx *ctx) {
  LT_RANGE("lt_run_device (enqueue: generation, placement, scoring, selection)");
  if (!ctx->uploaded) return fail(ctx, LT_ERR_STATE, "lt_run_device before lt_upload");
  if (!ctx->h_pinned) LT_FINISH(ctx);  // no pinned result slots: nothing may stay in flight
  const int set = ctx->run_pending ? (ctx->pend_set ^ 1) : 0;
  if (set == 1 && !ctx->ev_b[0])
    for (auto &e : ctx->ev_b) HIPCHK(ctx, hipEventCreate(&e));
  hipEvent_t *ev = set ? ctx->ev_b : ctx->ev;
  long long *hp = ctx->h_pinned ? ctx->h_pinned + 8 * set : nullptr;
  ctx->ex_staged_set[set] = false;
  const bool fine_gen = fine_gen_timers(), fine_score = fine_timers();
  HIPCHK(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  const long long G = ctx->G, P = ctx->P;
  {
    int rcp = upload_points(ctx);
    if (rcp) return rcp;
  }
  GenCfg gcfg = make_gen(ctx);
  // like the VP proposals, the point-guided ones do not depend on the algebraic gates
  if (ctx->pts_ready && (!ctx->cfg.disable_many_points_triangulation || !ctx->cfg.disable_one_point_triangulation))
    gcfg.force_undecided = 1;
  const ScoreCfg scfg = make_score(ctx);
  ENSURE(ctx, ctx->d_err, sizeof(int));
  ENSURE(ctx, ctx->d_pair_counter, 8);
  ENSURE(ctx, ctx->d_result3, 32);
  ENSURE(ctx, ctx->d_pairs, sizeof(PairRec) * (size_t)std::max(ctx->n_blk, 1));
  ENSURE(ctx, ctx->d_tri_off, sizeof(long long) * (size_t)(G + 1));
  HIPCHK(ctx, hipEventRecord(ev[0], st));
  // also zeroes the error flag, the pair statistic and the look-back state of k_node_prefix's scan
  // (+ the tile cost-class counters of k_cand_meta / k_score3 behind the scan's words: zeroed by the same kernel)
  const int n_status_scan = (int)((G + 1 + 255) / 256) + 1;
  // + the staging counters of the one-pass exhaustive mode; all counters 128 bytes apart
  const int n_status = n_status_scan + score3_tile_buckets() * 16 + ex_regions() * 16;
  ENSURE(ctx, ctx->d_scan_status, 8 * (size_t)n_status);
  launch_build_pairs(st, ctx->n_blk, ctx->d_blk_img.as<int>(), ctx->d_blk_nb.as<int>(), ctx->d_cams.as<Cam>(),
                     ctx->d_pairs.as<PairRec>(), ctx->d_err.as<int>(),
                     ctx->d_pair_counter.as<unsigned long long>(), ctx->d_scan_status.as<unsigned long long>(),
                     n_status);

  long long C_known = -1;  // candidate count once it is known on the host
  long long C_bound = 0;   // what sizes the compact arrays: the count, or an upper bound while it stays on the device
  int ev_gen_end = 3, ev_place_end = 4;  // events that close the generation / placement stage (see finish_run)
  long long C_run = 0;  // what finish_run reports as the run's candidate count unless the device copy does
  if (ctx->job_mode == 1) {
    const size_t Pn = (size_t)std::max<long long>(P, 1);
    const bool fast = ctx->rows_sorted;
    const long long n_waves = (long long)ctx->n_blk * gen_groups(ctx->max_rows);  // candidate lists
    const long long n_slots_all = (long long)ctx->n_blk * gen_slots(ctx->max_rows);  // survivor lists
    const long long n_entries = ctx->h_blk_line_base[ctx->n_blk];
    // ---- generation in row order; valid candidates appended in row order to per-wave lists ----
    // VP-guided proposals: up to three candidates per match row (vp of l1, vp of l2, algebraic)
    const bool vp_on = ctx->cfg.use_vp && !ctx->cfg.disable_vp_triangulation;
    if (vp_on && !ctx->vp_ready) return fail(ctx, LT_ERR_STATE, "use_vp is set but InitVPResults was not called");
    // point-guided proposals (SetBipartites2d): the many-points line fit (base_line_triangulator.cc:183-236) and the
    // one-point proposal (:238-248, one candidate per shared point; lt_devfn.h: one_point_candidate)
    const bool pts_any = ctx->pts_ready && (!ctx->cfg.disable_many_points_triangulation || !ctx->cfg.disable_one_point_triangulation);
    const bool many_on = pts_any && !ctx->cfg.disable_many_points_triangulation;
    const bool one_on = pts_any && !ctx->cfg.disable_one_point_triangulation;
    const bool pts_on = pts_any;
    // staging slots per match row: many-points, one candidate per shared point (at most the most points any
    // segment has, capped at kMaxOnePoints = 250 -- the kernel flags a connection with more), vp(l1), vp(l2), algebraic
    int mult = (vp_on || pts_on) ? 4 : 1;
    if (one_on) mult += (int)std::min<long long>(ctx->max_seg_pts, kMaxOnePoints);
    if (mult > 1 && (long long)mult * P >= (1ll << 32) - 1)
      return fail(ctx, LT_ERR_ARGUMENT, "too many match rows in one batch for the extra proposals");
    ENSURE(ctx, ctx->d_st_c, sizeof(CRec) * Pn * mult); ENSURE(ctx, ctx->d_st_l, sizeof(double) * Pn * mult);
    ENSURE(ctx, ctx->d_st_key, 4 * Pn * mult);
    ENSURE(ctx, ctx->d_wave_count, 4 * (size_t)(n_waves + 1));
    ENSURE(ctx, ctx->d_ntris_u, 4 * (size_t)(G + 1));
    if (fast) {
      // per-(block, line) counters: k_node_prefix zeroes every counter it reads, so the array only has to
      // be cleared when it is new or when the previous run did not get that far
      const size_t nb = 4 * (size_t)std::max<long long>(n_entries, 1);
      const void *before = ctx->d_cnt_bl.p;
      ENSURE(ctx, ctx->d_cnt_bl, nb);
      ENSURE(ctx, ctx->d_base_bl, nb);
      if (ctx->d_cnt_bl.p != before || !ctx->cnt_bl_clean || ctx->cnt_bl_bytes != nb) {
        HIPCHK(ctx, hipMemsetAsync(ctx->d_cnt_bl.p, 0, ctx->d_cnt_bl.cap, st));
        ctx->cnt_bl_bytes = nb;
      }
      ctx->cnt_bl_clean = false;
    }
    const bool no_lds_table = getenv("LT_GEN_NO_LDS_TABLE") != nullptr;  // developer / test switch
    // LDS tables of k_gates: the neighbour's gate records (T2) and the image's own segments (T1), 80 B
    // per segment each; two workgroups per CU need both within 80 KB, one workgroup within 160 KB
    int lds_segs = (!no_lds_table && ctx->max_nb_segs <= 1024) ? ctx->max_nb_segs : 0;
    int lds_segs1 = (!no_lds_table && ctx->max_own_segs <= 1024) ? ctx->max_own_segs : 0;
    // both tables only while two workgroups still fit a CU (80 KB each): beyond that the own segments come
    // from L2 -- measured at 700 / 1000 segments per image: k_gates -16 % / -14 % against one workgroup per CU
    if (lds_segs + lds_segs1 > 1024) lds_segs1 = 0;
    {
      ENSURE(ctx, ctx->d_st_row, 8 * Pn);
      ENSURE(ctx, ctx->d_surv_count, 4 * (size_t)(n_slots_all + 1));
      if (!ctx->d_seg_gates.p) return fail(ctx, LT_ERR_STATE, "segment gate records missing (Init not run?)");
      launch_gen_split(st, ctx->n_blk, ctx->max_rows, gcfg, ctx->d_m_off.as<long long>(), ctx->d_m_pairs.as<int>(),
                       ctx->d_blk_img.as<int>(), ctx->d_blk_nb.as<int>(), ctx->d_blk_slot.as<int>(),
                       ctx->d_seg_off.as<long long>(), ctx->d_cams.as<Cam>(), ctx->d_segs.as<Seg>(),
                       ctx->d_pairs.as<PairRec>(), ctx->d_blk_line_base.as<long long>(), ctx->d_st_c.as<CRec>(),
                       ctx->d_st_l.as<double>(), ctx->d_st_key.as<unsigned>(), ctx->d_wave_count.as<unsigned>(),
                       fast ? ctx->d_cnt_bl.as<unsigned>() : nullptr, lds_segs, lds_segs1, ctx->d_st_row.p,
                       ctx->d_surv_count.as<unsigned>(), G, ctx->d_seg_gates.p, ctx->d_blkrec.p, fine_gen ? &ev[8] : nullptr,
                       vp_on ? ctx->d_seg_vp.as<double>() : nullptr,
                       vp_on ? ctx->d_seg_has_vp.as<unsigned char>() : nullptr,
                       pts_on ? ctx->d_seg_pt_off.as<long long>() : nullptr, pts_on ? ctx->d_seg_pts.p : nullptr,
                       (pts_on && ctx->sfm_given) ? ctx->d_sfm_xyz.as<double>() : nullptr, ctx->d_err.as<int>(),
                       many_on ? 1 : 0, one_on ? 1 : 0, mult);
    }
    // with the per-kernel events on, the one after k_tri_rows also ends the generation stage
    if (fine_gen && ctx->n_blk > 0 && ctx->max_rows > 0) ev_gen_end = 10;
    else HIPCHK(ctx, hipEventRecord(ev[3], st));
    long long *hC = hp;  // this set's slot 0
    long long hC_fallback = 0;
    if (!hC) hC = &hC_fallback;
    if (fast) {
      // rows of every block are sorted by line id: sort-free placement
      launch_node_prefix(st, G, ctx->d_node_img.as<int>(), ctx->d_seg_off.as<long long>(),
                         ctx->d_nb_off.as<long long>(), ctx->d_blk_line_base.as<long long>(),
                         ctx->d_cnt_bl.as<unsigned>(), ctx->d_base_bl.as<unsigned>(), ctx->d_ntris_u.as<unsigned>(),
                         ctx->d_tri_off.as<long long>(), ctx->d_scan_status.as<unsigned long long>(),
                         ctx->d_err.as<int>());  // tri_off = exclusive scan of the counts, in the same kernel
      ctx->cnt_bl_clean = true;
      // Nothing below needs the candidate count on the host (the kernels read tri_off[G]; the grids of
      // k_place / k_score3 do not depend on it) except the SIZE of the compact arrays.  While the trivial
      // bound -- one candidate per staging slot -- fits kCountFreeBytes, the arrays get that size and the
      // whole run is enqueued without a host round trip (the count then arrives with the error flag);
      // otherwise (or with LT_TEST_SYNC_COUNT) one 8-byte copy + stream sync fetches the exact count.
      const long long bound = P * (long long)mult;
      constexpr long long kCountFreeBytes = 8ll << 30;
      const long long per_cand = (long long)(sizeof(CRec) + 8 + 8 + 4 + 4) + (long long)cand_meta_bytes();
      if (ctx->h_pinned && bound * per_cand <= kCountFreeBytes && !getenv("LT_TEST_SYNC_COUNT")) {
        C_known = -1;
        C_bound = bound;
      } else {
        HIPCHK(ctx, hipMemcpyAsync(hC, ctx->d_tri_off.as<long long>() + G, 8, hipMemcpyDeviceToHost, st));
        HIPCHK(ctx, hipStreamSynchronize(st));
        C_known = *hC;
      }
    } else {
      // generic rows: stable radix sort of the candidates by node (input is in row order)
      ENSURE(ctx, ctx->d_wave_pos, 8 * (size_t)(n_waves + 1));
      HIPCHK(ctx, hipMemsetAsync(ctx->d_wave_count.as<unsigned>() + n_waves, 0, 4, st));
      size_t tmp = scan_temp_bytes_u32_to_i64(n_waves + 1);
      ENSURE(ctx, ctx->d_scan_tmp, std::max<size_t>(tmp, 16));
      if (launch_scan_u32_to_i64(st, ctx->d_scan_tmp.p, tmp, n_waves + 1, ctx->d_wave_count.as<unsigned>(),
                                 ctx->d_wave_pos.as<long long>()) != 0)
        return fail(ctx, LT_ERR_HIP, "rocprim scan failed");
      HIPCHK(ctx, hipMemcpyAsync(hC, ctx->d_wave_pos.as<long long>() + n_waves, 8, hipMemcpyDeviceToHost, st));
      HIPCHK(ctx, hipStreamSynchronize(st));
      C_known = *hC;
    }
    // fast path: no record is moved -- k_place writes the permutation only
    const bool perm_mode = fast && !getenv("LT_TEST_PLACE_COPY");
    ctx->perm_mode = perm_mode;
    ctx->compact_valid = !perm_mode;
    if (C_known < 0) {
      // the bound is generous: if the device cannot give that much, fetch the exact count after all
      const size_t Bn = (size_t)std::max<long long>(C_bound, 1);
      const bool got = (perm_mode ? ctx->d_place_perm.ensure(4 * Bn)
                                  : (ctx->d_cand.ensure(sizeof(CRec) * Bn) && ctx->d_lite.ensure(sizeof(double) * Bn))) &&
                       ctx->d_score.ensure(8 * Bn) && ctx->d_edge_flag.ensure(4 * Bn) && ctx->d_cand_node.ensure(4 * Bn) &&
                       ctx->d_cand_meta.ensure(cand_meta_bytes() * Bn);
      if (!got) {
        (void)hipGetLastError();
        HIPCHK(ctx, hipMemcpyAsync(hC, ctx->d_tri_off.as<long long>() + G, 8, hipMemcpyDeviceToHost, st));
        HIPCHK(ctx, hipStreamSynchronize(st));
        C_known = *hC;
      }
    }
    if (C_known >= 0) C_bound = C_known;
    const size_t Cn = (size_t)std::max<long long>(C_bound, 1);
    if (perm_mode) {
      ENSURE(ctx, ctx->d_place_perm, 4 * Cn);
    } else {
      ENSURE(ctx, ctx->d_cand, sizeof(CRec) * Cn); ENSURE(ctx, ctx->d_lite, sizeof(double) * Cn);
    }
    ENSURE(ctx, ctx->d_score, 8 * Cn); ENSURE(ctx, ctx->d_edge_flag, 4 * Cn); ENSURE(ctx, ctx->d_cand_node, 4 * Cn);
    ctx->cand_cap = (long long)Cn;
    if (fast) {
      launch_place(st, ctx->n_blk, ctx->max_rows, ctx->d_m_off.as<long long>(), ctx->d_blk_img.as<int>(),
                   ctx->d_seg_off.as<long long>(), ctx->d_blk_line_base.as<long long>(),
                   ctx->d_base_bl.as<unsigned>(), ctx->d_wave_count.as<unsigned>(), ctx->d_tri_off.as<long long>(),
                   ctx->d_st_c.as<CRec>(), ctx->d_st_l.as<double>(), ctx->d_st_key.as<unsigned>(),
                   ctx->d_cand.as<CRec>(), ctx->d_lite.as<double>(), ctx->d_cand_node.as<unsigned>(), mult,
                   perm_mode ? ctx->d_place_perm.as<unsigned>() : nullptr);
    } else {
      ENSURE(ctx, ctx->d_keys, 4 * Cn); ENSURE(ctx, ctx->d_rows, 4 * Cn);
      ENSURE(ctx, ctx->d_skeys, 4 * Cn); ENSURE(ctx, ctx->d_srows, 4 * Cn);
      launch_pack_keys(st, ctx->n_blk, ctx->max_rows, ctx->d_m_off.as<long long>(),
                       ctx->d_wave_count.as<unsigned>(), ctx->d_wave_pos.as<long long>(),
                       ctx->d_st_key.as<unsigned>(), ctx->d_keys.as<unsigned>(), ctx->d_rows.as<unsigned>(), mult);
      if (C_known > 0) {
        int end_bit = bits_for(G + 1);
        size_t tmp = sort_temp_bytes(C_known, end_bit);
        ENSURE(ctx, ctx->d_sort_tmp, std::max<size_t>(tmp, 16));
        if (launch_sort(st, ctx->d_sort_tmp.p, tmp, C_known, ctx->d_keys.as<unsigned>(), ctx->d_skeys.as<unsigned>(),
                        ctx->d_rows.as<unsigned>(), ctx->d_srows.as<unsigned>(), end_bit) != 0)
          return fail(ctx, LT_ERR_HIP, "rocprim radix sort failed");
      }
      launch_node_offsets(st, C_known, G, ctx->d_skeys.as<unsigned>(), ctx->d_tri_off.as<long long>());
      launch_permute(st, C_known, ctx->d_skeys.as<unsigned>(), ctx->d_srows.as<unsigned>(), ctx->d_st_c.as<CRec>(),
                     ctx->d_st_l.as<double>(), ctx->d_cand.as<CRec>(), ctx->d_lite.as<double>(),
                     ctx->d_cand_node.as<unsigned>());
    }
    // ... and the one in front of k_score3 ends the placement stage (it then includes k_cand_meta)
    if (fine_score && C_bound > 0) ev_place_end = 11;
    else HIPCHK(ctx, hipEventRecord(ev[4], st));
  } else if (ctx->job_mode == 2) {
    ctx->perm_mode = false;
    ctx->compact_valid = true;
    const size_t In = (size_t)std::max<long long>(P, 1);
    // VP-guided proposals: three survivor ballots per work item (algebraic, vp of l1, vp of l2)
    const bool vp_on = ctx->cfg.use_vp && !ctx->cfg.disable_vp_triangulation;
    if (vp_on && !ctx->vp_ready) return fail(ctx, LT_ERR_STATE, "use_vp is set but InitVPResults was not called");
    const double *seg_vp = vp_on ? ctx->d_seg_vp.as<double>() : nullptr;
    const unsigned char *seg_has_vp = vp_on ? ctx->d_seg_has_vp.as<unsigned char>() : nullptr;
    const int n_masks = vp_on ? 3 : 1;
    // point-guided proposals: a variable number of candidates per connection -> per-connection counts
    // (one byte each) instead of ballots, see k_gen_exhaustive_pts
    const bool pts_any = ctx->pts_ready && (!ctx->cfg.disable_many_points_triangulation || !ctx->cfg.disable_one_point_triangulation);
    const int many_on = (pts_any && !ctx->cfg.disable_many_points_triangulation) ? 1 : 0;
    const int one_on = (pts_any && !ctx->cfg.disable_one_point_triangulation) ? 1 : 0;
    const double *sfm_xyz = (pts_any && ctx->sfm_given) ? ctx->d_sfm_xyz.as<double>() : nullptr;
    // (+ one ballot word: the plain mode scans the popcounts of the ballots directly, the word behind the last is 0)
    ENSURE(ctx, ctx->d_masks, pts_any ? 64 * In : 8 * (In * n_masks + 1)); ENSURE(ctx, ctx->d_mask_cnt, 4 * (In + 1));
    ENSURE(ctx, ctx->d_mask_pos, 8 * (In + 1));
    // Plain exhaustive mode (no VP / point proposals): pass 1 with the neighbour lines held in registers (k_gates_ex;
    // LT_TEST_EX_PASS1_BLOCK keeps the wave-per-(node, neighbour) form the VP variant uses), and, while the staging
    // capacity holds, in its ONE-PASS form: pass 1 only lists the connections that pass the cheap gates (k_gates_ex<true>),
    // k_tri_ex evaluates the list densely and writes the survivors to staging slots, a permutation orders them -- no
    // second triangulation pass, no host round trip for the candidate count.  The capacity is a
    // fraction of the connections (1/6 until a run of this context has measured its need, then 1.4 x that); a run that
    // overflows it (device error flag 5) is repeated in the two-pass form by finish_run.  LT_TEST_EX_TWO_PASS: always
    // two passes.
    const bool plain = !pts_any && !vp_on;
    bool staged = plain && ctx->h_pinned && !ctx->ex_two_pass && P > 0 && !getenv("LT_TEST_EX_TWO_PASS") &&
                  !getenv("LT_TEST_EX_PASS1_BLOCK");
    long long ex_cap = 0;
    unsigned region_cap = 0;
    unsigned long long *ex_ctr = ctx->d_scan_status.as<unsigned long long>() + n_status_scan + score3_tile_buckets() * 16;
    if (staged) {
      double frac = ctx->ex_frac > 0.0 ? ctx->ex_frac : 1.0 / 6.0;
      long long slack = 65536;
      if (const char *f = getenv("LT_TEST_EX_CAP_FRAC")) {  // test switch: force a (too small) capacity
        frac = atof(f);
        slack = 0;
      }
      const long long want = (long long)((double)ctx->n_conn * frac) + slack;
      const long long nreg = ex_regions();
      const long long rc8 = ((want + nreg - 1) / nreg + 63) & ~63ll;
      ex_cap = nreg * rc8;
      // ~210 bytes per slot over all arrays: beyond 64 GB (or the 32-bit slot index) the two-pass form, whose arrays
      // have the exact size
      if (ex_cap >= (1ll << 32) - 1 || ex_cap * 210 > (64ll << 30)) staged = false;
      else {
        region_cap = (unsigned)rc8;
        const size_t Bn = (size_t)ex_cap;
        const bool got = ctx->d_st_c.ensure(sizeof(CRec) * Bn) && ctx->d_st_l.ensure(sizeof(double) * Bn) &&
                         ctx->d_st_key.ensure(4 * Bn) && ctx->d_place_perm.ensure(4 * Bn) && ctx->d_score.ensure(8 * Bn) &&
                         ctx->d_edge_flag.ensure(4 * Bn) && ctx->d_cand_node.ensure(4 * Bn) &&
                         ctx->d_cand_meta.ensure(cand_meta_bytes() * Bn) && ctx->d_ex_rec.ensure(4 * Bn) &&
                         ctx->d_ex_ent.ensure(8 * Bn) && ctx->d_ex_z.ensure(4 * Bn);
        if (!got) {
          (void)hipGetLastError();
          staged = false;
        }
      }
    }
    ctx->ex_staged_set[set] = staged;
    if (pts_any) {
      launch_gen_exhaustive_pts(st, false, P, gcfg, ctx->d_item_off.as<long long>(), G, ctx->d_node_img.as<int>(),
                                ctx->d_nb_off.as<long long>(), ctx->d_blk_nb.as<int>(), ctx->d_seg_off.as<long long>(),
                                ctx->d_cams.as<Cam>(), ctx->d_segs.as<Seg>(), ctx->d_pairs.as<PairRec>(),
                                ctx->d_masks.as<unsigned char>(), ctx->d_mask_cnt.as<unsigned>(), nullptr, nullptr,
                                nullptr, seg_vp, seg_has_vp, ctx->d_seg_pt_off.as<long long>(), ctx->d_seg_pts.p,
                                sfm_xyz, ctx->d_err.as<int>(), many_on, one_on, ctx->d_blk_chunk_off.as<int>(),
                                ctx->max_nb, ctx->max_chunks, ctx->d_seg_gates.p);
    } else {
      if (plain && !getenv("LT_TEST_EX_PASS1_BLOCK"))
        launch_gates_exhaustive(st, ctx->n_blk, ctx->max_chunks, P, gcfg, ctx->d_item_off.as<long long>(),
                                ctx->d_blk_img.as<int>(), ctx->d_blk_nb.as<int>(), ctx->d_seg_off.as<long long>(),
                                ctx->d_cams.as<Cam>(), ctx->d_segs.as<Seg>(), ctx->d_pairs.as<PairRec>(),
                                ctx->d_masks.as<unsigned long long>(), ctx->d_blk_chunk_off.as<int>(), ctx->d_seg_gates.p,
                                staged ? ctx->d_ex_ent.as<unsigned long long>() : nullptr, ex_ctr, region_cap,
                                ctx->d_err.as<int>());
      else
      launch_gen_exhaustive(st, false, P, gcfg, ctx->d_item_off.as<long long>(), G, ctx->d_node_img.as<int>(),
                            ctx->d_nb_off.as<long long>(), ctx->d_blk_nb.as<int>(), ctx->d_seg_off.as<long long>(),
                            ctx->d_cams.as<Cam>(), ctx->d_segs.as<Seg>(), ctx->d_pairs.as<PairRec>(),
                            ctx->d_masks.as<unsigned long long>(), nullptr, nullptr, nullptr, seg_vp, seg_has_vp,
                            ctx->d_blk_chunk_off.as<int>(), ctx->max_nb, ctx->max_chunks, ctx->d_seg_gates.p);
      if (staged) {
        HIPCHK(ctx, hipMemsetAsync(ctx->d_masks.p, 0, 8 * In, st));
        launch_tri_exhaustive(st, ctx->d_ex_ent.as<unsigned long long>(), ex_ctr, region_cap, gcfg, P,
                              ctx->d_item_off.as<long long>(), ctx->d_blk_img.as<int>(), ctx->d_blk_nb.as<int>(),
                              ctx->d_nb_off.as<long long>(), ctx->d_seg_off.as<long long>(), ctx->d_cams.as<Cam>(),
                              ctx->d_segs.as<Seg>(), ctx->d_pairs.as<PairRec>(), ctx->d_blk_chunk_off.as<int>(),
                              ctx->d_masks.as<unsigned long long>(), ctx->d_st_c.as<CRec>(), ctx->d_st_l.as<double>(),
                              ctx->d_st_key.as<unsigned>(), ctx->d_ex_z.as<float>());
      }
      if (!plain) launch_popc(st, P, ctx->d_masks.as<unsigned long long>(), ctx->d_mask_cnt.as<unsigned>(), n_masks);
    }
    if (plain) {
      // one ballot per item: the scan reads the ballots through a popcount iterator (no count pass, no count array)
      HIPCHK(ctx, hipMemsetAsync(ctx->d_masks.as<unsigned long long>() + P, 0, 8, st));
      size_t tmp = scan_temp_bytes_popc(P + 1);
      ENSURE(ctx, ctx->d_scan_tmp, std::max<size_t>(tmp, 16));
      if (launch_scan_popc(st, ctx->d_scan_tmp.p, tmp, P + 1, ctx->d_masks.as<unsigned long long>(),
                           ctx->d_mask_pos.as<long long>()) != 0)
        return fail(ctx, LT_ERR_HIP, "rocprim scan failed");
    } else {
      HIPCHK(ctx, hipMemsetAsync(ctx->d_mask_cnt.as<unsigned>() + P, 0, 4, st));
      size_t tmp = scan_temp_bytes_u32_to_i64(P + 1);
      ENSURE(ctx, ctx->d_scan_tmp, std::max<size_t>(tmp, 16));
      if (launch_scan_u32_to_i64(st, ctx->d_scan_tmp.p, tmp, P + 1, ctx->d_mask_cnt.as<unsigned>(),
                                 ctx->d_mask_pos.as<long long>()) != 0)
        return fail(ctx, LT_ERR_HIP, "rocprim scan failed");
    }
    if (staged) {
      launch_tri_offsets_ex(st, G, ctx->d_item_off.as<long long>(), ctx->d_mask_pos.as<long long>(), P, -1, ex_cap,
                            ctx->d_tri_off.as<long long>(), ctx->d_err.as<int>());
      HIPCHK(ctx, hipEventRecord(ev[3], st));
      launch_place_exhaustive(st, ex_ctr, region_cap, ctx->d_ex_ent.as<unsigned long long>(),
                              ctx->d_st_key.as<unsigned>(), ctx->d_item_off.as<long long>(),
                              ctx->d_blk_chunk_off.as<int>(), ctx->d_masks.as<unsigned long long>(),
                              ctx->d_mask_pos.as<long long>(), P, ctx->d_tri_off.as<long long>(), G,
                              ctx->d_place_perm.as<unsigned>(), ctx->d_result3.as<long long>() + 3);
      ctx->ex_region_cap = region_cap;
      launch_cand_node(st, G, ctx->d_tri_off.as<long long>(), ctx->d_cand_node.as<unsigned>());
      ctx->perm_mode = true;
      ctx->compact_valid = false;
      ctx->cand_cap = ex_cap;
      C_known = -1;
      C_bound = ex_cap;
      HIPCHK(ctx, hipEventRecord(ev[4], st));
    } else {
    // the candidate count sizes the compacted arrays: one small host round trip
    long long total = 0;
    HIPCHK(ctx, hipMemcpyAsync(&total, ctx->d_mask_pos.as<long long>() + P, 8, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipStreamSynchronize(st));
    HIPCHK(ctx, hipEventRecord(ev[3], st));
    const size_t Cn = (size_t)std::max<long long>(total, 1);
    ENSURE(ctx, ctx->d_cand, sizeof(CRec) * Cn); ENSURE(ctx, ctx->d_lite, sizeof(double) * Cn);
    ENSURE(ctx, ctx->d_score, 8 * Cn); ENSURE(ctx, ctx->d_edge_flag, 4 * Cn);
    ctx->cand_cap = (long long)Cn;
    if (pts_any)
      launch_gen_exhaustive_pts(st, true, P, gcfg, ctx->d_item_off.as<long long>(), G, ctx->d_node_img.as<int>(),
                                ctx->d_nb_off.as<long long>(), ctx->d_blk_nb.as<int>(), ctx->d_seg_off.as<long long>(),
                                ctx->d_cams.as<Cam>(), ctx->d_segs.as<Seg>(), ctx->d_pairs.as<PairRec>(),
                                ctx->d_masks.as<unsigned char>(), ctx->d_mask_cnt.as<unsigned>(),
                                ctx->d_mask_pos.as<long long>(), ctx->d_cand.as<CRec>(), ctx->d_lite.as<double>(),
                                seg_vp, seg_has_vp, ctx->d_seg_pt_off.as<long long>(), ctx->d_seg_pts.p, sfm_xyz,
                                ctx->d_err.as<int>(), many_on, one_on, ctx->d_blk_chunk_off.as<int>(), ctx->max_nb,
                                ctx->max_chunks, ctx->d_seg_gates.p);
    else if (!vp_on && !getenv("LT_TEST_EX_PASS2_BLOCK"))
      launch_fill_exhaustive(st, ctx->n_blk, P, gcfg, ctx->d_item_off.as<long long>(), ctx->d_blk_img.as<int>(),
                             ctx->d_blk_nb.as<int>(), ctx->d_nb_off.as<long long>(), ctx->d_seg_off.as<long long>(),
                             ctx->d_cams.as<Cam>(), ctx->d_segs.as<Seg>(), ctx->d_pairs.as<PairRec>(),
                             ctx->d_masks.as<unsigned long long>(), ctx->d_mask_pos.as<long long>(),
                             ctx->d_cand.as<CRec>(), ctx->d_lite.as<double>(), ctx->d_blk_chunk_off.as<int>());
    else
      launch_gen_exhaustive(st, true, P, gcfg, ctx->d_item_off.as<long long>(), G, ctx->d_node_img.as<int>(),
                            ctx->d_nb_off.as<long long>(), ctx->d_blk_nb.as<int>(), ctx->d_seg_off.as<long long>(),
                            ctx->d_cams.as<Cam>(), ctx->d_segs.as<Seg>(), ctx->d_pairs.as<PairRec>(),
                            ctx->d_masks.as<unsigned long long>(), ctx->d_mask_pos.as<long long>(),
                            ctx->d_cand.as<CRec>(), ctx->d_lite.as<double>(), seg_vp, seg_has_vp,
                            ctx->d_blk_chunk_off.as<int>(), ctx->max_nb, ctx->max_chunks, ctx->d_seg_gates.p);
    launch_tri_offsets_ex(st, G, ctx->d_item_off.as<long long>(), ctx->d_mask_pos.as<long long>(), P, total, -1,
                          ctx->d_tri_off.as<long long>(), ctx->d_err.as<int>());
    ENSURE(ctx, ctx->d_cand_node, 4 * Cn);
    launch_cand_node(st, G, ctx->d_tri_off.as<long long>(), ctx->d_cand_node.as<unsigned>());
    C_known = total;
    C_bound = total;
    HIPCHK(ctx, hipEventRecord(ev[4], st));
    }
  } else {
    ctx->perm_mode = false;
    ctx->compact_valid = true;
    HIPCHK(ctx, hipMemsetAsync(ctx->d_tri_off.p, 0, sizeof(long long) * (size_t)(G + 1), st));
    ENSURE(ctx, ctx->d_cand, sizeof(CRec)); ENSURE(ctx, ctx->d_lite, sizeof(double));
    ENSURE(ctx, ctx->d_score, 8); ENSURE(ctx, ctx->d_edge_flag, 4); ENSURE(ctx, ctx->d_cand_node, 4);
    C_known = 0;
    C_bound = 0;
    for (int k = 3; k <= 4; ++k) HIPCHK(ctx, hipEventRecord(ev[k], st));
  }

  // ---- scoring ----
  // LT_TEST_SCORE_F64: the sweep's early exit in double precision (the default is the bounded single-precision form)
  const bool score_f32 = !getenv("LT_TEST_SCORE_F64");
  if (score3_lds_bytes(ctx->max_nb, score_f32) > 160 * 1024)
    return fail(ctx, LT_ERR_ARGUMENT, "too many neighbours for the scoring kernel's LDS budget");
  {
    // conservative square of the scale-invariant endpoint gate (see k_score3)
    double th = scfg.l3.th_scaleinv * (1.0 + 1e-6);
    double guard2 = (scfg.l3.th_scaleinv > 0.0 && scfg.l3.score_th > 0.0 && scfg.l3.score_th < 1.0) ? th * th : 1e300;
    if (getenv("LT_TEST_NO_SCORE_GUARDS")) guard2 = 1e300;
    if (ctx->h_nb_off[ctx->n_img] >= (1ll << 24))
      return fail(ctx, LT_ERR_ARGUMENT, "too many (image, neighbour) blocks in one batch (>= 2^24)");
    ENSURE(ctx, ctx->d_cand_meta, cand_meta_bytes() * (size_t)std::max<long long>(C_bound, 1));
    ENSURE(ctx, ctx->d_tile_order, 64 * 128);  // the tile draw counters of k_score3 (8 x 128 B)
    // tiles listed by cost class (LT_TEST_NO_TILE_CLASSES: natural tile order)
    // (matched mode only: the wide nodes of the exhaustive mode put every tile into the top class, whose one counter
    // per queue then serialises ~4e4 appends -- k_cand_meta 0.11 -> 0.50 ms -- for an order that changes nothing)
    const bool tile_classes = !getenv("LT_TEST_NO_TILE_CLASSES") && ctx->job_mode == 1;
    const unsigned tile_cap = (unsigned)(((std::max<long long>(C_bound, 1) + 63) / 64 + 7) / 8);  // tiles of one draw queue
    if (tile_classes) ENSURE(ctx, ctx->d_tile_list, 4 * (size_t)tile_cap * (size_t)score3_tile_buckets());
    // large nodes (exhaustive matching): depth-sorted sweep, see k_depth_order
    const bool score_sorted = score_f32 && ctx->job_mode == 2 && !getenv("LT_TEST_SCORE_UNSORTED");
    if (score_sorted) {
      ENSURE(ctx, ctx->d_perm, 4 * (size_t)std::max<long long>(C_bound, 1));
      ENSURE(ctx, ctx->d_rng, 8 * (size_t)std::max<long long>(C_bound, 1));
    }
    // depth-sorted sweep over the staged records of the one-pass exhaustive mode: see k_depth_order
    const bool staged_sorted = score_sorted && ctx->perm_mode && ctx->job_mode == 2;
    C_run = C_bound;  // replaced by the exact count when that arrives with the error flag (finish_run)
    // LT_SCORE_SPLIT: the three-kernel scoring (k_sweep6 / k_eval6 / k_reduce6) and its pair list -- capacity from the
    // candidate bound; a run that overflows it (device flag 7) is repeated with the fused kernel by finish_run
    unsigned split_region_cap = 0;
    long long split_tiles_b = 0;  // d_split_head = [tiles] window bounds (8 B) | [tiles] newest segment + count (8 B)
    const bool split = score_f32 && !score_sorted && !ctx->score_split_off && getenv("LT_SCORE_SPLIT") != nullptr;
    if (split) {
      const long long tiles_b = (std::max<long long>(C_bound, 1) + 63) / 64;
      split_tiles_b = tiles_b;
      long long pc = std::min<long long>(std::max<long long>(2 * C_bound, 4ll << 20), (1ll << 31) - 64);
      if (const char *e = getenv("LT_TEST_SPLIT_PAIR_CAP")) pc = std::max<long long>(64, atoll(e));  // test: force an overflow
      // 64 regions (one bump counter each); a record is a pair or a segment header
      const long long rcap = (pc + tiles_b * 2 + 63) / 64;
      if (ctx->d_split_pairs.ensure(16 * (size_t)rcap * 64) && ctx->d_split_head.ensure(16 * (size_t)tiles_b + 16) &&
          ctx->d_split_tot.ensure(64 * 128)) {
        split_region_cap = (unsigned)rcap;
      } else {
        (void)hipGetLastError();
      }
    }
    launch_score3(st, C_bound, G, ctx->d_tri_off.as<long long>(), ctx->d_cand_node.as<unsigned>(), ctx->d_cand_meta.p,
                  ctx->perm_mode ? ctx->d_st_c.as<CRec>() : ctx->d_cand.as<CRec>(), ctx->d_node_img.as<int>(),
                  ctx->d_nb_off.as<long long>(), ctx->d_blk_order.as<int>(), ctx->d_cams.as<Cam>(),
                  ctx->d_score.as<double>(), ctx->d_pair_counter.as<unsigned long long>(), ctx->max_nb, scfg,
                  guard2, fine_score ? ev[11] : nullptr, ctx->d_tile_order.as<unsigned>(), score_f32,
                  (ctx->perm_mode && !staged_sorted) ? ctx->d_place_perm.as<unsigned>()
                                                     : (score_sorted ? ctx->d_perm.as<unsigned>() : nullptr),
                  score_sorted ? ctx->d_rng.p : nullptr, ctx->perm_mode && !staged_sorted,
                  (ctx->exp_tile_order_C == C_bound || ctx->exp_tile_order_C == ctx->C_last) && ctx->exp_tile_order_C > 0
                      ? ctx->d_exp_tile_order.as<unsigned>() : nullptr,
                  tile_classes ? (unsigned *)(ctx->d_scan_status.as<unsigned long long>() + n_status_scan) : nullptr,
                  tile_classes ? ctx->d_tile_list.as<unsigned>() : nullptr, tile_cap,
                  staged_sorted ? ctx->d_place_perm.as<unsigned>() : nullptr,
                  staged_sorted ? ctx->d_ex_rec.as<unsigned>() : nullptr,
                  staged_sorted ? ctx->d_ex_z.as<float>() : nullptr, ctx->d_err.as<int>(),
                  split_region_cap ? ctx->d_split_pairs.p : nullptr,
                  split_region_cap ? (unsigned *)((char *)ctx->d_split_head.p + 8 * (size_t)split_tiles_b) : nullptr,
                  split_region_cap ? ctx->d_split_tot.as<unsigned>() : nullptr, split_region_cap,
                  split_region_cap ? ctx->d_split_head.p : nullptr);
  }
  HIPCHK(ctx, hipEventRecord(ev[5], st));
  ENSURE(ctx, ctx->d_best_idx, 8 * (size_t)std::max<long long>(G, 1));
  ENSURE(ctx, ctx->d_nvalid, 4 * (size_t)(G + 1));
  ENSURE(ctx, ctx->d_edge_off, 8 * (size_t)(G + 1));
  ENSURE(ctx, ctx->d_best_c, sizeof(Cand) * (size_t)std::max<long long>(G, 1));
  ENSURE(ctx, ctx->d_best_score, 8 * (size_t)std::max<long long>(G, 1));
  ENSURE(ctx, ctx->d_best_src, 8 * (size_t)std::max<long long>(G, 1));
  ENSURE(ctx, ctx->d_ntris, 4 * (size_t)std::max<long long>(G, 1));
  // per node: best candidate (gathered into the dense per-node arrays by the same kernel), valid-edge
  // flags and their number; the edge offsets (a scan) and the edge lists are produced at download time
  launch_select(st, G, ctx->d_tri_off.as<long long>(), ctx->d_score.as<double>(), scfg.fullscore_th,
                scfg.max_valid_conns, ctx->d_best_idx.as<long long>(), ctx->d_edge_flag.as<unsigned>(),
                ctx->d_nvalid.as<unsigned>(), ctx->perm_mode ? ctx->d_st_c.as<CRec>() : ctx->d_cand.as<CRec>(),
                ctx->perm_mode ? ctx->d_st_l.as<double>() : ctx->d_lite.as<double>(),
                ctx->d_best_c.as<Cand>(), ctx->d_best_score.as<double>(), ctx->d_best_src.as<int>(),
                ctx->d_ntris.as<int>(), /*wide=*/ctx->job_mode == 2, ctx->d_err.as<int>(),
                ctx->d_pair_counter.as<unsigned long long>(), ctx->d_result3.as<long long>(),
                ctx->perm_mode ? ctx->d_place_perm.as<unsigned>() : nullptr);
  if (!hp) HIPCHK(ctx, hipEventRecord(ev[7], st));  // with result slots the end marker below also ends the run
  HIPCHK(ctx, hipGetLastError());
  // the device error flag, the candidate count and the pair statistic ride on the stream into this set's
  // pinned slots; finish_run reads them behind the end marker
  if (hp) {
    // hp[0] candidate count, hp[1] error flag, hp[2] pair statistic: one record, gathered by k_select
    if (G <= 0) HIPCHK(ctx, hipMemsetAsync(ctx->d_result3.p, 0, 32, st));  // no nodes: k_select did not run
    // hp[3]: fullest staging region of the one-pass exhaustive mode (k_place_ex)
    HIPCHK(ctx, hipMemcpyAsync(&hp[0], ctx->d_result3.p, 32, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipEventRecord(ev[12], st));
  }
  int rc_prev = LT_OK;
  if (ctx->run_pending) {  // the previous run (the other set)
    ctx->in_run_async = true;
    rc_prev = finish_run(ctx);
    ctx->in_run_async = false;
  }
  ctx->run_pending = true;
  ctx->pend_set = set;
  ctx->pend_count_on_device = C_known < 0;
  ctx->pend_fine_gen = fine_gen;
  ctx->pend_fine_score = fine_score;
  ctx->pend_C = C_run;
  ctx->pend_ev_gen_end = ev_gen_end;
  ctx->pend_ev_place_end = ev_place_end;
  ctx->ran = true;
  ctx->downloaded = false;
  ctx->host_view_valid = false;
  return rc_prev;
}

int lt_run_device(lt_ctx *ctx) {
  int rc = lt_run_device_async(ctx);
  if (rc) return rc;
  return finish_run(ctx);
}

// Images that have no results (yet) hold a value-initialised best candidate, like the reference's TriTuple.
static void define_best_of_other_images(lt_ctx *ctx) {
  for (int i = 0; i < ctx->n_img; ++i)
    if (!ctx->best_c_set[(size_t)i]) {
      const long long a = ctx->seg_off[i], b = ctx->seg_off[i + 1];
      if (b > a) std::memset((void *)(ctx->best_c + a), 0, sizeof(Cand) * (size_t)(b - a));
      ctx->best_c_set[(size_t)i] = 1;
    }
}

// The split host-side view (Cand / CandLite in candidate order) of the last run's candidates, for the debug
// read-outs: converted on demand from the 128-byte device records -- through the placement permutation when the
// records are still in the staging lists.
static int materialize_compact(lt_ctx *ctx) {
  if (ctx->host_view_valid) return LT_OK;
  const long long C = ctx->C_last;
  const size_t Cn = (size_t)std::max<long long>(C, 1);
  ENSURE(ctx, ctx->d_hcand, sizeof(Cand) * Cn); ENSURE(ctx, ctx->d_hlite, sizeof(CandLite) * Cn);
  launch_host_view(ctx->stream, C, ctx->perm_mode ? ctx->d_place_perm.as<unsigned>() : nullptr,
                   ctx->perm_mode ? ctx->d_st_c.as<CRec>() : ctx->d_cand.as<CRec>(),
                   ctx->perm_mode ? ctx->d_st_l.as<double>() : ctx->d_lite.as<double>(), ctx->d_hcand.as<Cand>(),
                   ctx->d_hlite.as<CandLite>());
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  ctx->host_view_valid = true;
  return LT_OK;
}

int lt_download(lt_ctx *ctx) {
  LT_RANGE("lt_download (per-node results -> host)");
  LT_FINISH(ctx);
  if (!ctx->ran) return fail(ctx, LT_ERR_STATE, "lt_download before lt_run_device");
  if (ctx->downloaded) return LT_OK;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  double t0 = now_ms();
  hipStream_t st = ctx->stream;
  const long long G = ctx->G;
  // edges need their final positions: offsets (scan of the per-node counts) and lists are made now
  HIPCHK(ctx, hipMemsetAsync(ctx->d_nvalid.as<unsigned>() + G, 0, 4, st));
  {
    size_t tmp = scan_temp_bytes_u32_to_i64(G + 1);
    ENSURE(ctx, ctx->d_scan_tmp, std::max<size_t>(tmp, 16));
    if (launch_scan_u32_to_i64(st, ctx->d_scan_tmp.p, tmp, G + 1, ctx->d_nvalid.as<unsigned>(),
                               ctx->d_edge_off.as<long long>()) != 0)
      return fail(ctx, LT_ERR_HIP, "rocprim scan failed");
  }
  std::vector<long long> tri_off(G + 1), edge_off(G + 1);
  HIPCHK(ctx, hipMemcpyAsync(tri_off.data(), ctx->d_tri_off.p, 8 * (size_t)(G + 1), hipMemcpyDeviceToHost, st));
  HIPCHK(ctx, hipMemcpyAsync(edge_off.data(), ctx->d_edge_off.p, 8 * (size_t)(G + 1), hipMemcpyDeviceToHost, st));
  HIPCHK(ctx, hipStreamSynchronize(st));
  ctx->C = tri_off[G];
  ctx->E = edge_off[G];
  ENSURE(ctx, ctx->d_edges, 8 * (size_t)std::max<long long>(ctx->E, 1));
  launch_edge_fill(st, G, ctx->d_tri_off.as<long long>(), ctx->d_edge_flag.as<unsigned>(),
                   ctx->d_edge_off.as<long long>(), ctx->perm_mode ? ctx->d_st_c.as<CRec>() : ctx->d_cand.as<CRec>(),
                   ctx->d_edges.as<int>(), ctx->perm_mode ? ctx->d_place_perm.as<unsigned>() : nullptr);
  // one pooled page-locked block for all result arrays
  const size_t Gn = (size_t)std::max<long long>(G, 1), En = (size_t)std::max<long long>(ctx->E, 1);
  const size_t o_bc = 0, o_bs = o_bc + sizeof(Cand) * Gn, o_src = o_bs + 8 * Gn, o_nt = o_src + 8 * Gn,
               o_ed = (o_nt + 4 * Gn + 15) / 16 * 16, total = o_ed + 8 * En;
  lt_host::HostBlock hb = lt_host::host_block_acquire(total);
  if (!hb.p) return fail(ctx, LT_ERR_RUNTIME, "out of host memory for the results");
  struct Rel {
    lt_host::HostBlock b;
    ~Rel() { lt_host::host_block_release(b); }
  } rel{hb};
  char *base = (char *)hb.p;
  const Cand *bc = (const Cand *)(base + o_bc);
  const double *bs = (const double *)(base + o_bs);
  const int *bsrc = (const int *)(base + o_src);
  const int *nt = (const int *)(base + o_nt);
  const int *edges = (const int *)(base + o_ed);
  if (G > 0) {
    HIPCHK(ctx, hipMemcpyAsync(base + o_bc, ctx->d_best_c.p, sizeof(Cand) * (size_t)G, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipMemcpyAsync(base + o_bs, ctx->d_best_score.p, 8 * (size_t)G, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipMemcpyAsync(base + o_src, ctx->d_best_src.p, 8 * (size_t)G, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipMemcpyAsync(base + o_nt, ctx->d_ntris.p, 4 * (size_t)G, hipMemcpyDeviceToHost, st));
  }
  if (ctx->E > 0)
    HIPCHK(ctx, hipMemcpyAsync(base + o_ed, ctx->d_edges.p, 8 * (size_t)ctx->E, hipMemcpyDeviceToHost, st));
  HIPCHK(ctx, hipStreamSynchronize(st));
  // merge the nodes of the job's images into the persistent per-node results; the edge lists of the
  // whole run are appended to the pool in one piece (nodes outside the job have none)
  const long long pool_base = (long long)ctx->valid_edges.pool.size();
  ctx->valid_edges.pool.insert(ctx->valid_edges.pool.end(), edges, edges + 2 * (size_t)ctx->E);
  long long pairs = 0;
  const long long n_job = (long long)ctx->job_imgs.size();
#pragma omp parallel for num_threads(lt::host_threads()) schedule(dynamic, 4) reduction(+ : pairs)
  for (long long j = 0; j < n_job; ++j) {
    const int idx = ctx->job_imgs[(size_t)j];
    ctx->best_c_set[(size_t)idx] = 1;
    for (long long g = ctx->seg_off[idx]; g < ctx->seg_off[idx + 1]; ++g) {
      ctx->n_tris[g] = nt[g];
      pairs += (long long)nt[g] * nt[g];
      ctx->has_best[g] = nt[g] > 0 ? 1 : 0;
      ctx->best_c[g] = bc[g];
      ctx->best_score[g] = bs[g];
      // src image index -> id
      ctx->best_src2[2 * g] = nt[g] > 0 ? ctx->img_ids[bsrc[2 * g]] : 0;
      ctx->best_src2[2 * g + 1] = nt[g] > 0 ? bsrc[2 * g + 1] : 0;
      ctx->valid_edges.off[(size_t)g] = pool_base + 2 * edge_off[g];
      ctx->valid_edges.cnt[(size_t)g] = (int)(2 * (edge_off[g + 1] - edge_off[g]));
    }
  }
  define_best_of_other_images(ctx);
  ctx->stat_pairs = pairs;
  if (ctx->cfg.debug_mode && ctx->C > 0) {  // keep this batch's tris_ on the host (later batches reuse the device arrays)
    const long long C = ctx->C;
    {
      int rcm = materialize_compact(ctx);
      if (rcm) return rcm;
    }
    std::vector<Cand> c((size_t)C);
    std::vector<CandLite> l((size_t)C);
    std::vector<double> sc((size_t)C);
    HIPCHK(ctx, hipMemcpy(c.data(), ctx->d_hcand.p, sizeof(Cand) * (size_t)C, hipMemcpyDeviceToHost));
    HIPCHK(ctx, hipMemcpy(l.data(), ctx->d_hlite.p, sizeof(CandLite) * (size_t)C, hipMemcpyDeviceToHost));
    HIPCHK(ctx, hipMemcpy(sc.data(), ctx->d_score.p, 8 * (size_t)C, hipMemcpyDeviceToHost));
    for (long long j = 0; j < n_job; ++j) {
      const int idx = ctx->job_imgs[(size_t)j];
      for (long long g = ctx->seg_off[idx]; g < ctx->seg_off[idx + 1]; ++g) {
        ctx->dbg_off[(size_t)g] = (long long)ctx->dbg_pool.size();
        ctx->dbg_cnt[(size_t)g] = (int)(tri_off[g + 1] - tri_off[g]);
        for (long long t = tri_off[g]; t < tri_off[g + 1]; ++t) {
          lt_ctx::DebugTri r;
          for (int k = 0; k < 3; ++k) { r.line10[k] = c[t].s[k]; r.line10[3 + k] = c[t].e[k]; }
          r.line10[6] = c[t].depth[0]; r.line10[7] = c[t].depth[1]; r.line10[8] = c[t].unc; r.line10[9] = c[t].score3;
          r.score = sc[t];
          r.src2[0] = ctx->img_ids[lite_img(l[t])];
          r.src2[1] = l[t].ng_line;
          ctx->dbg_pool.push_back(r);
        }
      }
    }
  }
  ctx->downloaded = true;
  ctx->timers[9] = now_ms() - t0;
  return LT_OK;
}

int lt_flush(lt_ctx *ctx) {
  int rc;
  if (ctx->job_mode == 0 && !ctx->uploaded) {
    if (!ctx->inited) return fail(ctx, LT_ERR_STATE, "flush before Init");
    ctx->downloaded = true;
    return LT_OK;
  }
  if (!ctx->uploaded && (rc = lt_upload(ctx))) return rc;
  if (!ctx->ran && (rc = lt_run_device(ctx))) return rc;
  if (!ctx->downloaded && (rc = lt_download(ctx))) return rc;
  return LT_OK;
}

// ---------------------------------------------------------------------------------------------
// host tail
// ---------------------------------------------------------------------------------------------
// Device half of the tail (lt_kernels_tail.hip): possible when the results of the whole scene are those of the run
// that is still resident in HBM (one batch, nothing imported, nothing read back yet) and no node filter applies
// (min_num_outer_edges == 0, the value of cfgs/triangulation/default.yaml:81).  LT_TAIL_HOST=1 forces the host form.
static bool tail_on_device(const lt_ctx *ctx) {
  if (getenv("LT_TAIL_HOST") != nullptr || ctx->cfg.min_num_outer_edges > 0) return false;
  if (!ctx->inited || ctx->job_mode == 0 || ctx->downloaded || ctx->job_imgs.empty()) return false;
  if (ctx->G <= 0 || ctx->G >= (1ll << 31)) return false;
  for (char c : ctx->best_c_set)
    if (c) return false;  // an earlier batch or imported shards live on the host
  return true;
}

// sorted unique undirected edges + their similarities; the graph nodes' best candidates land in ctx->best_c etc.
// Two host synchronisations: one for the number of valid edges (it sizes the sort), one at the end; the graph
// nodes' records are written by the gather kernel straight into page-locked host memory.
extern "C++" {
template <class AddEdge>
static int tail_from_device(lt_ctx *ctx, AddEdge &&add_edge) {
  LT_FINISH(ctx);
  static const bool trace = getenv("LT_TAIL_TRACE") != nullptr;
  double tp = now_ms();
  auto lap = [&](const char *what) {
    if (!trace) return;
    double t = now_ms();
    fprintf(stderr, "[tail]   %-16s %.3f ms\n", what, t - tp);
    tp = t;
  };
  HIPCHK(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  const long long G = ctx->G;
  // valid-edge offsets (the scan lt_download would run)
  HIPCHK(ctx, hipMemsetAsync(ctx->d_nvalid.as<unsigned>() + G, 0, 4, st));
  const size_t scan_tmp = scan_temp_bytes_u32_to_i64(G + 1);
  ENSURE(ctx, ctx->d_scan_tmp, std::max<size_t>(scan_tmp, 16));
  if (launch_scan_u32_to_i64(st, ctx->d_scan_tmp.p, scan_tmp, G + 1, ctx->d_nvalid.as<unsigned>(),
                             ctx->d_edge_off.as<long long>()) != 0)
    return fail(ctx, LT_ERR_HIP, "rocprim scan failed");
  long long *hp = ctx->h_pinned ? ctx->h_pinned + 16 : nullptr;  // slots behind the two result sets
  long long fallback[2] = {0, 0};
  if (!hp) hp = fallback;
  HIPCHK(ctx, hipMemcpyAsync(&hp[0], ctx->d_edge_off.as<long long>() + G, 8, hipMemcpyDeviceToHost, st));
  ENSURE(ctx, ctx->d_tail_mark, 4 * (size_t)(G + 1)); ENSURE(ctx, ctx->d_tail_pos, 8 * (size_t)(G + 1));
  HIPCHK(ctx, hipMemsetAsync(ctx->d_tail_mark.p, 0, 4 * (size_t)(G + 1), st));
  HIPCHK(ctx, hipStreamSynchronize(st));
  const long long E = hp[0];
  lap("scan + sync (E)");
  ctx->E = E;
  ctx->C = ctx->C_last;
  if (E <= 0) return LT_OK;
  const size_t En = (size_t)E;
  ENSURE(ctx, ctx->d_tail_keys, 8 * En); ENSURE(ctx, ctx->d_tail_skeys, 8 * En); ENSURE(ctx, ctx->d_tail_sims, 8 * En);
  ENSURE(ctx, ctx->d_tail_keep, 4 * (En + 1)); ENSURE(ctx, ctx->d_tail_kpos, 8 * (En + 1));
  const int kb = bits_for(G + 1);  // key = (min node << kb) | max node
  const int end_bit = 2 * kb;
  const size_t sort_tmp = tail_sort_temp_bytes(E, end_bit);
  const size_t scan_tmp2 = scan_temp_bytes_u32_to_i64(E + 1);
  ENSURE(ctx, ctx->d_tail_tmp, std::max<size_t>(std::max(sort_tmp, scan_tmp2), 16));
  // host side of the transfer: counts | (key, sim) of the graph's edges | records | node ids, one pooled page-locked
  // block; at most E distinct edges and min(G, 2 E) nodes enter the graph
  const size_t max_nodes = (size_t)std::min<long long>(G, 2 * E);
  const size_t o_pairs = 64, o_recs = o_pairs + 16 * En, o_nodes = o_recs + tail_rec_bytes() * max_nodes;
  lt_host::HostBlock hb = lt_host::host_block_acquire(o_nodes + 4 * max_nodes);
  if (!hb.p) return fail(ctx, LT_ERR_RUNTIME, "out of host memory for the edge list");
  struct Rel {
    lt_host::HostBlock b;
    ~Rel() { lt_host::host_block_release(b); }
  } rel{hb};
  char *base = (char *)hb.p;
  long long *hn = (long long *)base;  // [0] graph nodes, [1] graph edges
  hn[0] = hn[1] = 0;
  const unsigned long long *hpairs = (const unsigned long long *)(base + o_pairs);
  launch_tail_keys(st, G, ctx->d_tri_off.as<long long>(), ctx->d_edge_flag.as<unsigned>(), ctx->d_edge_off.as<long long>(),
                   ctx->perm_mode ? ctx->d_st_c.as<CRec>() : ctx->d_cand.as<CRec>(), ctx->d_seg_off.as<long long>(), kb,
                   ctx->d_tail_keys.as<unsigned long long>(), ctx->perm_mode ? ctx->d_place_perm.as<unsigned>() : nullptr);
  if (launch_tail_sort(st, ctx->d_tail_tmp.p, sort_tmp, E, ctx->d_tail_keys.as<unsigned long long>(),
                       ctx->d_tail_skeys.as<unsigned long long>(), end_bit) != 0)
    return fail(ctx, LT_ERR_HIP, "rocprim radix sort failed");
  LinkCfg3 l3 = make_l3(ctx->cfg);
  l3.use_angle = 1; l3.use_overlap = 1; l3.use_perp = 0; l3.use_innerseg = 1; l3.use_scaleinv = 0;  // line_linker.h:123-129
  launch_tail_sims(st, E, ctx->d_tail_skeys.as<unsigned long long>(), ctx->d_ntris.as<int>(), ctx->d_best_c.as<Cand>(), l3,
                   kb, ctx->d_tail_sims.as<double>(), ctx->d_tail_mark.as<unsigned>(), ctx->d_tail_keep.as<unsigned>());
  if (launch_scan_u32_to_i64(st, ctx->d_tail_tmp.p, scan_tmp2, E + 1, ctx->d_tail_keep.as<unsigned>(),
                             ctx->d_tail_kpos.as<long long>()) != 0 ||
      launch_scan_u32_to_i64(st, ctx->d_scan_tmp.p, scan_tmp, G + 1, ctx->d_tail_mark.as<unsigned>(),
                             ctx->d_tail_pos.as<long long>()) != 0)
    return fail(ctx, LT_ERR_HIP, "rocprim scan failed");
  if (hb.pinned) {  // the kernels write across PCIe: the host needs no size before the copies
    launch_tail_compact(st, E, ctx->d_tail_skeys.as<unsigned long long>(), ctx->d_tail_sims.as<double>(),
                        ctx->d_tail_keep.as<unsigned>(), ctx->d_tail_kpos.as<long long>(), base + o_pairs, hn + 1);
    launch_tail_gather(st, G, ctx->d_tail_mark.as<unsigned>(), ctx->d_tail_pos.as<long long>(), ctx->d_best_c.as<Cand>(),
                       ctx->d_best_score.as<double>(), ctx->d_best_src.as<int>(), base + o_recs, (int *)(base + o_nodes), hn);
  } else {  // no page-locked memory: pack on the device, copy the bounds
    ENSURE(ctx, ctx->d_tail_recs, 16 * En + tail_rec_bytes() * max_nodes + 64); ENSURE(ctx, ctx->d_tail_nodes, 4 * max_nodes);
    char *dp = (char *)ctx->d_tail_recs.p;
    launch_tail_compact(st, E, ctx->d_tail_skeys.as<unsigned long long>(), ctx->d_tail_sims.as<double>(),
                        ctx->d_tail_keep.as<unsigned>(), ctx->d_tail_kpos.as<long long>(), dp, (long long *)ctx->d_tail_keys.p);
    launch_tail_gather(st, G, ctx->d_tail_mark.as<unsigned>(), ctx->d_tail_pos.as<long long>(), ctx->d_best_c.as<Cand>(),
                       ctx->d_best_score.as<double>(), ctx->d_best_src.as<int>(), dp + 16 * En, ctx->d_tail_nodes.as<int>(),
                       nullptr);
    HIPCHK(ctx, hipMemcpyAsync(hn, ctx->d_tail_pos.as<long long>() + G, 8, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipMemcpyAsync(hn + 1, ctx->d_tail_kpos.as<long long>() + E, 8, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipMemcpyAsync(base + o_pairs, dp, 16 * En, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipMemcpyAsync(base + o_recs, dp + 16 * En, tail_rec_bytes() * max_nodes, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipMemcpyAsync(base + o_nodes, ctx->d_tail_nodes.p, 4 * max_nodes, hipMemcpyDeviceToHost, st));
  }
  HIPCHK(ctx, hipGetLastError());
  lap("enqueue");
  HIPCHK(ctx, hipStreamSynchronize(st));
  lap("sync");
  const long long Nm = hn[0], Ne = hn[1];
  if (Nm < 0 || (size_t)Nm > max_nodes || Ne < 0 || Ne > E)
    return fail(ctx, LT_ERR_RUNTIME, "internal: graph size out of range");
  const unsigned long long mask = (1ull << kb) - 1ull;
  for (long long i = 0; i < Ne; ++i) {  // distinct keys with score != 0 (:284-285), in std::set order
    double sim;
    std::memcpy(&sim, &hpairs[2 * i + 1], 8);
    add_edge((long long)(hpairs[2 * i] >> kb), (long long)(hpairs[2 * i] & mask), sim);
  }
  lap("graph");
  struct Rec {
    Cand c;
    double score;
    int src[2];
  };
  static_assert(sizeof(Rec) == 128, "TailRec layout");
  const Rec *recs = (const Rec *)(base + o_recs);
  const int *nodes = (const int *)(base + o_nodes);
  lt_host::pool_for(Nm, 1024, [&](long long k0, long long k1) {
    for (long long k = k0; k < k1; ++k) {  // distinct nodes: no two iterations touch the same entry
      const long long g = nodes[k];
      ctx->best_c[g] = recs[k].c;
      ctx->best_score[g] = recs[k].score;
      ctx->best_src2[2 * g] = ctx->img_ids[recs[k].src[0]];
      ctx->best_src2[2 * g + 1] = recs[k].src[1];
      ctx->has_best[g] = 1;
    }
  });
  lap("host unpack");
  return LT_OK;
}
}  // extern "C++"

int lt_compute_tracks(lt_ctx *ctx) {
  LT_RANGE("lt_compute_tracks (tail: edge set, similarities, union-find, aggregation)");
  if (ctx->cfg.merging_strategy < 0 || ctx->cfg.merging_strategy > 2)  // global_line_triangulator.cc:314-316
    return fail(ctx, LT_ERR_RUNTIME, "Error!The given merging strategy is not implemented");
  const bool on_device = tail_on_device(ctx);
  lt_host::SpinPool::get(lt_host::row_workers()).wake();  // the host half of the tail shares its loops with the team
  int rc;
  if (on_device) {
    if (!ctx->uploaded && (rc = lt_upload(ctx))) return rc;
    if (!ctx->ran && (rc = lt_run_device(ctx))) return rc;
  } else {
    if ((rc = lt_flush(ctx))) return rc;
    if (ctx->inited) define_best_of_other_images(ctx);
  }
  double t0 = now_ms();
  static const bool tail_trace = getenv("LT_TAIL_TRACE") != nullptr;  // developer: stage times to stderr
  double tprev = t0;
  auto lap = [&](const char *what) {
    if (!tail_trace) return;
    double t = now_ms();
    fprintf(stderr, "[tail] %-18s %.3f ms\n", what, t - tprev);
    tprev = t;
  };
  const long long G = ctx->G;
  // graph in edge order (base/graph.cc:57-87); scratch kept in the context
  using GEdge = lt_ctx::GEdge;
  std::vector<int> &gmap = ctx->tail_gmap;            // global node -> graph node
  std::vector<long long> &gnode = ctx->tail_gnode;    // graph node -> global node
  std::vector<GEdge> &ge = ctx->tail_ge;
  if ((long long)gmap.size() != G) gmap.assign((size_t)G, -1);
  gnode.clear();
  ge.clear();
  auto find_or_create = [&](long long g) {
    if (gmap[(size_t)g] >= 0) return gmap[(size_t)g];
    int id = (int)gnode.size();
    gnode.push_back(g);
    gmap[(size_t)g] = id;
    return id;
  };
  auto add_edge = [&](long long a, long long b, double sim) {  // edges arrive in std::set order, score != 0
    const int n1 = find_or_create(a);
    const int n2 = find_or_create(b);
    ge.push_back(GEdge{sim, n1, n2});
  };
  std::vector<unsigned long long> edges;
  std::vector<double> sims;
  if (on_device) {
    ctx->valid_flags.assign((size_t)G, 1);
    if ((rc = tail_from_device(ctx, add_edge))) return rc;
    lap("device edges+sims");
  } else {
  const int min_outer = ctx->cfg.min_num_outer_edges;
  auto node2 = [&](long long g, int slot, int ng_line) -> long long {
    int img = ctx->h_node_img[g];
    return ctx->seg_off[ctx->neighbors[img][slot]] + ng_line;
  };
  // filterNodeByNumOuterEdges (global_line_triangulator.cc:168-232)
  std::vector<char> flags(G, 1);
  if (min_outer > 0) {
    std::vector<int> counters(G);
    std::vector<std::vector<unsigned>> parents(G);
    for (long long g = 0; g < G; ++g) {
      const auto ve = ctx->valid_edges[g];
      counters[g] = (int)(ve.size() / 2);
      for (size_t e = 0; e + 1 < ve.size(); e += 2) parents[node2(g, ve[e], ve[e + 1])].push_back((unsigned)g);
      if (counters[g] < min_outer) flags[g] = 0;
    }
    std::queue<long long> q;
    for (long long g = 0; g < G; ++g)
      if (!flags[g]) q.push(g);
    while (!q.empty()) {
      long long nd = q.front();
      q.pop();
      for (unsigned p : parents[nd]) {
        if (!flags[p]) continue;
        if (--counters[p] < min_outer) {
          flags[p] = 0;
          q.push(p);
        }
      }
    }
  }
  ctx->valid_flags.assign(flags.begin(), flags.end());
  lap("filter nodes");
  // undirected edge set, ordered like std::set<pair<LineNode, LineNode>> (:243-261): the global
  // node index is monotone in (img_id, line_id)
  {
    // two passes (count, fill) over the nodes in parallel, then a parallel sort
    std::vector<long long> eoff((size_t)G + 1, 0);
#pragma omp parallel for num_threads(lt::host_threads()) schedule(static)
    for (long long g = 0; g < G; ++g) {
      long long n = 0;
      if (flags[g]) {
        const auto ve = ctx->valid_edges[g];
        for (size_t e = 0; e + 1 < ve.size(); e += 2) n += flags[node2(g, ve[e], ve[e + 1])] ? 1 : 0;
      }
      eoff[(size_t)g + 1] = n;
    }
    for (long long g = 0; g < G; ++g) eoff[(size_t)g + 1] += eoff[(size_t)g];
    edges.resize((size_t)eoff[(size_t)G]);
#pragma omp parallel for num_threads(lt::host_threads()) schedule(static)
    for (long long g = 0; g < G; ++g) {
      if (!flags[g]) continue;
      const auto ve = ctx->valid_edges[g];
      long long w = eoff[(size_t)g];
      for (size_t e = 0; e + 1 < ve.size(); e += 2) {
        long long h = node2(g, ve[e], ve[e + 1]);
        if (!flags[h]) continue;
        unsigned long long a = (unsigned long long)std::min(g, h), b = (unsigned long long)std::max(g, h);
        edges[(size_t)w++] = (a << 32) | b;
      }
    }
    __gnu_parallel::sort(edges.begin(), edges.end(), std::less<unsigned long long>(),
                         __gnu_parallel::default_parallel_tag(lt::host_threads()));
    edges.erase(std::unique(edges.begin(), edges.end()), edges.end());
  }
  lap("edge set");
  // edge similarity: score_3d in spatial-merging mode between the two best candidates (:264-290)
  LinkCfg3 l3 = make_l3(ctx->cfg);
  l3.use_angle = 1; l3.use_overlap = 1; l3.use_perp = 0; l3.use_innerseg = 1; l3.use_scaleinv = 0;  // line_linker.h:123-129
  sims.assign(edges.size(), 0.0);
  const long long nEh = (long long)edges.size();
#pragma omp parallel for num_threads(lt::host_threads()) schedule(static)
  for (long long e = 0; e < nEh; ++e) {
    long long a = (long long)(edges[e] >> 32), b = (long long)(edges[e] & 0xFFFFFFFFull);
    const Cand &ca = ctx->best_c[a];
    const Cand &cb = ctx->best_c[b];
    L3 la{mk3(ca.s[0], ca.s[1], ca.s[2]), mk3(ca.e[0], ca.e[1], ca.e[2])};
    L3 lb{mk3(cb.s[0], cb.s[1], cb.s[2]), mk3(cb.e[0], cb.e[1], cb.e[2])};
    // nodes without any candidate hold a value-initialised TriTuple in the reference; its zero
    // line scores 0 against everything (direction 0 -> angle 90 deg)
    sims[e] = (ctx->has_best[a] && ctx->has_best[b]) ? score3d(l3, la, lb, ca.unc, cb.unc, ca.depth) : 0.0;
  }
  lap("edge sims");
    for (size_t e = 0; e < edges.size(); ++e)
      if (sims[e] != 0) add_edge((long long)(edges[e] >> 32), (long long)(edges[e] & 0xFFFFFFFFull), sims[e]);
  }
  ctx->stat_graph_nodes = (long long)gnode.size();
  ctx->stat_graph_edges = (long long)ge.size();
  lap("graph");
  // ComputeLineTrackLabelsGreedy (merging/merging.cc:18-103): edges descending by (sim, idx1, idx2), a total order
  const int n_nodes = (int)gnode.size();
  auto ge_before = [](const GEdge &x, const GEdge &y) {
    if (x.sim != y.sim) return x.sim > y.sim;
    if (x.n1 != y.n1) return x.n1 > y.n1;
    return x.n2 > y.n2;
  };
  {
    // similarities are positive doubles: they order like their bit patterns.  LSD radix sort on the 64 bits (six
    // 11-bit digits, descending), then a comparison sort inside the (rare) runs of equal similarity.
    std::vector<GEdge> &tmp = ctx->tail_ge2;
    tmp.resize(ge.size());
    GEdge *src = ge.data(), *dst = tmp.data();
    const size_t n = ge.size();
    bool all_pos = true;
    for (size_t i = 0; i < n; ++i) all_pos = all_pos && src[i].sim > 0.0;
    if (all_pos && n > 64) {
      unsigned cnt[2048];
      for (int pass = 0; pass < 6; ++pass) {
        const int sh = 11 * pass;
        std::memset(cnt, 0, sizeof(cnt));
        for (size_t i = 0; i < n; ++i) {
          unsigned long long u;
          std::memcpy(&u, &src[i].sim, 8);
          ++cnt[(u >> sh) & 2047u];
        }
        unsigned run = 0;  // descending: the largest digit first
        for (int d = 2047; d >= 0; --d) {
          const unsigned c = cnt[d];
          cnt[d] = run;
          run += c;
        }
        for (size_t i = 0; i < n; ++i) {
          unsigned long long u;
          std::memcpy(&u, &src[i].sim, 8);
          dst[cnt[(u >> sh) & 2047u]++] = src[i];
        }
        std::swap(src, dst);
      }
      // six passes: the result is back in ge (src == ge.data())
      for (size_t i = 0; i < n;) {
        size_t j = i + 1;
        while (j < n && src[j].sim == src[i].sim) ++j;
        if (j - i > 1) std::sort(src + i, src + j, ge_before);
        i = j;
      }
    } else {
      std::sort(ge.begin(), ge.end(), ge_before);
    }
  }
  lap("edge sort");
  std::vector<int> parent(n_nodes, -1);
  // images_in_track (std::set<int> per root in the reference, :52-84): only the set SIZES steer the
  // union, so a bit set per node over the image indices is equivalent
  const size_t W = ((size_t)ctx->n_img + 63) / 64;
  std::vector<unsigned long long> img_bits((size_t)n_nodes * W, 0ull);
  std::vector<int> img_cnt(n_nodes, 1);
  for (int i = 0; i < n_nodes; ++i) {
    const int im = ctx->h_node_img[gnode[i]];
    img_bits[(size_t)i * W + (size_t)im / 64] |= 1ull << (im & 63);
  }
  auto absorb = [&](int dst, int src) {  // images[dst] |= images[src]; images[src] = {}
    int c = 0;
    for (size_t w = 0; w < W; ++w) {
      unsigned long long v = img_bits[(size_t)dst * W + w] | img_bits[(size_t)src * W + w];
      img_bits[(size_t)dst * W + w] = v;
      img_bits[(size_t)src * W + w] = 0ull;
      c += __builtin_popcountll(v);
    }
    img_cnt[dst] = c;
    img_cnt[src] = 0;
  };
  // merging strategies (global_line_triangulator.cc:306-316): greedy unions every edge; "exhaustive" and "avg"
  // first test the two unions with LineLinker3d::check_connection in avgtest mode (line_linker.h:131-137)
  const int strategy = ctx->cfg.merging_strategy;
  LinkCfg3 lavg = make_l3(ctx->cfg);
  lavg.use_angle = 1; lavg.use_overlap = 0; lavg.use_perp = 1; lavg.use_innerseg = 0; lavg.use_scaleinv = 0;
  struct UL {  // a line of a union: endpoints + uncertainty (depths are not read in avgtest mode)
    L3 l;
    double unc;
  };
  auto node_line = [&](int i) {
    const Cand &c = ctx->best_c[gnode[(size_t)i]];
    return UL{L3{mk3(c.s[0], c.s[1], c.s[2]), mk3(c.e[0], c.e[1], c.e[2])}, c.unc};
  };
  std::vector<std::vector<UL>> lines_in_track;  // exhaustive: merging.cc:130, members in insertion order
  std::vector<UL> avg_line;                     // avg: merging.cc:271 (+ the member count)
  std::vector<int> avg_cnt;
  if (strategy == 1) {
    lines_in_track.resize((size_t)n_nodes);
    for (int i = 0; i < n_nodes; ++i) lines_in_track[(size_t)i].push_back(node_line(i));
  } else if (strategy == 2) {
    avg_line.resize((size_t)n_nodes);
    avg_cnt.assign((size_t)n_nodes, 1);
    for (int i = 0; i < n_nodes; ++i) avg_line[(size_t)i] = node_line(i);
  }
  const double nodepth[2] = {0.0, 0.0};
  for (const GEdge &ed : ge) {
    int r1 = uf_root(ed.n1, parent), r2 = uf_root(ed.n2, parent);
    if (r1 == r2) continue;
    if (strategy == 1) {  // merging.cc:150-168: every overlapping pair of the two unions must connect
      bool ok = true;
      for (const UL &a : lines_in_track[(size_t)r1]) {
        for (const UL &b : lines_in_track[(size_t)r2]) {
          if (overlap_oneway(a.l, b.l) <= 0) continue;
          if (!check3d(lavg, a.l, b.l, a.unc, b.unc, nodepth)) {
            ok = false;
            break;
          }
        }
        if (!ok) break;
      }
      if (!ok) continue;
    } else if (strategy == 2) {  // merging.cc:289-292: the running averages must connect
      const UL &a = avg_line[(size_t)r1], &b = avg_line[(size_t)r2];
      if (!check3d(lavg, a.l, b.l, a.unc, b.unc, nodepth)) continue;
    }
    int dst = r1, src = r2;
    if (img_cnt[r1] < img_cnt[r2]) { dst = r2; src = r1; }
    parent[src] = dst;
    absorb(dst, src);
    if (strategy == 1) {
      auto &d = lines_in_track[(size_t)dst];
      auto &sv = lines_in_track[(size_t)src];
      d.insert(d.end(), sv.begin(), sv.end());
      std::vector<UL>().swap(sv);
    } else if (strategy == 2) {  // merging.cc:300-307: count-weighted mean; the new Line3d has uncertainty -1
      const UL d1 = avg_line[(size_t)dst], d2 = avg_line[(size_t)src];
      const double n1 = (double)avg_cnt[(size_t)dst], n2 = (double)avg_cnt[(size_t)src];
      const double ns = (double)(avg_cnt[(size_t)dst] + avg_cnt[(size_t)src]);
      auto wmean = [&](d3 p, d3 q) {
        return mk3((p.x * n1 + q.x * n2) / ns, (p.y * n1 + q.y * n2) / ns, (p.z * n1 + q.z * n2) / ns);
      };
      avg_line[(size_t)dst] = UL{L3{wmean(d1.l.s, d2.l.s), wmean(d1.l.e, d2.l.e)}, -1.0};
      avg_cnt[(size_t)dst] += avg_cnt[(size_t)src];
    }
  }
  // NOTE: the reference's recursive root lookup compresses paths as a side effect and reads
  // parent_nodes[node] afterwards; labels are assigned from the parent array as it stands after
  // the union loop.  uf_root() above applies the same full path compression per lookup.
  std::vector<int> labels(n_nodes, -1);
  int n_tracks = 0;
  for (int i = 0; i < n_nodes; ++i) {
    if (parent[i] == -1) continue;
    int p = parent[i];
    if (parent[p] == -1 && labels[p] == -1) labels[p] = n_tracks++;
  }
  for (int i = 0; i < n_nodes; ++i) {
    if (parent[i] == -1) continue;
    labels[i] = labels[uf_root(i, parent)];
  }
  lap("union-find");
  // build_tracks_from_clusters (:293-351): members in node order per track, flat arrays
  ctx->tracks.clear();
  if (n_nodes > 0) {
    int mx = -1;
    for (int l : labels) mx = std::max(mx, l);
    const size_t nT = (size_t)(mx + 1);
    TrackStore &ts = ctx->tracks;
    ts.off.assign(nT + 1, 0);
    for (int i = 0; i < n_nodes; ++i)
      if (labels[i] >= 0) ++ts.off[(size_t)labels[i] + 1];
    for (size_t t = 0; t < nT; ++t) ts.off[t + 1] += ts.off[t];
    const size_t nM = (size_t)ts.off[nT];
    ts.img_ids.resize(nM); ts.line_ids.resize(nM); ts.node_ids.resize(nM); ts.scores.resize(nM); ts.gnodes.resize(nM);
    ts.line7.resize(7 * nT);
    std::vector<long long> wr(ts.off.begin(), ts.off.end() - 1);
    for (int i = 0; i < n_nodes; ++i) {
      const int tl = labels[i];
      if (tl == -1) continue;
      const long long g = gnode[i];
      const int img = ctx->h_node_img[g];
      const size_t w = (size_t)wr[(size_t)tl]++;
      ts.node_ids[w] = i;
      ts.img_ids[w] = ctx->img_ids[img];
      ts.line_ids[w] = (int)(g - ctx->seg_off[img]);
      ts.scores[w] = ctx->best_score[g];
      ts.gnodes[w] = g;
    }
    // shared with the workers of the persistent team that are awake (lt_pool.h): a few thousand tracks aggregate
    // faster than a sleeping OpenMP team starts on a big host
    lt_host::pool_for((long long)nT, 16, [&](long long t0_, long long t1_) {
      static thread_local AggScratch scratch;
      for (long long t = t0_; t < t1_; ++t) {
        const size_t a = (size_t)ts.off[(size_t)t], n = (size_t)ts.off[(size_t)t + 1] - a;
        aggregate(ctx->best_c, ts.gnodes.data() + a, ts.scores.data() + a, (int)n, ctx->cfg.num_outliers_aggregator,
                  ts.line7.data() + 7 * (size_t)t, scratch);
      }
    });
  }
  for (long long g : gnode) gmap[(size_t)g] = -1;  // leave the scratch map clean
  lap("tracks+aggregate");
  ctx->tracks_done = true;
  ctx->timers[10] = now_ms() - t0;
  return LT_OK;
}

// ---------------------------------------------------------------------------------------------
// getters
// ---------------------------------------------------------------------------------------------
int64_t lt_count_images(lt_ctx *ctx) { return ctx->n_img; }
int64_t lt_count_lines(lt_ctx *ctx, int img_id) {
  auto it = ctx->id2idx.find(img_id);
  if (it == ctx->id2idx.end()) {
    ctx->err = "unknown image id " + std::to_string(img_id);
    return -1;
  }
  return ctx->seg_off[it->second + 1] - ctx->seg_off[it->second];
}
int64_t lt_num_nodes(lt_ctx *ctx) { return ctx->G; }

int lt_get_best(lt_ctx *ctx, double *out_line10, double *out_score, int32_t *out_src2, uint8_t *out_has_best) {
  int rc = lt_flush(ctx);
  if (rc) return rc;
  if (ctx->inited) define_best_of_other_images(ctx);
  for (long long g = 0; g < ctx->G; ++g) {
    const Cand &c = ctx->best_c[g];
    double *o = out_line10 + 10 * g;
    bool hb = ctx->has_best[g];
    for (int k = 0; k < 3; ++k) { o[k] = hb ? c.s[k] : 0.0; o[3 + k] = hb ? c.e[k] : 0.0; }
    o[6] = hb ? c.depth[0] : 0.0; o[7] = hb ? c.depth[1] : 0.0; o[8] = hb ? c.unc : 0.0; o[9] = hb ? c.score3 : 0.0;
    out_score[g] = hb ? ctx->best_score[g] : 0.0;
    out_src2[2 * g] = ctx->best_src2[2 * g];
    out_src2[2 * g + 1] = ctx->best_src2[2 * g + 1];
    out_has_best[g] = ctx->has_best[g];
  }
  return LT_OK;
}

int lt_get_num_tris(lt_ctx *ctx, int32_t *out) {
  int rc = lt_flush(ctx);
  if (rc) return rc;
  std::memcpy(out, ctx->n_tris.data(), 4 * (size_t)ctx->G);
  return LT_OK;
}

// valid_flags_ (global_line_triangulator.cc:168-232, filled by run_clustering :236): needs lt_compute_tracks
int lt_get_valid_flags(lt_ctx *ctx, uint8_t *out_flags) {
  if (!ctx->tracks_done || (long long)ctx->valid_flags.size() != ctx->G)
    return fail(ctx, LT_ERR_STATE, "valid flags are filled by ComputeLineTracks (run_clustering)");
  std::memcpy(out_flags, ctx->valid_flags.data(), (size_t)ctx->G);
  return LT_OK;
}

int64_t lt_num_valid_edges(lt_ctx *ctx) {
  if (lt_flush(ctx)) return -1;
  int64_t n = 0;
  for (int c : ctx->valid_edges.cnt) n += (int64_t)c / 2;
  return n;
}

int lt_get_valid_edges(lt_ctx *ctx, int64_t *out_off, int32_t *out_edges2) {
  int rc = lt_flush(ctx);
  if (rc) return rc;
  int64_t e = 0;
  out_off[0] = 0;
  for (long long g = 0; g < ctx->G; ++g) {
    const auto v = ctx->valid_edges[g];
    if (!v.empty()) std::memcpy(out_edges2 + 2 * e, v.data(), 4 * v.size());
    e += (int64_t)v.size() / 2;
    out_off[g + 1] = e;
  }
  return LT_OK;
}

int64_t lt_num_all_tris(lt_ctx *ctx) {
  if (lt_flush(ctx)) return -1;
  if (ctx->cfg.debug_mode) {  // every batch since Init
    int64_t n = 0;
    for (int c : ctx->dbg_cnt) n += c;
    return n;
  }
  return ctx->C;
}

int lt_get_all_tris(lt_ctx *ctx, int64_t *out_off, double *out_line10, double *out_score, int32_t *out_src2) {
  LT_FINISH(ctx);
  int rc = lt_flush(ctx);
  if (rc) return rc;
  if (ctx->cfg.debug_mode) {  // host store: the candidates of every batch since Init
    int64_t t = 0;
    out_off[0] = 0;
    for (long long g = 0; g < ctx->G; ++g) {
      const lt_ctx::DebugTri *r = ctx->dbg_pool.data() + ctx->dbg_off[(size_t)g];
      for (int k = 0; k < ctx->dbg_cnt[(size_t)g]; ++k, ++t) {
        std::memcpy(out_line10 + 10 * t, r[k].line10, 80);
        out_score[t] = r[k].score;
        out_src2[2 * t] = r[k].src2[0];
        out_src2[2 * t + 1] = r[k].src2[1];
      }
      out_off[g + 1] = t;
    }
    return LT_OK;
  }
  HIPCHK(ctx, hipSetDevice(ctx->device));
  if ((rc = materialize_compact(ctx))) return rc;
  const long long G = ctx->G, C = ctx->C;
  std::vector<long long> tri_off(G + 1);
  HIPCHK(ctx, hipMemcpy(tri_off.data(), ctx->d_tri_off.p, 8 * (size_t)(G + 1), hipMemcpyDeviceToHost));
  for (long long g = 0; g <= G; ++g) out_off[g] = tri_off[g];
  if (C == 0) return LT_OK;
  std::vector<Cand> c(C);
  std::vector<CandLite> l(C);
  HIPCHK(ctx, hipMemcpy(c.data(), ctx->d_hcand.p, sizeof(Cand) * (size_t)C, hipMemcpyDeviceToHost));
  HIPCHK(ctx, hipMemcpy(l.data(), ctx->d_hlite.p, sizeof(CandLite) * (size_t)C, hipMemcpyDeviceToHost));
  HIPCHK(ctx, hipMemcpy(out_score, ctx->d_score.p, 8 * (size_t)C, hipMemcpyDeviceToHost));
  for (long long g = 0; g < G; ++g) {
    for (long long t = tri_off[g]; t < tri_off[g + 1]; ++t) {
      double *o = out_line10 + 10 * t;
      for (int k = 0; k < 3; ++k) { o[k] = c[t].s[k]; o[3 + k] = c[t].e[k]; }
      o[6] = c[t].depth[0]; o[7] = c[t].depth[1]; o[8] = c[t].unc; o[9] = c[t].score3;
      out_src2[2 * t] = ctx->img_ids[lite_img(l[t])];
      out_src2[2 * t + 1] = l[t].ng_line;
    }
  }
  return LT_OK;
}

int64_t lt_num_tracks(lt_ctx *ctx) { return (int64_t)ctx->tracks.size(); }
int64_t lt_num_track_members(lt_ctx *ctx) { return (int64_t)ctx->tracks.members(); }
int lt_get_tracks(lt_ctx *ctx, double *out_line7, int64_t *out_off, int32_t *out_img_ids, int32_t *out_line_ids,
                  int32_t *out_node_ids, double *out_scores, double *out_line3d6) {
  const TrackStore &ts = ctx->tracks;
  const size_t nT = ts.size(), nM = ts.members();
  for (size_t t = 0; t <= nT; ++t) out_off[t] = ts.off[t];
  if (nT) std::memcpy(out_line7, ts.line7.data(), 56 * nT);
  if (nM) {
    std::memcpy(out_img_ids, ts.img_ids.data(), 4 * nM);
    std::memcpy(out_line_ids, ts.line_ids.data(), 4 * nM);
    std::memcpy(out_node_ids, ts.node_ids.data(), 4 * nM);
    std::memcpy(out_scores, ts.scores.data(), 8 * nM);
  }
  for (size_t e = 0; e < nM; ++e) {
    const Cand &c = ctx->best_c[ts.gnodes[e]];
    for (int q = 0; q < 3; ++q) { out_line3d6[6 * e + q] = c.s[q]; out_line3d6[6 * e + 3 + q] = c.e[q]; }
  }
  return LT_OK;
}

// ---- per-image results: export on the rank that triangulated the image, import on the rank that
// runs the tail (multi-GPU: SURVEY.md 8e "Tail") ----
int64_t lt_image_results_size(lt_ctx *ctx, int img_id, int64_t *n_edges) {
  if (lt_flush(ctx)) return -1;
  auto it = ctx->id2idx.find(img_id);
  if (it == ctx->id2idx.end()) {
    ctx->err = "unknown image id " + std::to_string(img_id);
    return -1;
  }
  int idx = it->second;
  int64_t e = 0;
  for (long long g = ctx->seg_off[idx]; g < ctx->seg_off[idx + 1]; ++g) e += (int64_t)ctx->valid_edges[g].size() / 2;
  if (n_edges) *n_edges = e;
  return ctx->seg_off[idx + 1] - ctx->seg_off[idx];
}

int lt_export_image_results(lt_ctx *ctx, int img_id, int32_t *out_nb_ids, int32_t *out_n_nb, double *out_line10,
                            double *out_score, int32_t *out_src2, int32_t *out_n_tris, int64_t *out_edge_off,
                            int32_t *out_edges2) {
  int rc = lt_flush(ctx);
  if (rc) return rc;
  auto it = ctx->id2idx.find(img_id);
  if (it == ctx->id2idx.end()) return fail(ctx, LT_ERR_ARGUMENT, "unknown image id " + std::to_string(img_id));
  int idx = it->second;
  if (!ctx->triangulated[idx]) return fail(ctx, LT_ERR_STATE, "image was not triangulated on this context");
  const auto &nb = ctx->neighbors[idx];
  *out_n_nb = (int32_t)nb.size();
  for (size_t k = 0; k < nb.size(); ++k) out_nb_ids[k] = ctx->img_ids[nb[k]];
  int64_t e = 0;
  long long g0 = ctx->seg_off[idx];
  out_edge_off[0] = 0;
  for (long long g = g0; g < ctx->seg_off[idx + 1]; ++g) {
    long long l = g - g0;
    const Cand &c = ctx->best_c[g];
    double *o = out_line10 + 10 * l;
    for (int k = 0; k < 3; ++k) { o[k] = c.s[k]; o[3 + k] = c.e[k]; }
    o[6] = c.depth[0]; o[7] = c.depth[1]; o[8] = c.unc; o[9] = c.score3;
    out_score[l] = ctx->best_score[g];
    out_src2[2 * l] = ctx->best_src2[2 * g];
    out_src2[2 * l + 1] = ctx->best_src2[2 * g + 1];
    out_n_tris[l] = ctx->n_tris[g];
    const auto v = ctx->valid_edges[g];
    if (!v.empty()) std::memcpy(out_edges2 + 2 * e, v.data(), 4 * v.size());
    e += (int64_t)v.size() / 2;
    out_edge_off[l + 1] = e;
  }
  return LT_OK;
}

int lt_import_image_results(lt_ctx *ctx, int img_id, int n_nb, const int32_t *nb_ids, const double *line10,
                            const double *score, const int32_t *src2, const int32_t *n_tris,
                            const int64_t *edge_off, const int32_t *edges2) {
  if (!ctx->inited) return fail(ctx, LT_ERR_STATE, "import before Init");
  auto it = ctx->id2idx.find(img_id);
  if (it == ctx->id2idx.end()) return fail(ctx, LT_ERR_ARGUMENT, "unknown image id " + std::to_string(img_id));
  int idx = it->second;
  std::vector<int> nb;
  for (int k = 0; k < n_nb; ++k) {
    auto jt = ctx->id2idx.find(nb_ids[k]);
    if (jt == ctx->id2idx.end()) return fail(ctx, LT_ERR_ARGUMENT, "unknown neighbour image id " + std::to_string(nb_ids[k]));
    nb.push_back(jt->second);
  }
  ctx->neighbors[idx] = nb;
  ctx->triangulated[idx] = 1;
  long long g0 = ctx->seg_off[idx];
  for (long long g = g0; g < ctx->seg_off[idx + 1]; ++g) {
    long long l = g - g0;
    Cand &c = ctx->best_c[g];
    c = Cand{};
    const double *o = line10 + 10 * l;
    for (int k = 0; k < 3; ++k) { c.s[k] = o[k]; c.e[k] = o[3 + k]; }
    c.depth[0] = o[6]; c.depth[1] = o[7]; c.unc = o[8]; c.score3 = o[9];
    ctx->best_score[g] = score[l];
    ctx->best_src2[2 * g] = src2[2 * l];
    ctx->best_src2[2 * g + 1] = src2[2 * l + 1];
    ctx->n_tris[g] = n_tris[l];
    ctx->has_best[g] = n_tris[l] > 0 ? 1 : 0;
    ctx->valid_edges.set(g, edges2 + 2 * edge_off[l], edges2 + 2 * edge_off[l + 1]);
  }
  ctx->best_c_set[(size_t)idx] = 1;
  define_best_of_other_images(ctx);
  ctx->tracks_done = false;
  return LT_OK;
}

int lt_get_stats(lt_ctx *ctx, int64_t out[8]) {
  LT_FINISH(ctx);
  if (ctx->inited && ctx->ran && !ctx->downloaded) {  // the pair statistic is summed from the per-node counts
    int rc = lt_download(ctx);
    if (rc) return rc;
  }
  out[0] = ctx->n_conn; out[1] = ctx->C; out[2] = ctx->stat_pairs; out[3] = ctx->E;
  out[4] = ctx->stat_graph_nodes; out[5] = ctx->stat_graph_edges; out[6] = (int64_t)ctx->tracks.size();
  out[7] = ctx->G;
  return LT_OK;
}
int lt_get_timers(lt_ctx *ctx, double out[24]) {
  LT_FINISH(ctx);
  ctx->timers[11] = (double)ctx->stat_pairs_eval;
  if (ctx->stat_survivors < 0) {
    ctx->stat_survivors = 0;
    if (ctx->ran && ctx->job_mode == 1 && ctx->n_blk > 0 && ctx->max_rows > 0 && ctx->d_surv_count.p) {
      const size_t n = (size_t)ctx->n_blk * (size_t)gen_slots(ctx->max_rows);
      std::vector<unsigned> sc(n);
      HIPCHK(ctx, hipSetDevice(ctx->device));
      HIPCHK(ctx, hipMemcpy(sc.data(), ctx->d_surv_count.p, 4 * n, hipMemcpyDeviceToHost));
      long long tot = 0;
      for (unsigned v : sc) tot += v;
      ctx->stat_survivors = tot;
    }
  }
  ctx->timers[16] = (double)ctx->stat_survivors;
  std::memcpy(out, ctx->timers, sizeof(ctx->timers));
  return LT_OK;
}

int lt_get_timer_sums(lt_ctx *ctx, double out[24], int64_t *n_runs, int reset) {
  LT_FINISH(ctx);
  std::memcpy(out, ctx->timer_sums, sizeof(ctx->timer_sums));
  if (n_runs) *n_runs = ctx->timer_runs;
  if (reset) {
    std::memset(ctx->timer_sums, 0, sizeof(ctx->timer_sums));
    ctx->timer_runs = 0;
  }
  return LT_OK;
}

// ---- free functions ----
static int fn_query(lt_ctx *ctx, const double *seg1, const double *cam1, const double *seg2, const double *cam2,
                    int by_endpoints, double out50[50], const double *v3 = nullptr, const double *p1 = nullptr,
                    const double *p2 = nullptr) {
  HIPCHK(ctx, hipSetDevice(ctx->device));
  double in[37] = {0};
  std::memcpy(in, seg1, 32); std::memcpy(in + 4, cam1, 88); std::memcpy(in + 15, seg2, 32); std::memcpy(in + 19, cam2, 88);
  if (v3) std::memcpy(in + 30, v3, 24);
  if (p1) std::memcpy(in + 33, p1, 16);
  if (p2) std::memcpy(in + 35, p2, 16);
  DevBuf din, dout;
  ENSURE(ctx, din, sizeof(in)); ENSURE(ctx, dout, 50 * 8);
  HIPCHK(ctx, hipMemcpyAsync(din.p, in, sizeof(in), hipMemcpyHostToDevice, ctx->stream));
  launch_fn_query(ctx->stream, din.as<double>(), by_endpoints, dout.as<double>());
  HIPCHK(ctx, hipMemcpyAsync(out50, dout.p, 50 * 8, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  din.release(); dout.release();
  return LT_OK;
}

int lt_fn_get_normal_direction(lt_ctx *ctx, const double seg[4], const double cam[11], double out[3]) {
  double o[50];
  int rc = fn_query(ctx, seg, cam, seg, cam, 0, o);
  if (rc) return rc;
  std::memcpy(out, o, 24);
  return LT_OK;
}
int lt_fn_get_direction_from_vp(lt_ctx *ctx, const double vp[3], const double cam[11], double out[3]) {
  double o[50], seg[4] = {0, 0, 1, 1};
  int rc = fn_query(ctx, seg, cam, seg, cam, 0, o, vp);
  if (rc) return rc;
  std::memcpy(out, o + 23, 24);
  return LT_OK;
}
int lt_fn_compute_fundamental_matrix(lt_ctx *ctx, const double cam1[11], const double cam2[11], double out[9]) {
  double o[50], seg[4] = {0, 0, 1, 1};
  int rc = fn_query(ctx, seg, cam1, seg, cam2, 0, o);
  if (rc) return rc;
  std::memcpy(out, o + 3, 72);
  return LT_OK;
}
int lt_fn_compute_epipolar_IoU(lt_ctx *ctx, const double seg1[4], const double cam1[11], const double seg2[4],
                               const double cam2[11], double *out) {
  double o[50];
  int rc = fn_query(ctx, seg1, cam1, seg2, cam2, 0, o);
  if (rc) return rc;
  *out = o[12];
  return LT_OK;
}
int lt_fn_triangulate_point(lt_ctx *ctx, const double p1[2], const double cam1[11], const double p2[2],
                            const double cam2[11], double out[3], int *ok) {
  double o[50], seg[4] = {0, 0, 1, 1};
  int rc = fn_query(ctx, seg, cam1, seg, cam2, 0, o, nullptr, p1, p2);
  if (rc) return rc;
  std::memcpy(out, o + 26, 24);
  if (ok) *ok = o[29] != 0.0;
  return LT_OK;
}
int lt_fn_triangulate_line(lt_ctx *ctx, const double seg1[4], const double cam1[11], const double seg2[4],
                           const double cam2[11], int by_endpoints, double out_line10[10]) {
  double o[50];
  int rc = fn_query(ctx, seg1, cam1, seg2, cam2, by_endpoints, o);
  if (rc) return rc;
  std::memcpy(out_line10, o + 13, 80);
  return LT_OK;
}
int lt_fn_triangulate_line_with_direction(lt_ctx *ctx, const double seg1[4], const double cam1[11],
                                          const double seg2[4], const double cam2[11], const double direction[3],
                                          double out_line10[10]) {
  double o[50];
  int rc = fn_query(ctx, seg1, cam1, seg2, cam2, 0, o, direction);
  if (rc) return rc;
  std::memcpy(out_line10, o + 30, 80);
  return LT_OK;
}
int lt_fn_triangulate_line_with_one_point(lt_ctx *ctx, const double seg1[4], const double cam1[11],
                                          const double seg2[4], const double cam2[11], const double point[3],
                                          double out_line10[10]) {
  double o[50];
  int rc = fn_query(ctx, seg1, cam1, seg2, cam2, 0, o, point);
  if (rc) return rc;
  std::memcpy(out_line10, o + 40, 80);
  return LT_OK;
}

int lt_fn_pack_match_rows(const int32_t *rows, int64_t n, uint32_t *out, uint32_t stats[3], int level) {
  if (n < 0 || (n > 0 && (!rows || !out)) || !stats) return LT_ERR_ARGUMENT;
  const lt::RowStats rs = lt::pack_rows(rows, n, out, level);
  stats[0] = rs.mx_line;
  stats[1] = rs.mx_ng;
  stats[2] = (uint32_t)rs.unsorted;
  return LT_OK;
}

int lt_fn_aggregate_line3d_list(int n, const double *lines10, const double *scores, int num_outliers, double out7[7]) {
  if (n <= 0 || !lines10 || !scores || !out7 || num_outliers < 0) return LT_ERR_ARGUMENT;
  if (n >= 4 && 2 * num_outliers >= 2 * n) return LT_ERR_ARGUMENT;  // projections[num_outliers] would be out of range
  std::vector<Cand> c((size_t)n);
  std::vector<const Cand *> ptr((size_t)n);
  for (int i = 0; i < n; ++i) {
    const double *o = lines10 + 10 * (size_t)i;
    for (int k = 0; k < 3; ++k) { c[(size_t)i].s[k] = o[k]; c[(size_t)i].e[k] = o[3 + k]; }
    c[(size_t)i].depth[0] = o[6]; c[(size_t)i].depth[1] = o[7]; c[(size_t)i].unc = o[8]; c[(size_t)i].score3 = o[9];
    ptr[(size_t)i] = &c[(size_t)i];
  }
  std::vector<double> sc(scores, scores + n);
  lt::aggregate(ptr, sc, num_outliers, out7);
  return LT_OK;
}

}  // extern "C"
