// lt_api.cpp -- C ABI (include/limap_amd.h) of the MI355X line-triangulation backend: memory pools, context,
// configuration and the scene (Init / InitVPResults / SetBipartites2d / device-resident scenes).  The rest of the
// ABI: lt_api_rows.cpp (TriangulateImage*: buffering the match rows), lt_api_run.cpp (upload, the device run,
// download), lt_api_tail.cpp (ComputeLineTracks), lt_api_query.cpp (getters, per-image export / import, timers, free
// functions), lt_tracks.cpp (track post-processing).  Shared declarations: lt_host.h.
//
// The tail follows global_line_triangulator.cc:168-351 (filterNodeByNumOuterEdges, run_clustering,
// build_tracks_from_clusters), base/graph.cc:57-87,156-165, merging/merging.cc:18-103
// (ComputeLineTrackLabelsGreedy) and merging/aggregator.cc:8-101; it runs on the host in C++
// because it is a serial union-find over a few 10^4..10^6 edges (SURVEY.md 8e "Tail").
// There is no CPU fallback for the kernels: without a GPU lt_create fails.

#include "lt_host.h"

using namespace lt;
using namespace lt_impl;

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
Range::Range(const char *name) {
  init_once();
  on = g_push != nullptr;
  if (on) g_push(name);
}
Range::~Range() {
  if (on) g_pop();
}
}  // namespace lt_trace

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

extern "C" int lt_reserve_host(uint64_t bytes, int blocks) {
  if (blocks <= 0 || bytes == 0) return LT_OK;
  std::vector<lt_host::HostBlock> held;
  bool ok = true;
  for (int k = 0; k < blocks && k < 8; ++k) {  // held together so that they are distinct blocks
    lt_host::HostBlock b = lt_host::host_block_acquire((size_t)bytes);
    ok = ok && b.p != nullptr && b.pinned;
    held.push_back(b);
  }
  for (auto &b : held) lt_host::host_block_release(b);
  return ok ? LT_OK : LT_ERR_HIP;
}

namespace lt_impl {

// per-kernel HIP events cost a few microseconds of stream bubble each.  LT_FINE_TIMERS (read per run): unset / 1 =
// the event in front of k_score3 only (timer [15]: bench.py prices the dominant kernel with it, inside its timed
// region), 2 = also the events around k_gates and k_tri_rows (timers [13], [14]: +5 us per step), 0 = none
const char *test_switch(const char *name) {
  static const bool on = [] {
    const char *e = getenv("LT_ENABLE_TEST_SWITCHES");
    return e && e[0] == '1';
  }();
  return on ? getenv(name) : nullptr;
}
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

}  // namespace lt_impl

namespace lt_impl {

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
  g.force_undecided = test_switch("LT_TEST_NO_FAST_GATES") != nullptr;
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
      g.sens_lo2 = g.sens_lo * g.sens_lo * (1.0 - 1e-9);
      g.sens_hi2 = g.sens_hi * g.sens_hi * (1.0 + 1e-9);
    } else {
      g.sens_lo = -1.0;   // the band covers everything: always the exact expression
      g.sens_hi = 1e300;
      g.sens_lo2 = -1.0;
      g.sens_hi2 = 1e300;
    }
    if (test_switch("LT_TEST_NO_FAST_GATES")) {  // every candidate through the reference's form of the sensitivity test
      g.sens_lo2 = -1.0;
      g.sens_hi2 = 1e300;
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
  if (test_switch("LT_TEST_NO_SCORE_GUARDS")) s.cos_guard = -1.0;
  s.fullscore_th = ctx->cfg.fullscore_th;
  s.max_valid_conns = ctx->cfg.max_valid_conns;
  // single-exp form of pair_score: gate bands around q_th = sqrt(-2 ln score_th) (+-1e-9 relative: at score_th in
  // [1e-3, 0.999] the exponential at the band's edges differs from score_th by >= 2e-12 relative, four orders above the
  // error of exp and of q * q).  Outside that range, or with the 2D inner-segment term on, the term-by-term form runs.
  auto band = [](double th, double *lo, double *hi) -> bool {
    if (!(th >= 1e-3 && th <= 0.999)) {
      *lo = -1.0;
      *hi = 1e300;
      return false;
    }
    const double q = std::sqrt(-2.0 * std::log(th));
    *lo = q * (1.0 - 1e-9);
    *hi = q * (1.0 + 1e-9);
    return true;
  };
  const bool b3 = band(s.l3.score_th, &s.q3_lo, &s.q3_hi), b2 = band(s.l2.score_th, &s.q2_lo, &s.q2_hi);
  // (positive thresholds: every q = value / (threshold x multiplier) is then >= 0 or NaN, which is what the bands assume)
  const bool pos = s.l3.th_angle > 0.0 && s.l3.th_scaleinv > 0.0 && s.l2.th_angle > 0.0 && s.l2.th_perp > 0.0 &&
                   s.l2.th_smartangle > 0.0 && s.l2.th_smartangle <= s.l2.th_angle && s.l2.th_smartoverlap > s.l2.th_overlap;
  s.fast = (b3 && b2 && pos && !s.l2.use_innerseg && !test_switch("LT_TEST_PAIR_SCORE_TERMS")) ? 1 : 0;
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
  ctx->h_c_off.clear(); ctx->h_ovf_off.clear(); ctx->h_line0.clear(); ctx->h_ovf.clear();
  ctx->rows_sorted = true;
  ctx->uploaded = ctx->ran = ctx->downloaded = false;
  return LT_OK;
}

// hk / hq / ht / hs: the scene in host memory (ascending id order) if the caller has it, else it is read back
// sync = false: the caller's uploads read page-locked memory that outlives them (lt_init) -- nothing here needs the
// device to have finished, later work is ordered behind it on the context's stream
int build_invariants(lt_ctx *ctx, const double *hk = nullptr, const double *hq = nullptr, const double *ht = nullptr,
                     const double *hs = nullptr, bool sync = true) {
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
  if (sync) HIPCHK(ctx, hipStreamSynchronize(st));
  {  // host copies for the tail-side filters (small: 88 B per image + 32 B per segment)
    const int n = ctx->n_img;
    std::vector<double> k(4 * (size_t)n), q(4 * (size_t)n), t(3 * (size_t)n);
    if (hk && hq && ht && hs) {
      std::memcpy(k.data(), hk, 32 * (size_t)n);
      std::memcpy(q.data(), hq, 32 * (size_t)n);
      std::memcpy(t.data(), ht, 24 * (size_t)n);
      std::vector<double>().swap(ctx->h_segs);
      ctx->h_segs_ptr = hs;  // (the context's own scene block: lives until the next Init)
      ctx->h_segs_add = ctx->cfg.add_halfpix ? 0.5 : 0.0;
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
    if (!(hk && hq && ht && hs)) {
      if (ctx->cfg.add_halfpix)
        for (double &v : ctx->h_segs) v = v + 0.5;
      ctx->h_segs_ptr = ctx->h_segs.data();
      ctx->h_segs_add = 0.0;
    }
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

}  // namespace lt_impl

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

int lt_abi_version(void) { return 2; }  // 2: lt_get_tracks hands out line10 per support (was start3, end3)
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

void lt_destroy(lt_ctx *ctx) {
  if (!ctx) return;
  (void)finish_run(ctx);
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  DevBuf *bufs[] = {&ctx->d_kvec, &ctx->d_qvec, &ctx->d_tvec, &ctx->d_segs_raw, &ctx->d_cams, &ctx->d_segs,
                    &ctx->d_seg_off, &ctx->d_node_img, &ctx->d_nb_off, &ctx->d_blk_img, &ctx->d_blk_nb,
                    &ctx->d_blk_slot, &ctx->d_blk_order, &ctx->d_m_off, &ctx->d_m_pairs, &ctx->d_c_stream, &ctx->d_ovf, &ctx->d_rowdesc, &ctx->d_pairs,
                    &ctx->d_keys, &ctx->d_rows, &ctx->d_row_blk, &ctx->d_skeys, &ctx->d_srows, &ctx->d_sort_tmp,
                    &ctx->d_conn_off, &ctx->d_st_c, &ctx->d_st_l, &ctx->d_flags, &ctx->d_pos, &ctx->d_scan_tmp,
                    &ctx->d_item_off, &ctx->d_masks, &ctx->d_mask_cnt, &ctx->d_mask_pos, &ctx->d_cand, &ctx->d_hcand, &ctx->d_hlite,
                    &ctx->d_rm_line, &ctx->d_rm_act, &ctx->d_rm_edges, &ctx->d_rm_cnt,
                    &ctx->d_lite, &ctx->d_tri_off, &ctx->d_score, &ctx->d_best_idx, &ctx->d_edge_flag,
                    &ctx->d_nvalid, &ctx->d_edge_off, &ctx->d_edges, &ctx->d_best_c, &ctx->d_best_score,
                    &ctx->d_best_src, &ctx->d_ntris, &ctx->d_err, &ctx->d_blk_line_base, &ctx->d_cnt_bl,
                    &ctx->d_st_key, &ctx->d_wave_count, &ctx->d_wave_pos, &ctx->d_ntris_u, &ctx->d_cand_node,
                    &ctx->d_pair_counter, &ctx->d_result3, &ctx->d_tile_order, &ctx->d_scan_status, &ctx->d_perm, &ctx->d_rng, &ctx->d_chunks, &ctx->d_cand_meta, &ctx->d_st_row, &ctx->d_surv_count, &ctx->d_seg_gates, &ctx->d_blkrec, &ctx->d_seg_vp, &ctx->d_seg_has_vp, &ctx->d_base_bl, &ctx->d_blk_chunk_off, &ctx->d_needed, &ctx->d_seg_pts, &ctx->d_seg_pt_off, &ctx->d_sfm_xyz,
                    &ctx->d_place_perm, &ctx->d_ex_rec, &ctx->d_ex_ent, &ctx->d_node_rec, &ctx->d_ex_z, &ctx->d_tile_list, &ctx->d_tail_keys, &ctx->d_tail_skeys, &ctx->d_tail_sims, &ctx->d_tail_mark,
                    &ctx->d_tail_pos, &ctx->d_tail_recs, &ctx->d_tail_nodes, &ctx->d_tail_tmp, &ctx->d_tail_keep,
                    &ctx->d_tail_kpos, &ctx->d_sp_slots, &ctx->d_sp_cnt, &ctx->d_sp_ovf, &ctx->d_sp_pairs, &ctx->d_sp_desc,
                    &ctx->d_run_len, &ctx->d_slot_row0, &ctx->d_blk_nruns, &ctx->d_ln_flag, &ctx->d_blk_surv, &ctx->d_blk_rnd0,
                    &ctx->d_round_count};
  // (DevBuf releases itself when the context is deleted below; the list only makes the order explicit)
  lt_host::host_block_release(ctx->h_pinned_blk);
  if (ctx->tail_pend.active) lt_host::host_block_release(ctx->tail_pend.hb);  // a tail begun and never collected
  if (ctx->ev_tail) (void)hipEventDestroy(ctx->ev_tail);
  lt_host::host_block_release(ctx->best_c_blk);
  lt_host::host_block_release(ctx->init_blk);
  for (DevBuf *b : bufs) b->release();
  for (auto &e : ctx->ev_b)
    if (e) (void)hipEventDestroy(e);
  for (auto &e : ctx->ev_end)
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
  // gather into ascending-id order, into ONE pooled page-locked block [k | q | t | s]: the uploads are then truly
  // asynchronous (from pageable memory the runtime stages every copy before it returns) and Init does not wait for them
  const size_t nI = (size_t)n_img, nG = (size_t)ctx->G;
  const size_t o_q = 4 * nI, o_t = 8 * nI, o_s = 11 * nI + (nI & 1), n_dbl = o_s + 4 * nG;
  if (ctx->init_blk.p) {  // an earlier Init of this context: its copies may not have been consumed yet
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    lt_host::host_block_release(ctx->init_blk);
    ctx->init_blk = lt_host::HostBlock{};
  }
  ctx->init_blk = lt_host::host_block_acquire(8 * std::max<size_t>(n_dbl, 1));
  if (!ctx->init_blk.p) return fail(ctx, LT_ERR_RUNTIME, "out of host memory for the scene");
  double *k = (double *)ctx->init_blk.p, *q = k + o_q, *t = k + o_t, *s = k + o_s;
  for (int i = 0; i < n_img; ++i) {
    int p = perm[i];
    std::memcpy(&k[4 * i], kvec + 4 * p, 32);
    std::memcpy(&q[4 * i], qvec + 4 * p, 32);
    std::memcpy(&t[3 * i], tvec + 3 * p, 24);
    long long m = seg_off[p + 1] - seg_off[p];
    if (m > 0) std::memcpy(&s[4 * ctx->seg_off[i]], segs + 4 * seg_off[p], (size_t)m * 32);
  }
  lap("gather");
  {
    hipStream_t st = ctx->stream;
    ENSURE(ctx, ctx->d_kvec, 32 * std::max<size_t>(nI, 1)); ENSURE(ctx, ctx->d_qvec, 32 * std::max<size_t>(nI, 1));
    ENSURE(ctx, ctx->d_tvec, 24 * std::max<size_t>(nI, 1)); ENSURE(ctx, ctx->d_segs_raw, 32 * std::max<size_t>(nG, 1));
    if (nI > 0) {
      HIPCHK(ctx, hipMemcpyAsync(ctx->d_kvec.p, k, 32 * nI, hipMemcpyHostToDevice, st));
      HIPCHK(ctx, hipMemcpyAsync(ctx->d_qvec.p, q, 32 * nI, hipMemcpyHostToDevice, st));
      HIPCHK(ctx, hipMemcpyAsync(ctx->d_tvec.p, t, 24 * nI, hipMemcpyHostToDevice, st));
    }
    if (nG > 0) HIPCHK(ctx, hipMemcpyAsync(ctx->d_segs_raw.p, s, 32 * nG, hipMemcpyHostToDevice, st));
  }
  lap("upload");
  rc = build_invariants(ctx, k, q, t, s, /*sync=*/!ctx->init_blk.pinned);
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
extern "C++" {
namespace lt_impl {
int upload_points(lt_ctx *ctx) {
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
}  // namespace lt_impl
}  // extern "C++"

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

}  // extern "C"
