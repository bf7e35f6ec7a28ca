// lt_ctx.h -- context of the C ABI, shared by lt_api.cpp (pipeline, tail) and lt_tracks.cpp
// (post-triangulation filters and remerge).
#pragma once

#include "../../include/limap_amd.h"
#include "lt_device.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <hip/hip_runtime.h>
#include <unordered_map>
#include <vector>

namespace lt_host {
using namespace lt;

// Process-wide cache of device blocks (lt_api.cpp): contexts are short-lived (one per scene), hipMalloc /
// hipFree are not (each is a driver call, hipFree also synchronises the device).
void *dev_block_acquire(size_t bytes, size_t *cap);
void dev_block_release(void *p, size_t cap);

struct DevBuf {
  void *p = nullptr;
  size_t cap = 0;
  DevBuf() = default;
  DevBuf(const DevBuf &) = delete;
  DevBuf &operator=(const DevBuf &) = delete;
  // every buffer of a context returns to the block cache when the context goes (lt_destroy synchronises first): a member
  // that is missing from a hand-kept release list can no longer leak
  ~DevBuf() { release(); }
  void take(DevBuf &o) {  // this buffer's block goes back to the cache, o's block moves here
    release();
    p = o.p; cap = o.cap;
    o.p = nullptr; o.cap = 0;
  }
  // NOTE: growing replaces the block (contents are NOT kept); the old block returns to the cache
  // only after the device is idle, because work of this context may still be using it.
  bool ensure(size_t bytes) {
    if (bytes <= cap) return true;
    if (p) {
      (void)hipDeviceSynchronize();
      dev_block_release(p, cap);
    }
    p = nullptr;
    cap = 0;
    p = dev_block_acquire(bytes + bytes / 8 + 256, &cap);
    return p != nullptr;
  }
  void release() {  // callers make sure no work is in flight on the block (lt_destroy synchronises)
    if (p) dev_block_release(p, cap);
    p = nullptr;
    cap = 0;
  }
  template <class T>
  T *as() const { return reinterpret_cast<T *>(p); }
};

// Process-wide pool of page-locked host blocks (lt_api.cpp).  Staging the match rows of a batch needs
// tens of MB; fresh pageable memory costs a page fault per 4 KB and a bounce copy in the H2D path,
// a pooled pinned block costs neither after its first use.  Falls back to malloc if pinning fails.
struct HostBlock {
  void *p = nullptr;
  size_t bytes = 0;
  bool pinned = false;
};
HostBlock host_block_acquire(size_t bytes);
void host_block_release(HostBlock b);

// growable int buffer without value-initialisation (the match rows are overwritten right away)
struct RawInts {
  HostBlock blk;
  size_t n = 0;
  ~RawInts() { host_block_release(blk); }
  RawInts() = default;
  RawInts(const RawInts &) = delete;
  RawInts &operator=(const RawInts &) = delete;
  size_t size() const { return n; }
  size_t capacity() const { return blk.bytes / sizeof(int); }
  int *data() { return (int *)blk.p; }
  const int *data() const { return (const int *)blk.p; }
  void clear() { n = 0; }
  // size := want, contents beyond the old size are uninitialised.  NOTE: may move the data; the caller
  // must make sure no asynchronous copy still reads the old block (capacity() tells in advance).
  bool grow_to(size_t want) {
    if (want > capacity()) {
      size_t nc = std::max(want, capacity() + capacity() / 2 + 1024);
      HostBlock q = host_block_acquire(nc * sizeof(int));
      if (!q.p) return false;
      if (n) std::memcpy(q.p, blk.p, n * sizeof(int));
      host_block_release(blk);
      blk = q;
    }
    n = want;
    return true;
  }
  bool reserve(size_t want) {
    size_t keep = n;
    if (!grow_to(std::max(want, n))) return false;
    n = keep;
    return true;
  }
};

// A per-node host array that starts as zeros WITHOUT being written: calloc of a large block is fresh zero pages from the
// kernel, touched only where somebody writes or reads.  Init of a context used to assign ~45 bytes of zeros per node to the
// per-node result arrays below -- 8 ms at 10^6 nodes -- although a job whose tail runs on the device never downloads them.
template <class T>
struct ZeroVec {
  T *p = nullptr;
  size_t n = 0;
  ZeroVec() = default;
  ZeroVec(const ZeroVec &) = delete;
  ZeroVec &operator=(const ZeroVec &) = delete;
  ~ZeroVec() { std::free(p); }
  void assign(size_t count, T zero) {  // (only zeros)
    (void)zero;
    std::free(p);
    p = count ? static_cast<T *>(std::calloc(count, sizeof(T))) : nullptr;
    n = p ? count : 0;
    if (count && !p) throw std::bad_alloc();
  }
  size_t size() const { return n; }
  bool empty() const { return n == 0; }
  T *data() { return p; }
  const T *data() const { return p; }
  T &operator[](size_t i) { return p[i]; }
  const T &operator[](size_t i) const { return p[i]; }
  const T *begin() const { return p; }
  const T *end() const { return p + n; }
};

// Valid edges of every node, flat (slot, ng_line) pairs: one pool + per-node (offset, count), so that a
// download is one bulk append instead of one allocation per node.  Nodes of later batches append.
struct EdgeStore {
  struct View {
    const int *p;
    size_t n;
    size_t size() const { return n; }
    bool empty() const { return n == 0; }
    const int *data() const { return p; }
    int operator[](size_t i) const { return p[i]; }
  };
  ZeroVec<long long> off;
  ZeroVec<int> cnt;   // ints (2 per edge)
  std::vector<int> pool;
  void reset(long long G) { off.assign((size_t)G, 0); cnt.assign((size_t)G, 0); pool.clear(); }
  View operator[](long long g) const { return View{pool.data() + off[(size_t)g], (size_t)cnt[(size_t)g]}; }
  void set(long long g, const int *first, const int *last) {
    off[(size_t)g] = (long long)pool.size();
    cnt[(size_t)g] = (int)(last - first);
    pool.insert(pool.end(), first, last);
  }
};

// GetTracks() as flat arrays (CSR over the members): LineTrack fields of base/linetrack.h:33-42
struct TrackStore {
  std::vector<long long> off{0};          // T + 1
  std::vector<double> line7;              // start3, end3, uncertainty per track
  std::vector<int> img_ids, line_ids, node_ids;
  std::vector<double> scores;
  std::vector<long long> gnodes;          // global node index of every member
  size_t size() const { return off.size() - 1; }
  size_t members() const { return img_ids.size(); }
  void clear() {
    off.assign(1, 0);
    line7.clear(); img_ids.clear(); line_ids.clear(); node_ids.clear(); scores.clear(); gnodes.clear();
  }
};

}  // namespace lt_host

using lt_host::DevBuf;
using lt_host::TrackStore;

struct lt_ctx {
  using Cand = lt::Cand;
  lt_config cfg;
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = true;
  hipStream_t pool_stream = nullptr;  // the stream this context created (returned to the cache on destroy)
  std::string err;
  bool ranges_on = false;
  double lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};

  // ---- scene ----
  bool inited = false;
  int n_img = 0;
  std::vector<int> img_ids;  // ascending
  std::unordered_map<int, int> id2idx;
  std::vector<long long> seg_off;  // n_img+1
  long long G = 0;
  std::vector<int> h_node_img;  // node -> image index
  std::vector<lt::Cam> h_cams;     // host copy of the camera table (post-triangulation filters)
  std::vector<double> h_segs;     // host copy of the 2D segments, x1 y1 x2 y2 (after add_halfpix) -- only where Init got device buffers
  const double *h_segs_ptr = nullptr;  // the 2D segments on the host: h_segs, or the scene block of lt_init (init_blk: no 32 B per
  double h_segs_add = 0.0;             // segment copied at Init -- 3 ms at 10^6 segments); a reader adds h_segs_add (add_halfpix)
  DevBuf d_kvec, d_qvec, d_tvec, d_segs_raw, d_cams, d_segs, d_seg_off, d_node_img;

  // ---- buffered job ----
  int job_mode = 0;  // 0 none, 1 matched, 2 exhaustive
  std::vector<int> job_imgs;               // image indices in call order
  std::vector<std::vector<int>> job_nbs;   // per job image: neighbour image indices, processing order
  std::vector<std::vector<int>> job_order; // per job image: slots in ascending neighbour-id order
  std::vector<long long> h_m_off;          // per block row offsets (n_blk+1), matched mode
  // the staged match rows: a stream of 32-bit words (pinned, call order) holding every block in the COMPRESSED form of
  // lt_rows.h (17 bits per row: neighbour lines + "new line" bits) at h_c_off[block]; blocks that cannot take that form
  // lie in the plain form (line | neighbour line << 16) in h_ovf at h_ovf_off[block] (-1: compressed).  k_expand_rows
  // rebuilds the plain rows of every block on the device (d_m_pairs, device block order) from both.
  lt_host::RawInts h_m_pairs;
  size_t streamed_ints = 0;                // prefix of the stream already enqueued to d_c_stream
  std::vector<long long> h_c_off, h_ovf_off;  // per call-order block
  std::vector<int> h_line0;                // per call-order block: line id of its first row
  std::vector<unsigned> h_ovf;
  std::mutex ovf_mu;
  DevBuf d_c_stream, d_ovf, d_rowdesc;
  std::vector<char> triangulated;          // per image: already passed to TriangulateImage*
  bool uploaded = false, ran = false, downloaded = false;
  // neighbours_ of every triangulated image (ids), persists for the tail
  std::vector<std::vector<int>> neighbors;  // image idx -> neighbour image indices (slot order)

  // ---- device job tables ----
  int n_blk = 0;
  int max_nb = 1;
  long long P = 0;        // connections (matched) / work items (exhaustive: n_items)
  long long max_rows = 0; // matched: most rows of any (image, neighbour) block
  long long n_conn = 0;   // connections tested (stat)
  std::vector<long long> h_nb_off;  // n_img+1
  std::vector<int> h_blk_img, h_blk_nb, h_blk_slot, h_blk_order;
  std::vector<long long> h_item_off;  // exhaustive: per node first item (G+1)
  DevBuf d_nb_off, d_blk_img, d_blk_nb, d_blk_slot, d_blk_order, d_m_off, d_m_pairs, d_pairs;
  DevBuf d_keys, d_rows, d_row_blk, d_skeys, d_srows, d_sort_tmp, d_conn_off;
  DevBuf d_st_c, d_st_l, d_flags, d_pos, d_scan_tmp;
  DevBuf d_item_off, d_masks, d_mask_cnt, d_mask_pos;
  DevBuf d_rm_line, d_rm_act, d_rm_edges, d_rm_cnt;  // lt_ts_remerge_once: kept across the passes of a remerge
  std::vector<unsigned long long> h_rm_edges;
  std::vector<unsigned long long> h_rm_back;  // count + first edges of a remerge pass (one copy)
  std::vector<double> h_rm_in;                // host image of a pass's input (lines | active flags)
  DevBuf d_hcand, d_hlite;  // split host-side view of the candidates (debug read-outs), see materialize_compact
  DevBuf d_cand, d_lite, d_tri_off, d_score, d_best_idx, d_edge_flag, d_nvalid, d_edge_off, d_edges;
  DevBuf d_best_c, d_best_score, d_best_src, d_ntris, d_err;
  DevBuf d_blk_line_base, d_cnt_bl, d_st_key, d_wave_count, d_wave_pos, d_ntris_u, d_cand_node, d_pair_counter;
  int max_own_segs = 0;  // most segments of any image with a job (LDS table sizing)
  int max_nb_segs = 0;   // most segments of any neighbour image in the job (LDS table sizing)
  long long stat_pairs_eval = 0;
  std::vector<long long> h_blk_line_base;
  bool rows_sorted = true;   // every (image, neighbour) block lists its rows in non-decreasing line id
  // line-slot form of the matched pipeline (k_gates_ln, lt_kernels_v2.hip): decided per upload -- every block in the
  // compressed form, no run of equal line ids longer than gen_max_run(), neighbour tables within the LDS
  bool rows_ln = false;
  int ln_slots = 0;          // line slots per block (gen_slots_ln)
  DevBuf d_run_len, d_slot_row0, d_blk_nruns, d_ln_flag;
  // ... with stage B in rounds of 64 survivors of a block's dense survivor list (k_tri_rounds): survivors per block
  // (k_gates_ln's cursors), first round slot of every block (exclusive sum of ceil(rows / 64)), candidates per round
  DevBuf d_blk_surv, d_blk_rnd0, d_round_count;
  long long n_round_slots = 0;
  lt_host::HostBlock h_pinned_blk;
  long long *h_pinned = nullptr;  // pinned scratch for small device->host scalars
  DevBuf d_chunks, d_cand_meta, d_st_row, d_surv_count, d_seg_gates, d_blkrec, d_seg_vp, d_seg_has_vp;
  DevBuf d_needed;           // image indices the uploaded job references (own images and neighbours)
  int n_needed = 0;
  long long max_needed_segs = 0;
  DevBuf d_blk_chunk_off;    // exhaustive mode: per block, 64-line chunks of the image's earlier blocks
  int max_chunks = 1;        // exhaustive mode: most 64-line chunks of any neighbour
  DevBuf d_perm, d_rng;      // k_depth_order: depth-sorted candidate order of every node + per-position sweep range
  DevBuf d_result3;          // the run's result scalars (candidate count, error flag, pair statistic), one record
  DevBuf d_scan_status;      // k_node_prefix: ticket counter + per-tile look-back state
  DevBuf d_tile_order;       // k_score3: tile draw counters
  DevBuf d_tile_list;        // k_score3: tiles by cost class (kTileBuckets lists of cand_cap / 64 entries)
  DevBuf d_blk_vorder;       // k_gates_ln: the blocks in (neighbour, image) order
  bool blk_vorder_ok = false;
  DevBuf d_pc_cnt, d_pc_list;  // split scoring: the sweep's lists of tiles by pair count (counters 128 B apart | 8-byte entries)
  DevBuf d_sp_slots, d_sp_cnt, d_sp_ovf, d_sp_pairs, d_sp_desc;  // split scoring: per-tile slots of the pairs that pass the sweep, counts, overflow chunks
  bool score_two_kernels = false;  // k_score_q raised device flag 8 once: this context scores with k_score3<split> + k_dense8
  bool score_fused = false;      // the chunk store overflowed once (device flag 7): this context scores with the fused kernel
  DevBuf d_base_bl;          // exclusive prefix of cnt_bl over the neighbour blocks of a node
  // matched fast path: the candidate records stay in the staging lists of stage B; k_place writes only
  // place_perm[final position] = staging slot, and every consumer reads through it (LT_TEST_PLACE_COPY=1: the
  // records are moved into compact arrays instead, as the generic and exhaustive paths do)
  DevBuf d_place_perm;
  bool perm_mode = false;      // the last run left its candidates in the staging lists
  // one-pass exhaustive mode (k_gates_ex<true>): staging capacity as a fraction of the connections (0 = not measured
  // yet), adapted to the last run's yield; ex_two_pass: the next run uses the two-pass form (after an overflow)
  DevBuf d_ex_rec;             // record of every depth-sorted position (k_depth_order over staged records)
  DevBuf d_node_rec;           // per node: the scoring prologue record of its candidates (k_node_prefix -> k_cand_meta)
  DevBuf d_ex_ent;             // entry blocks of k_gates_ex<true> (8 B per staging slot)
  DevBuf d_ex_z;               // single-precision start depth of every staging slot (keys of k_depth_order)
  long long ex_region_cap = 0; // of the run in flight
  double ex_frac = 0.0;
  bool ex_two_pass = false;
  bool ex_staged_set[2] = {false, false};  // the run of event set 0 / 1 used the one-pass form
  bool in_run_async = false;
  bool pend_fine_gen = false, pend_fine_score = false;  // which per-kernel events the run in flight recorded
  int ex_retry_depth = 0;
  bool host_view_valid = false;  // d_hcand / d_hlite hold the last run's candidates
  bool compact_valid = false;  // d_cand / d_lite hold the compact arrays of the last run
  DevBuf d_tail_keys, d_tail_skeys, d_tail_sims, d_tail_mark, d_tail_pos, d_tail_recs, d_tail_nodes, d_tail_tmp, d_tail_keep, d_tail_kpos;  // lt_kernels_tail.hip
  bool cnt_bl_clean = false;  // d_cnt_bl is all zero (k_node_prefix cleans up after itself)
  size_t cnt_bl_bytes = 0;
  // point-guided proposals: per segment its (point3D_id, sfm row, x, y) records (24 B, kept as 3 doubles), CSR
  std::vector<double> h_seg_pts, h_sfm_xyz;
  std::vector<long long> h_seg_pt_off;
  std::vector<int> h_sfm_ids;
  DevBuf d_seg_pts, d_seg_pt_off, d_sfm_xyz;
  bool pts_ready = false, sfm_given = false, pts_dirty = false;
  long long max_seg_pts = 0;  // most point records of any segment (staging bound of the one-point proposal)
  bool vp_ready = false;  // InitVPResults was called for the current scene
  int n_chunks = 0;
  long long cand_cap = 0;
  long long C = 0, E = 0;  // candidates / valid edges of the last run
  // multi-GPU merge on the device (lt_shard_*): keys accumulated in d_tail_keys for the next tail (-1: the tail builds
  // them from the resident run), their capacity, this rank's own count
  long long shard_keys = -1, shard_keys_cap = 0, shard_own_keys = -1;

  // ---- host results (all nodes) ----
  // best candidate per node: a pooled host block (5.6 MB at 50 000 nodes; value-initialising a fresh vector of
  // that size costs more than the device run) -- an image's range is defined once best_c_set[image] is set
  lt_host::HostBlock best_c_blk;
  lt_host::HostBlock init_blk;  // page-locked copy of the scene lt_init uploads from (copies may still be in flight)
  Cand *best_c = nullptr;
  std::vector<char> best_c_set;
  lt_host::ZeroVec<double> best_score;
  lt_host::ZeroVec<int> best_src2, n_tris;
  lt_host::ZeroVec<unsigned char> has_best;
  lt_host::EdgeStore valid_edges;  // per node: flat (slot, ng_line) pairs
  // debug_mode: tris_ of EVERY batch (the reference keeps tris_ for all images, global_line_triangulator.cc:156-159);
  // records appended at download time, per node a range of the pool
  struct DebugTri {
    double line10[10], score;
    int src2[2];
  };
  std::vector<DebugTri> dbg_pool;
  lt_host::ZeroVec<long long> dbg_off;
  lt_host::ZeroVec<int> dbg_cnt;
  // ---- tail ----
  TrackStore tracks;
  // scratch of the tail, kept between calls (a fresh 200 KB vector costs more in page faults than the graph build)
  struct GEdge {
    double sim;
    int n1, n2;
  };
  // device half of a tail that is enqueued but not collected yet (lt_compute_tracks_begin .. _end)
  struct TailPending {
    bool active = false;
    long long E = 0;
    lt_host::HostBlock hb;
    size_t max_nodes = 0, o_pairs = 0, o_recs = 0, o_nodes = 0, o_flags = 0;
    int kb = 0;
    bool filtered = false;  // the node filter ran on the device: valid_flags come from d_outer_flags
  } tail_pend;
  DevBuf d_outer_flags;     // k_outer_filter: one byte per node (+ the "changed" word behind them)
  hipEvent_t ev_tail = nullptr;
  std::vector<int> tail_gmap;         // global node -> graph node, -1 outside a call (host form of the tail)
  std::vector<int> tail_kmap;         // device form: rank among the graph's nodes -> graph node
  std::vector<long long> tail_gnode;  // graph node -> global node
  std::vector<GEdge> tail_ge, tail_ge2;
  std::vector<int> tail_img_tmp;
  std::vector<int> tail_parent, tail_img_cnt, tail_labels, tail_nimg, tail_img_arena, tail_set_len;
  std::vector<long long> tail_set_off;
  std::vector<double> tail_cscore;   // device tail: score / record of every graph node, in graph-node order
  std::vector<lt::Cand> tail_ccand;
  std::vector<unsigned char> valid_flags;  // valid_flags_ of run_clustering (filterNodeByNumOuterEdges), per node
  bool tracks_done = false;
  long long stat_graph_nodes = 0, stat_graph_edges = 0, stat_pairs = 0;
  double timers[24] = {0};
  double timer_sums[24] = {0};  // over the runs since the last reset (lt_get_timer_sums)
  long long timer_runs = 0;
  long long stat_survivors = 0;  // connections that passed stage A (k_gates) in the last run
  hipEvent_t ev[14] = {nullptr};  // [0..11] timing, [12] end-of-run marker
  hipEvent_t ev_b[14] = {nullptr};  // second set, created when a run is enqueued behind one still in flight
  // end-of-run markers of pipelined runs: a ring of three, so that run k+1 can use run k's marker as its START event
  // (two events recorded back to back at the boundary of two runs cost two ~5 us bubbles in the stream, one is enough)
  // and still read it when it is finished -- during the enqueue of run k+2, which records the third one
  hipEvent_t ev_end[3] = {nullptr};
  unsigned run_seq = 0;
  hipEvent_t pend_ev_start = nullptr, pend_ev_end = nullptr;  // of the run in flight
  // Stage events of pipelined runs are SAMPLED (every event between two kernels is a ~5 us bubble in the stream): a run
  // enqueued behind one in flight carries them only every LT_TIMER_SAMPLE-th time (default 8); a run that starts on an
  // idle context always does.  The stage timers keep the last sampled values in between; the sums count sampled runs.
  unsigned async_seq = 0;
  bool pend_sampled = true;
  long long timer_stage_runs = 0;
  bool run_pending = false;         // lt_run_device_async left a run in flight (finish_run completes it)
  int pend_set = 0;                 // its event / pinned-slot set
  bool pend_count_on_device = false;
  long long pend_C = 0;
  int pend_ev_gen_end = 3, pend_ev_place_end = 4;
  long long C_last = 0;  // candidates of the last lt_run_device (known on the host once the scoring grid is sized)
};

#define HIPCHK(ctx, call)                                                                  \
  do {                                                                                     \
    hipError_t e_ = (call);                                                                \
    if (e_ != hipSuccess) {                                                                \
      (ctx)->err = std::string("HIP error: ") + hipGetErrorString(e_) + " at " #call;      \
      return LT_ERR_HIP;                                                                   \
    }                                                                                      \
  } while (0)

#define ENSURE(ctx, buf, bytes)                                                 \
  do {                                                                          \
    if (!(buf).ensure(bytes)) {                                                 \
      (ctx)->err = "hipMalloc failed for " #buf;                                \
      return LT_ERR_HIP;                                                        \
    }                                                                           \
  } while (0)

static inline int fail(lt_ctx *ctx, int code, const std::string &msg) {
  ctx->err = msg;
  return code;
}
