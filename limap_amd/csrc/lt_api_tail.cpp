// lt_api_tail.cpp -- C ABI, part 4: ComputeLineTracks (global_line_triangulator.cc:168-351) -- edge set and similarities on
// the device (lt_kernels_tail.hip), union-find labels, tracks and aggregation on the host (lt_tail.h).
#include "lt_host.h"

using namespace lt;
using namespace lt_impl;

extern "C" {

// ---------------------------------------------------------------------------------------------
// host tail
// ---------------------------------------------------------------------------------------------
// Device half of the tail (lt_kernels_tail.hip): possible when the results of the whole scene are those of the run
// that is still resident in HBM (one batch, nothing imported, nothing read back yet) and no node filter applies
// The node filter (min_num_outer_edges > 0, global_line_triangulator.cc:168-232) runs on the device too since round 5
// (k_outer_filter) -- since round 6 also over imported shards: with the filter on a shard ships DIRECTED keys
// (source << kb | target) and the filter runs over the merged key list (k_outer_*_keys).  LT_TAIL_HOST=1 forces the host form.
static bool tail_on_device(const lt_ctx *ctx) {
  if (test_switch("LT_TAIL_HOST") != nullptr) return false;
  if (!ctx->inited || ctx->job_mode == 0 || ctx->downloaded || ctx->job_imgs.empty()) return false;
  if (ctx->G <= 0 || ctx->G >= (1ll << 31)) return false;
  // results that live only on the host -- an earlier batch that was read back, imported shards -- rule the device form
  // out; host copies of the CURRENT job's images do not (the resident run supersedes them: the same job run again after
  // its results were read), nor do images that merely hold the value-initialised candidate (best_c_set == 2)
  bool foreign = false;
  for (char c : ctx->best_c_set) foreign = foreign || c == 1;
  if (!foreign) return true;
  std::vector<char> in_job((size_t)ctx->n_img, 0);
  for (int idx : ctx->job_imgs) in_job[(size_t)idx] = 1;
  for (int i = 0; i < ctx->n_img; ++i)
    if (ctx->best_c_set[(size_t)i] == 1 && !in_job[(size_t)i]) return false;
  return true;
}

// number of directed valid edges of the resident run = keys it contributes (one scan of the per-node counts + a sync)
static int tail_count_keys(lt_ctx *ctx, long long *E_out) {
  hipStream_t st = ctx->stream;
  const long long G = ctx->G;
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
  HIPCHK(ctx, hipStreamSynchronize(st));
  *E_out = hp[0];
  return LT_OK;
}
// keys (min node << kb | max node; directed: source << kb | target) of the resident run's valid edges -> d_tail_keys[0 .. E),
// in node / candidate order
static void tail_build_keys(lt_ctx *ctx, int kb, bool directed = false) {
  launch_tail_keys(ctx->stream, ctx->G, ctx->d_tri_off.as<long long>(), ctx->d_edge_flag.as<unsigned>(),
                   ctx->d_edge_off.as<long long>(), ctx->perm_mode ? ctx->d_st_c.as<CRec>() : ctx->d_cand.as<CRec>(),
                   ctx->d_seg_off.as<long long>(), kb, ctx->d_tail_keys.as<unsigned long long>(),
                   ctx->perm_mode ? ctx->d_place_perm.as<unsigned>() : nullptr, directed ? 1 : 0);
}
// a shard's keys are directed iff the node filter is on (every rank of a job has the same configuration)
static bool shard_keys_directed(const lt_ctx *ctx) { return ctx->cfg.min_num_outer_edges > 0; }

// sorted unique undirected edges + their similarities; the graph nodes' best candidates land in ctx->best_c etc.
// Two host synchronisations: one for the number of valid edges (it sizes the sort), one at the end; the graph
// nodes' records are written by the gather kernel straight into page-locked host memory.
// Two halves since round 5: tail_device_enqueue leaves the device half of the tail in the stream behind the run it
// belongs to (one host synchronisation inside, for the number of valid edges that sizes the sort) and records an event;
// tail_device_collect waits for THAT EVENT only -- a later run may already be enqueued behind it -- and builds the graph
// from the page-locked results.  lt_compute_tracks calls both; lt_compute_tracks_begin / _end let a caller that streams
// steps put the next step's kernels between them, so that the host half of step k's tail runs while the device is busy
// with step k + 1.
extern "C++" {
static int tail_device_enqueue(lt_ctx *ctx) {
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
  const size_t scan_tmp = scan_temp_bytes_u32_to_i64(G + 1);
  // the undirected edge keys: of the resident run (counted and built here), or -- shards of other ranks were imported
  // (lt_shard_*) -- the list that is already complete in d_tail_keys
  const bool merged = ctx->shard_keys >= 0;
  ENSURE(ctx, ctx->d_scan_tmp, std::max<size_t>(scan_tmp, 16));
  long long E = 0;
  if (merged) {
    E = ctx->shard_keys;
    ctx->shard_keys = -1;
  } else {
    int rck = tail_count_keys(ctx, &E);
    if (rck) return rck;
  }
  ENSURE(ctx, ctx->d_tail_mark, 4 * (size_t)(G + 1)); ENSURE(ctx, ctx->d_tail_pos, 8 * (size_t)(G + 1));
  HIPCHK(ctx, hipMemsetAsync(ctx->d_tail_mark.p, 0, 4 * (size_t)(G + 1), st));
  lap("scan + sync (E)");
  ctx->E = E;
  ctx->C = ctx->C_last;
  lt_ctx::TailPending &tp_ = ctx->tail_pend;
  tp_ = lt_ctx::TailPending();
  tp_.E = E;
  if (E <= 0) {
    // no valid edge anywhere: with a node filter every node falls short of min_num_outer_edges
    if (ctx->cfg.min_num_outer_edges > 0) ctx->valid_flags.assign((size_t)G, 0);
    tp_.active = true;
    return LT_OK;
  }
  // the tail counts as "in flight" only once its event is recorded.  Every error return before that point
  // releases the page-locked block and leaves tail_pend cleared, so a retry starts from scratch instead of collecting
  // from a block that was never filled.
  struct PendGuard {
    lt_ctx *c;
    bool armed = true;
    ~PendGuard() {
      if (!armed) return;
      if (c->tail_pend.hb.p) lt_host::host_block_release(c->tail_pend.hb);
      c->tail_pend = lt_ctx::TailPending();
    }
  } guard{ctx};
  const size_t En = (size_t)E;
  if (!merged) ENSURE(ctx, ctx->d_tail_keys, 8 * En);
  ENSURE(ctx, ctx->d_tail_skeys, 8 * En); ENSURE(ctx, ctx->d_tail_sims, 8 * En);
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
  // with the node filter the per-node flags travel in the same block (copied on the context stream in front of the
  // event: tail_device_collect needs no device copy of its own and cannot wait on a run enqueued behind the tail)
  const bool with_filter = ctx->cfg.min_num_outer_edges > 0;
  const size_t o_flags = (o_nodes + 4 * max_nodes + 63) / 64 * 64;
  lt_host::HostBlock hb = lt_host::host_block_acquire(o_flags + (with_filter ? (size_t)G : 0));
  if (!hb.p) return fail(ctx, LT_ERR_RUNTIME, "out of host memory for the edge list");
  tp_.hb = hb;  // released by tail_device_collect (or by the guard above on an error return)
  tp_.max_nodes = max_nodes; tp_.o_pairs = o_pairs; tp_.o_recs = o_recs; tp_.o_nodes = o_nodes; tp_.kb = kb;
  tp_.o_flags = o_flags;
  char *base = (char *)hb.p;
  long long *hn = (long long *)base;  // [0] graph nodes, [1] graph edges
  hn[0] = hn[1] = 0;
  if (!merged) tail_build_keys(ctx, kb);
  // filterNodeByNumOuterEdges over a MERGED key list (the shards shipped directed keys): passes of count / apply until one
  // changes nothing, then the keys take their undirected form for the sort
  const unsigned char *d_flags = nullptr;
  if (with_filter && merged) {
    ENSURE(ctx, ctx->d_outer_flags, (size_t)G + 64);
    ENSURE(ctx, ctx->d_tail_pos, 8 * (size_t)(G + 1));  // (free until the scan below: the per-node counters live here)
    unsigned char *fl = ctx->d_outer_flags.as<unsigned char>();
    int *d_changed = reinterpret_cast<int *>(fl + (((size_t)G + 15) / 16) * 16);
    unsigned *counts = ctx->d_tail_pos.as<unsigned>();
    HIPCHK(ctx, hipMemsetAsync(fl, 1, (size_t)G, st));
    HIPCHK(ctx, hipMemsetAsync(counts, 0, 4 * (size_t)G, st));
    for (int round = 0; round < (1 << 20); ++round) {
      HIPCHK(ctx, hipMemsetAsync(d_changed, 0, 4, st));
      for (int k = 0; k < 4; ++k)
        launch_outer_pass_keys(st, E, ctx->d_tail_keys.as<unsigned long long>(), kb, G, counts, ctx->cfg.min_num_outer_edges,
                               fl, d_changed);
      int changed = 0;
      HIPCHK(ctx, hipMemcpyAsync(&changed, d_changed, 4, hipMemcpyDeviceToHost, st));
      HIPCHK(ctx, hipStreamSynchronize(st));
      if (!changed) break;
    }
    launch_keys_undirect(st, E, ctx->d_tail_keys.as<unsigned long long>(), kb);
    d_flags = fl;
    tp_.filtered = true;
    HIPCHK(ctx, hipMemcpyAsync(base + o_flags, fl, (size_t)G, hipMemcpyDeviceToHost, st));
  }
  if (launch_tail_sort(st, ctx->d_tail_tmp.p, sort_tmp, E, ctx->d_tail_keys.as<unsigned long long>(),
                       ctx->d_tail_skeys.as<unsigned long long>(), end_bit) != 0)
    return fail(ctx, LT_ERR_HIP, "rocprim radix sort failed");
  // filterNodeByNumOuterEdges (:168-232) on the resident run: passes of k_outer_filter until one changes nothing (the
  // flag comes back every four passes; a scene needs a handful)
  if (with_filter && !merged) {
    ENSURE(ctx, ctx->d_outer_flags, (size_t)G + 64);
    unsigned char *fl = ctx->d_outer_flags.as<unsigned char>();
    int *d_changed = reinterpret_cast<int *>(fl + (((size_t)G + 15) / 16) * 16);
    HIPCHK(ctx, hipMemsetAsync(fl, 1, (size_t)G, st));
    for (int round = 0; round < (1 << 20); ++round) {
      HIPCHK(ctx, hipMemsetAsync(d_changed, 0, 4, st));
      for (int k = 0; k < 4; ++k)
        launch_outer_filter(st, G, ctx->d_tri_off.as<long long>(), ctx->d_edge_flag.as<unsigned>(),
                            ctx->perm_mode ? ctx->d_st_c.as<CRec>() : ctx->d_cand.as<CRec>(), ctx->d_seg_off.as<long long>(),
                            ctx->perm_mode ? ctx->d_place_perm.as<unsigned>() : nullptr, ctx->cfg.min_num_outer_edges, fl,
                            d_changed);
      int changed = 0;
      HIPCHK(ctx, hipMemcpyAsync(&changed, d_changed, 4, hipMemcpyDeviceToHost, st));
      HIPCHK(ctx, hipStreamSynchronize(st));
      if (!changed) break;
    }
    d_flags = fl;
    tp_.filtered = true;
    HIPCHK(ctx, hipMemcpyAsync(base + o_flags, fl, (size_t)G, hipMemcpyDeviceToHost, st));
  }
  LinkCfg3 l3 = make_l3(ctx->cfg);
  l3.use_angle = 1; l3.use_overlap = 1; l3.use_perp = 0; l3.use_innerseg = 1; l3.use_scaleinv = 0;  // line_linker.h:123-129
  launch_tail_sims(st, E, ctx->d_tail_skeys.as<unsigned long long>(), ctx->d_ntris.as<int>(), ctx->d_best_c.as<Cand>(), l3,
                   kb, ctx->d_tail_sims.as<double>(), ctx->d_tail_mark.as<unsigned>(), ctx->d_tail_keep.as<unsigned>(), d_flags);
  if (launch_scan_u32_to_i64(st, ctx->d_tail_tmp.p, scan_tmp2, E + 1, ctx->d_tail_keep.as<unsigned>(),
                             ctx->d_tail_kpos.as<long long>()) != 0 ||
      launch_scan_u32_to_i64(st, ctx->d_scan_tmp.p, scan_tmp, G + 1, ctx->d_tail_mark.as<unsigned>(),
                             ctx->d_tail_pos.as<long long>()) != 0)
    return fail(ctx, LT_ERR_HIP, "rocprim scan failed");
  if (hb.pinned) {  // the kernels write across PCIe: the host needs no size before the copies
    launch_tail_compact(st, E, ctx->d_tail_skeys.as<unsigned long long>(), ctx->d_tail_sims.as<double>(),
                        ctx->d_tail_keep.as<unsigned>(), ctx->d_tail_kpos.as<long long>(), base + o_pairs, hn + 1,
                        ctx->d_tail_pos.as<long long>(), kb);
    launch_tail_gather(st, G, ctx->d_tail_mark.as<unsigned>(), ctx->d_tail_pos.as<long long>(), ctx->d_best_c.as<Cand>(),
                       ctx->d_best_score.as<double>(), ctx->d_best_src.as<int>(), base + o_recs, (int *)(base + o_nodes), hn);
  } else {  // no page-locked memory: pack on the device, copy the bounds
    ENSURE(ctx, ctx->d_tail_recs, 16 * En + tail_rec_bytes() * max_nodes + 64); ENSURE(ctx, ctx->d_tail_nodes, 4 * max_nodes);
    char *dp = (char *)ctx->d_tail_recs.p;
    launch_tail_compact(st, E, ctx->d_tail_skeys.as<unsigned long long>(), ctx->d_tail_sims.as<double>(),
                        ctx->d_tail_keep.as<unsigned>(), ctx->d_tail_kpos.as<long long>(), dp, (long long *)ctx->d_tail_keys.p,
                        ctx->d_tail_pos.as<long long>(), kb);
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
  if (!ctx->ev_tail) HIPCHK(ctx, hipEventCreateWithFlags(&ctx->ev_tail, hipEventDisableTiming));
  HIPCHK(ctx, hipEventRecord(ctx->ev_tail, st));
  tp_.active = true;
  guard.armed = false;
  lap("enqueue");
  return LT_OK;
}

template <class AddEdge>
static int tail_device_collect(lt_ctx *ctx, AddEdge &&add_edge) {
  static const bool trace = getenv("LT_TAIL_TRACE") != nullptr;
  double tp = now_ms();
  auto lap = [&](const char *what) {
    if (!trace) return;
    double t = now_ms();
    fprintf(stderr, "[tail]   %-16s %.3f ms\n", what, t - tp);
    tp = t;
  };
  lt_ctx::TailPending &tp_ = ctx->tail_pend;
  if (!tp_.active) return fail(ctx, LT_ERR_STATE, "internal: no device tail in flight");
  tp_.active = false;
  const long long E = tp_.E;
  if (E <= 0) return LT_OK;
  if (!tp_.hb.p) return fail(ctx, LT_ERR_STATE, "internal: device tail in flight without its host block");
  struct Rel {
    lt_host::HostBlock b;
    ~Rel() { lt_host::host_block_release(b); }
  } rel{tp_.hb};
  tp_.hb = lt_host::HostBlock();
  const size_t max_nodes = tp_.max_nodes, o_pairs = tp_.o_pairs, o_recs = tp_.o_recs, o_nodes = tp_.o_nodes;
  (void)tp_.kb;
  char *base = (char *)rel.b.p;
  long long *hn = (long long *)base;
  const unsigned long long *hpairs = (const unsigned long long *)(base + o_pairs);
  HIPCHK(ctx, hipSetDevice(ctx->device));
  HIPCHK(ctx, hipEventSynchronize(ctx->ev_tail));
  if (tp_.filtered)  // valid_flags_ of the reference (GetAllValidBestTris ... read them): copied in front of the event
    std::memcpy(ctx->valid_flags.data(), base + tp_.o_flags, (size_t)ctx->G);
  lap("sync");
  const long long Nm = hn[0], Ne = hn[1];
  if (Nm < 0 || (size_t)Nm > max_nodes || Ne < 0 || Ne > E)
    return fail(ctx, LT_ERR_RUNTIME, "internal: graph size out of range");
  // graph nodes in edge order (base/graph.cc:57-87): the edges name their endpoints by RANK among the graph's nodes (the
  // order of `nodes` / `recs` below), so the node map has one entry per graph node, not per node of the scene
  const int *nodes = (const int *)(base + o_nodes);
  std::vector<int> &kmap = ctx->tail_kmap;  // rank -> graph node
  kmap.assign((size_t)Nm, -1);
  for (long long i = 0; i < Ne; ++i) {  // distinct keys with score != 0 (:284-285), in std::set order
    double sim;
    std::memcpy(&sim, &hpairs[2 * i + 1], 8);
    const unsigned long long ka = hpairs[2 * i] >> 32, kb2 = hpairs[2 * i] & 0xFFFFFFFFull;
    if (ka >= (unsigned long long)Nm || kb2 >= (unsigned long long)Nm)
      return fail(ctx, LT_ERR_RUNTIME, "internal: graph edge names a node outside the graph");
    add_edge(kmap[(size_t)ka], (long long)nodes[ka], kmap[(size_t)kb2], (long long)nodes[kb2], sim);
  }
  lap("graph");
  struct Rec {
    Cand c;
    double score;
    int src[2];
  };
  static_assert(sizeof(Rec) == 128, "TailRec layout");
  const Rec *recs = (const Rec *)(base + o_recs);
  // the same records once more in graph-node order (what the union-find and the aggregation read: the G-entry tables are
  // 100 MB at a million nodes, every access a miss)
  const size_t n_graph = ctx->tail_gnode.size();
  ctx->tail_nimg.resize(n_graph);
  ctx->tail_cscore.resize(n_graph);
  ctx->tail_ccand.resize(n_graph);
  lt_host::pool_for(Nm, 1024, [&](long long k0, long long k1) {
    for (long long k = k0; k < k1; ++k) {  // distinct nodes: no two iterations touch the same entry
      const long long g = nodes[k];
      ctx->best_c[g] = recs[k].c;
      ctx->best_score[g] = recs[k].score;
      ctx->best_src2[2 * g] = ctx->img_ids[recs[k].src[0]];
      ctx->best_src2[2 * g + 1] = recs[k].src[1];
      ctx->has_best[g] = 1;
      const int gi = kmap[(size_t)k];
      if (gi >= 0) {
        ctx->tail_nimg[(size_t)gi] = ctx->h_node_img[g];
        ctx->tail_cscore[(size_t)gi] = recs[k].score;
        ctx->tail_ccand[(size_t)gi] = recs[k].c;
      }
    }
  });
  lap("host unpack");
  return LT_OK;
}
}  // extern "C++"

int lt_compute_tracks_begin(lt_ctx *ctx) {
  LT_RANGE("lt_compute_tracks_begin (device half of the tail enqueued)");
  if (ctx->cfg.merging_strategy < 0 || ctx->cfg.merging_strategy > 2)  // global_line_triangulator.cc:314-316
    return fail(ctx, LT_ERR_RUNTIME, "Error!The given merging strategy is not implemented");
  if (ctx->tail_pend.active) return fail(ctx, LT_ERR_STATE, "lt_compute_tracks_begin: a tail is already in flight");
  if (!tail_on_device(ctx))
    return fail(ctx, LT_ERR_STATE, "lt_compute_tracks_begin needs the device form of the tail (results of the run "
                                   "resident on the device; with imported shards min_num_outer_edges == 0): call "
                                   "lt_compute_tracks instead");
  lt_host::SpinPool::get(lt_host::row_workers()).wake();
  int rc;
  if (!ctx->uploaded && (rc = lt_upload(ctx))) return rc;
  if (!ctx->ran && (rc = lt_run_device(ctx))) return rc;
  ctx->valid_flags.assign((size_t)ctx->G, 1);
  return tail_device_enqueue(ctx);
}
int lt_compute_tracks_end(lt_ctx *ctx) {
  if (!ctx->tail_pend.active) return fail(ctx, LT_ERR_STATE, "lt_compute_tracks_end without lt_compute_tracks_begin");
  return lt_compute_tracks(ctx);
}

int lt_compute_tracks(lt_ctx *ctx) {
  LT_RANGE("lt_compute_tracks (tail: edge set, similarities, union-find, aggregation)");
  if (ctx->cfg.merging_strategy < 0 || ctx->cfg.merging_strategy > 2)  // global_line_triangulator.cc:314-316
    return fail(ctx, LT_ERR_RUNTIME, "Error!The given merging strategy is not implemented");
  const bool in_flight = ctx->tail_pend.active;  // lt_compute_tracks_begin ran: the device half is (being) done
  const bool on_device = in_flight || tail_on_device(ctx);
  if (ctx->shard_keys >= 0 && !on_device) {
    ctx->shard_keys = -1;
    return fail(ctx, LT_ERR_STATE, "shards were imported on the device (lt_shard_import), but the device form of the tail "
                                   "is not available (LT_TAIL_HOST, or results already read back)");
  }
  lt_host::SpinPool::get(lt_host::row_workers()).wake();  // the host half of the tail shares its loops with the team
  int rc;
  if (in_flight) {
    // (a later run may be enqueued behind the tail: nothing is flushed or waited for here)
  } else if (on_device) {
    if (!ctx->uploaded && (rc = lt_upload(ctx))) return rc;
    if (!ctx->ran && (rc = lt_run_device(ctx))) return rc;
  } else {
    if ((rc = lt_flush(ctx))) return rc;
    if (ctx->inited) define_best_of_other_images(ctx);
  }
  double t0 = now_ms();
  static const bool tail_trace = getenv("LT_TAIL_TRACE") != nullptr;  // developer: stage times to stderr
  double tprev = t0;
  // the stages' times also go to the context (lt_get_timers slots 22 / 23: device half + graph, edge order + union-find; the
  // rest of slot 10 is members + aggregation)
  double acc_dev = 0.0, acc_uf = 0.0;
  auto lap = [&](const char *what) {
    double t = now_ms();
    const double d = t - tprev;
    tprev = t;
    if (!std::strncmp(what, "device", 6) || !std::strncmp(what, "graph", 5) || !std::strncmp(what, "edge sims", 9)) acc_dev += d;
    else if (!std::strncmp(what, "edge sort", 9) || !std::strncmp(what, "union", 5) || !std::strncmp(what, "  uf", 4)) acc_uf += d;
    if (tail_trace) fprintf(stderr, "[tail] %-18s %.3f ms\n", what, d);
  };
  const long long G = ctx->G;
  // graph in edge order (base/graph.cc:57-87); scratch kept in the context
  using GEdge = lt_ctx::GEdge;
  std::vector<int> &gmap = ctx->tail_gmap;            // global node -> graph node
  std::vector<long long> &gnode = ctx->tail_gnode;    // graph node -> global node
  std::vector<GEdge> &ge = ctx->tail_ge;
  // (the device form names the graph's nodes by rank: it needs no map over the scene's nodes)
  if (!on_device && (long long)gmap.size() != G) gmap.assign((size_t)G, -1);
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
    if (!in_flight) {
      ctx->valid_flags.assign((size_t)G, 1);
      if ((rc = tail_device_enqueue(ctx))) return rc;
    }
    auto add_edge_ranked = [&](int &ma, long long ga, int &mb, long long gb, double sim) {
      if (ma < 0) {
        ma = (int)gnode.size();
        gnode.push_back(ga);
      }
      if (mb < 0) {
        mb = (int)gnode.size();
        gnode.push_back(gb);
      }
      ge.push_back(GEdge{sim, ma, mb});
    };
    if ((rc = tail_device_collect(ctx, add_edge_ranked))) return rc;
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
  // (the per-node arrays of the union-find live in the context: fresh vectors of a few MB are page faults per call)
  std::vector<int> &parent = ctx->tail_parent;
  parent.assign((size_t)n_nodes, -1);
  // images_in_track (std::set<int> per root in the reference, :52-84): only the set SIZES steer the union.  A root's
  // images as a small sorted list, made when it first absorbs another root (a singleton is its node's image): a bit set
  // per node over the image indices -- rounds 1-4 -- is n_nodes x n_images / 8 bytes to clear per call, 13 MB and most of
  // the 1.0-1.8 ms this loop took at 1000 images.
  std::vector<int> &img_cnt = ctx->tail_img_cnt;
  img_cnt.assign((size_t)n_nodes, 1);
  std::vector<int> &nimg = ctx->tail_nimg;  // image of every graph node (tail_from_device fills it from the records)
  if (!on_device) {
    nimg.resize((size_t)n_nodes);
    for (int i = 0; i < n_nodes; ++i) nimg[(size_t)i] = ctx->h_node_img[gnode[(size_t)i]];
  }
  // the lists live in ONE arena (a vector per root cost an allocation per union: most of the 50 ns a union took): a union
  // writes the merged list behind everything else; the dead lists are swept when the arena has grown past a few times
  // the live ones
  std::vector<int> &arena = ctx->tail_img_arena;
  std::vector<long long> &set_off = ctx->tail_set_off;  // root -> first entry of its list, -1: the singleton {nimg[root]}
  std::vector<int> &set_len = ctx->tail_set_len;
  arena.clear();
  set_off.assign((size_t)n_nodes, -1);
  set_len.assign((size_t)n_nodes, 0);
  size_t sweep_at = std::max<size_t>((size_t)1 << 16, 8 * (size_t)n_nodes);
  auto absorb = [&](int dst, int src) {  // images[dst] |= images[src]; images[src] = {}
    const size_t na = set_off[(size_t)dst] >= 0 ? (size_t)set_len[(size_t)dst] : 1;
    const size_t nb = set_off[(size_t)src] >= 0 ? (size_t)set_len[(size_t)src] : 1;
    if (arena.size() + na + nb > sweep_at) {  // sweep: the live lists move to the front of a fresh arena
      std::vector<int> &fresh = ctx->tail_img_tmp;
      fresh.clear();
      for (int r = 0; r < n_nodes; ++r)
        if (set_off[(size_t)r] >= 0 && img_cnt[(size_t)r] > 0) {
          const long long o = set_off[(size_t)r];
          set_off[(size_t)r] = (long long)fresh.size();
          fresh.insert(fresh.end(), arena.begin() + o, arena.begin() + o + set_len[(size_t)r]);
        }
      arena.swap(fresh);
      sweep_at = std::max(sweep_at, 4 * arena.size() + na + nb);
    }
    const size_t o_new = arena.size();
    arena.resize(o_new + na + nb);
    const int *a = set_off[(size_t)dst] >= 0 ? arena.data() + set_off[(size_t)dst] : &nimg[(size_t)dst];
    const int *b = set_off[(size_t)src] >= 0 ? arena.data() + set_off[(size_t)src] : &nimg[(size_t)src];
    int *out = arena.data() + o_new;
    const size_t n = (size_t)(std::set_union(a, a + na, b, b + nb, out) - out);
    arena.resize(o_new + n);
    set_off[(size_t)dst] = (long long)o_new;
    set_len[(size_t)dst] = (int)n;
    set_off[(size_t)src] = -1;
    img_cnt[(size_t)dst] = (int)n;
    img_cnt[(size_t)src] = 0;
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
  lap("  uf: unions");
  std::vector<int> &labels = ctx->tail_labels;
  labels.assign((size_t)n_nodes, -1);
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
      const int img = nimg[(size_t)i];
      const size_t w = (size_t)wr[(size_t)tl]++;
      ts.node_ids[w] = i;
      ts.img_ids[w] = ctx->img_ids[img];
      ts.line_ids[w] = (int)(g - ctx->seg_off[img]);
      ts.scores[w] = on_device ? ctx->tail_cscore[(size_t)i] : ctx->best_score[g];
      ts.gnodes[w] = g;
    }
    lap("  tracks: members");
    // shared with the workers of the persistent team that are awake (lt_pool.h): a few thousand tracks aggregate
    // faster than a sleeping OpenMP team starts on a big host
    lt_host::pool_for((long long)nT, 16, [&](long long t0_, long long t1_) {
      static thread_local AggScratch scratch;
      for (long long t = t0_; t < t1_; ++t) {
        const size_t a = (size_t)ts.off[(size_t)t], n = (size_t)ts.off[(size_t)t + 1] - a;
        // (device tail: the graph nodes' records in graph-node order, 2-3 MB, instead of the G-entry table)
        if (on_device)
          aggregate(ctx->tail_ccand.data(), ts.node_ids.data() + a, ts.scores.data() + a, (int)n, ctx->cfg.num_outliers_aggregator,
                    ts.line7.data() + 7 * (size_t)t, scratch);
        else
          aggregate(ctx->best_c, ts.gnodes.data() + a, ts.scores.data() + a, (int)n, ctx->cfg.num_outliers_aggregator,
                    ts.line7.data() + 7 * (size_t)t, scratch);
      }
    });
  }
  if (!on_device)
    for (long long g : gnode) gmap[(size_t)g] = -1;  // leave the scratch map clean
  lap("tracks+aggregate");
  ctx->tracks_done = true;
  ctx->timers[10] = now_ms() - t0;
  ctx->timers[22] = acc_dev;
  ctx->timers[23] = acc_uf;
  return LT_OK;
}


// ---------------------------------------------------------------------------------------------
// Shards of a multi-GPU run, device to device (round 4).  Images are sharded over the ranks in id order, so a
// rank's nodes are ONE range [g_lo, g_hi) of the global node index and its per-node results are slices of the arrays the
// device tail reads.  What rank 0 needs of another rank is therefore
//   nodes blob  [n x 112 B best candidate | n x 8 B score | n x 8 B source (image index, line) | n x 4 B candidate count]
//   keys blob   the rank's valid-edge keys, 8 B each: undirected (min node << kb | max node: G is global, so is kb), or --
//               with the node filter on (min_num_outer_edges > 0) -- directed (source << kb | target), from which rank 0
//               runs the filter over the whole scene before it brings them to the undirected form,
// both produced and consumed by copies / kernels on the devices -- no per-image export, no Python loop, no host tail.
// Protocol: every rank lt_shard_count -> (the counts travel) -> lt_shard_build(total on rank 0) -> the others
// lt_shard_export into the collective's send buffers -> gather -> rank 0 lt_shard_import per rank -> lt_compute_tracks.
// Pointers may be device or host memory (hipMemcpyDefault): a gloo group gathers host tensors.
// ---------------------------------------------------------------------------------------------
int lt_shard_node_bytes(void) { return (int)(sizeof(Cand) + 8 + 8 + 4); }

int lt_shard_count(lt_ctx *ctx, int64_t *n_keys) {
  LT_FINISH(ctx);
  if (!ctx->ran || ctx->downloaded) return fail(ctx, LT_ERR_STATE, "lt_shard_count needs the results of a run resident on the device");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  long long E = 0;
  int rc = tail_count_keys(ctx, &E);
  if (rc) return rc;
  ctx->shard_own_keys = E;
  *n_keys = E;
  return LT_OK;
}

int lt_shard_build(lt_ctx *ctx, int64_t total_keys) {
  if (ctx->shard_own_keys < 0) return fail(ctx, LT_ERR_STATE, "lt_shard_build before lt_shard_count");
  if (total_keys < ctx->shard_own_keys) return fail(ctx, LT_ERR_ARGUMENT, "lt_shard_build: fewer keys than this rank's own");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  ENSURE(ctx, ctx->d_tail_keys, 8 * (size_t)std::max<long long>(total_keys, 1));
  if (ctx->shard_own_keys > 0) tail_build_keys(ctx, bits_for(ctx->G + 1), shard_keys_directed(ctx));
  HIPCHK(ctx, hipGetLastError());
  ctx->shard_keys = ctx->shard_own_keys;  // imports append behind them
  ctx->shard_keys_cap = total_keys;
  return LT_OK;
}

static int shard_range_ok(lt_ctx *ctx, int64_t g_lo, int64_t g_hi) {
  if (g_lo < 0 || g_hi < g_lo || g_hi > ctx->G) return fail(ctx, LT_ERR_ARGUMENT, "node range out of bounds");
  return LT_OK;
}

int lt_shard_export(lt_ctx *ctx, int64_t g_lo, int64_t g_hi, void *nodes_blob, void *keys_blob) {
  int rc = shard_range_ok(ctx, g_lo, g_hi);
  if (rc) return rc;
  if (ctx->shard_keys < 0) return fail(ctx, LT_ERR_STATE, "lt_shard_export before lt_shard_build");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  const size_t n = (size_t)(g_hi - g_lo);
  char *out = static_cast<char *>(nodes_blob);
  if (n > 0) {
    HIPCHK(ctx, hipMemcpyAsync(out, ctx->d_best_c.as<Cand>() + g_lo, sizeof(Cand) * n, hipMemcpyDefault, st));
    HIPCHK(ctx, hipMemcpyAsync(out + sizeof(Cand) * n, ctx->d_best_score.as<double>() + g_lo, 8 * n, hipMemcpyDefault, st));
    HIPCHK(ctx, hipMemcpyAsync(out + (sizeof(Cand) + 8) * n, ctx->d_best_src.as<int>() + 2 * g_lo, 8 * n, hipMemcpyDefault, st));
    HIPCHK(ctx, hipMemcpyAsync(out + (sizeof(Cand) + 16) * n, ctx->d_ntris.as<int>() + g_lo, 4 * n, hipMemcpyDefault, st));
  }
  if (ctx->shard_own_keys > 0)
    HIPCHK(ctx, hipMemcpyAsync(keys_blob, ctx->d_tail_keys.p, 8 * (size_t)ctx->shard_own_keys, hipMemcpyDefault, st));
  HIPCHK(ctx, hipStreamSynchronize(st));  // the collective that follows runs on another stream
  ctx->shard_keys = -1;  // this rank's part is done: it does not run a merged tail itself
  return LT_OK;
}

int lt_shard_import(lt_ctx *ctx, int64_t g_lo, int64_t g_hi, const void *nodes_blob, int64_t n_keys, const void *keys_blob) {
  int rc = shard_range_ok(ctx, g_lo, g_hi);
  if (rc) return rc;
  if (ctx->shard_keys < 0) return fail(ctx, LT_ERR_STATE, "lt_shard_import before lt_shard_build");
  if (n_keys < 0 || ctx->shard_keys + n_keys > ctx->shard_keys_cap)
    return fail(ctx, LT_ERR_ARGUMENT, "lt_shard_import: more keys than lt_shard_build reserved");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  const size_t n = (size_t)(g_hi - g_lo);
  const char *in = static_cast<const char *>(nodes_blob);
  if (n > 0) {
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_best_c.as<Cand>() + g_lo, in, sizeof(Cand) * n, hipMemcpyDefault, st));
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_best_score.as<double>() + g_lo, in + sizeof(Cand) * n, 8 * n, hipMemcpyDefault, st));
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_best_src.as<int>() + 2 * g_lo, in + (sizeof(Cand) + 8) * n, 8 * n, hipMemcpyDefault, st));
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_ntris.as<int>() + g_lo, in + (sizeof(Cand) + 16) * n, 4 * n, hipMemcpyDefault, st));
  }
  int bad = 0;
  if (n_keys > 0) {
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_tail_keys.as<unsigned long long>() + ctx->shard_keys, keys_blob, 8 * (size_t)n_keys,
                               hipMemcpyDefault, st));
    // the keys index the per-node arrays in the similarity kernel: both ids must be nodes of this scene
    ENSURE(ctx, ctx->d_err, sizeof(int));
    HIPCHK(ctx, hipMemsetAsync(ctx->d_err.p, 0, sizeof(int), st));
    launch_check_keys(st, n_keys, ctx->d_tail_keys.as<unsigned long long>() + ctx->shard_keys, bits_for(ctx->G + 1), ctx->G,
                      ctx->d_err.as<int>(), shard_keys_directed(ctx) ? 1 : 0);
    HIPCHK(ctx, hipMemcpyAsync(&bad, ctx->d_err.p, sizeof(int), hipMemcpyDeviceToHost, st));
  }
  HIPCHK(ctx, hipStreamSynchronize(st));  // the source buffers belong to the caller
  if (bad) return fail(ctx, LT_ERR_ARGUMENT, "lt_shard_import: a key names a node outside this scene (or is not min << kb | max)");
  ctx->shard_keys += n_keys;
  ctx->tracks_done = false;
  return LT_OK;
}

}  // extern "C"
