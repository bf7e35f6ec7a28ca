// lt_api_query.cpp -- C ABI, part 5: getters (bindings.cc:100-119), per-image export / import for the multi-GPU merge,
// statistics and timers, the free functions of triangulation/functions.h.
#include "lt_host.h"

using namespace lt;
using namespace lt_impl;

extern "C" {

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
                  int32_t *out_node_ids, double *out_scores, double *out_line3d10) {
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
    double *o = out_line3d10 + 10 * e;  // the supporting Line3d as the reference's LineTrack::line3d_list holds it
    for (int q = 0; q < 3; ++q) { o[q] = c.s[q]; o[3 + q] = c.e[q]; }
    o[6] = c.depth[0]; o[7] = c.depth[1]; o[8] = c.unc; o[9] = c.score3;
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

// ---- bulk form (include/limap_amd.h: the layout of limap_amd.dist.pack_image_results) ----
int lt_export_images_size(lt_ctx *ctx, int n, const int32_t *img_ids, int64_t *n_ints, int64_t *n_dbls) {
  int rc = lt_flush(ctx);
  if (rc) return rc;
  long long ni = 1, nd = 0;
  for (int k = 0; k < n; ++k) {
    auto it = ctx->id2idx.find(img_ids[k]);
    if (it == ctx->id2idx.end()) return fail(ctx, LT_ERR_ARGUMENT, "unknown image id " + std::to_string(img_ids[k]));
    const int idx = it->second;
    if (!ctx->triangulated[idx]) return fail(ctx, LT_ERR_STATE, "image was not triangulated on this context");
    const long long m = ctx->seg_off[idx + 1] - ctx->seg_off[idx];
    long long ne = 0;
    for (long long g = ctx->seg_off[idx]; g < ctx->seg_off[idx + 1]; ++g) ne += (long long)ctx->valid_edges[g].size() / 2;
    ni += 4 + (long long)ctx->neighbors[idx].size() + 4 * m + 2 * ne;
    nd += 11 * m;
  }
  *n_ints = ni;
  *n_dbls = nd;
  return LT_OK;
}

int lt_export_images_packed(lt_ctx *ctx, int n, const int32_t *img_ids, int32_t *ints, double *dbls) {
  int rc = lt_flush(ctx);
  if (rc) return rc;
  int32_t *ip = ints;
  double *dp = dbls;
  *ip++ = n;
  for (int k = 0; k < n; ++k) {
    auto it = ctx->id2idx.find(img_ids[k]);
    if (it == ctx->id2idx.end()) return fail(ctx, LT_ERR_ARGUMENT, "unknown image id " + std::to_string(img_ids[k]));
    const int idx = it->second;
    if (!ctx->triangulated[idx]) return fail(ctx, LT_ERR_STATE, "image was not triangulated on this context");
    const auto &nb = ctx->neighbors[idx];
    const long long g0 = ctx->seg_off[idx], m = ctx->seg_off[idx + 1] - g0;
    int32_t *head = ip;
    ip += 4;
    for (size_t j = 0; j < nb.size(); ++j) *ip++ = ctx->img_ids[nb[j]];
    int32_t *src = ip, *nt = src + 2 * m, *cnt = nt + m, *edges = cnt + m;
    double *line = dp, *score = dp + 10 * m;
    long long ne = 0;
    for (long long l = 0; l < m; ++l) {
      const long long g = g0 + l;
      const Cand &c = ctx->best_c[g];
      double *o = line + 10 * l;
      for (int q = 0; q < 3; ++q) { o[q] = c.s[q]; o[3 + q] = c.e[q]; }
      o[6] = c.depth[0]; o[7] = c.depth[1]; o[8] = c.unc; o[9] = c.score3;
      score[l] = ctx->best_score[g];
      src[2 * l] = ctx->best_src2[2 * g];
      src[2 * l + 1] = ctx->best_src2[2 * g + 1];
      nt[l] = ctx->n_tris[g];
      const auto v = ctx->valid_edges[g];
      if (!v.empty()) std::memcpy(edges + 2 * ne, v.data(), 4 * v.size());
      cnt[l] = (int32_t)(v.size() / 2);
      ne += (long long)v.size() / 2;
    }
    head[0] = img_ids[k]; head[1] = (int32_t)nb.size(); head[2] = (int32_t)m; head[3] = (int32_t)ne;
    ip = edges + 2 * ne;
    dp += 11 * m;
  }
  return LT_OK;
}

int lt_import_images_packed(lt_ctx *ctx, const int32_t *ints, int64_t n_ints, const double *dbls, int64_t n_dbls) {
  if (!ctx->inited) return fail(ctx, LT_ERR_STATE, "import before Init");
  if (n_ints < 1 || ints[0] < 0) return fail(ctx, LT_ERR_ARGUMENT, "lt_import_images_packed: malformed blob");
  const int n = ints[0];
  // first pass: everything in the blob is checked before the context changes
  {
    long long ip = 1, dp = 0;
    for (int k = 0; k < n; ++k) {
      if (ip + 4 > n_ints) return fail(ctx, LT_ERR_ARGUMENT, "lt_import_images_packed: blob ends inside an image header");
      const long long n_nb = ints[ip + 1], m = ints[ip + 2], ne = ints[ip + 3];
      auto it = ctx->id2idx.find(ints[ip]);
      if (it == ctx->id2idx.end()) return fail(ctx, LT_ERR_ARGUMENT, "unknown image id " + std::to_string(ints[ip]));
      if (n_nb < 0 || n_nb > 255 || ne < 0 || m != ctx->seg_off[it->second + 1] - ctx->seg_off[it->second])
        return fail(ctx, LT_ERR_ARGUMENT, "lt_import_images_packed: image " + std::to_string(ints[ip]) +
                                              " has other line / neighbour counts than the blob says");
      const long long body = n_nb + 4 * m + 2 * ne;
      if (ip + 4 + body > n_ints || dp + 11 * m > n_dbls)
        return fail(ctx, LT_ERR_ARGUMENT, "lt_import_images_packed: blob shorter than its headers say");
      const int32_t *nbp = ints + ip + 4, *cnt = nbp + n_nb + 3 * m;
      for (long long j = 0; j < n_nb; ++j)
        if (ctx->id2idx.find(nbp[j]) == ctx->id2idx.end())
          return fail(ctx, LT_ERR_ARGUMENT, "unknown neighbour image id " + std::to_string(nbp[j]));
      long long sum = 0;
      for (long long l = 0; l < m; ++l) {
        if (cnt[l] < 0) return fail(ctx, LT_ERR_ARGUMENT, "lt_import_images_packed: negative edge count");
        sum += cnt[l];
      }
      if (sum != ne) return fail(ctx, LT_ERR_ARGUMENT, "lt_import_images_packed: edge counts do not add up");
      ip += 4 + body;
      dp += 11 * m;
    }
  }
  long long ip = 1, dp = 0;
  for (int k = 0; k < n; ++k) {
    const int idx = ctx->id2idx.find(ints[ip])->second;
    const long long n_nb = ints[ip + 1], m = ints[ip + 2], ne = ints[ip + 3];
    const int32_t *nbp = ints + ip + 4, *src = nbp + n_nb, *nt = src + 2 * m, *cnt = nt + m, *edges = cnt + m;
    const double *line = dbls + dp, *score = line + 10 * m;
    std::vector<int> nb((size_t)n_nb);
    for (long long j = 0; j < n_nb; ++j) nb[(size_t)j] = ctx->id2idx.find(nbp[j])->second;
    ctx->neighbors[idx] = nb;
    ctx->triangulated[idx] = 1;
    const long long g0 = ctx->seg_off[idx];
    long long e = 0;
    for (long long l = 0; l < m; ++l) {
      const long long g = g0 + l;
      Cand &c = ctx->best_c[g];
      c = Cand{};
      const double *o = line + 10 * l;
      for (int q = 0; q < 3; ++q) { c.s[q] = o[q]; c.e[q] = o[3 + q]; }
      c.depth[0] = o[6]; c.depth[1] = o[7]; c.unc = o[8]; c.score3 = o[9];
      ctx->best_score[g] = score[l];
      ctx->best_src2[2 * g] = src[2 * l];
      ctx->best_src2[2 * g + 1] = src[2 * l + 1];
      ctx->n_tris[g] = nt[l];
      ctx->has_best[g] = nt[l] > 0 ? 1 : 0;
      ctx->valid_edges.set(g, edges + 2 * e, edges + 2 * (e + cnt[l]));
      e += cnt[l];
    }
    ctx->best_c_set[(size_t)idx] = 1;
    ip += 4 + n_nb + 4 * m + 2 * ne;
    dp += 11 * m;
  }
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
    if (ctx->ran && ctx->job_mode == 1 && ctx->n_blk > 0 && ctx->max_rows > 0 && (ctx->d_surv_count.p || ctx->d_blk_surv.p)) {
      // survivors per slot (row-slot form) or per block (line-slot form)
      const bool ln = ctx->rows_ln && ctx->d_blk_surv.p;
      const size_t n = ln ? (size_t)ctx->n_blk : (size_t)ctx->n_blk * (size_t)gen_slots(ctx->max_rows);
      std::vector<unsigned> sc(n);
      HIPCHK(ctx, hipSetDevice(ctx->device));
      HIPCHK(ctx, hipMemcpy(sc.data(), ln ? ctx->d_blk_surv.p : ctx->d_surv_count.p, 4 * n, hipMemcpyDeviceToHost));
      long long tot = 0;
      for (unsigned v : sc) tot += v;
      ctx->stat_survivors = tot;
    }
  }
  ctx->timers[16] = (double)ctx->stat_survivors;
  ctx->timers[20] = (ctx->ran && ctx->job_mode == 1 && ctx->rows_ln) ? 1.0 : 0.0;  // stage A ran in the line-slot form
  ctx->timers[19] = ctx->score_fused ? 1.0 : 0.0;  // the split scoring form's pair store overflowed once: fused from then on
  ctx->timers[21] = ctx->score_two_kernels ? 1.0 : 0.0;  // k_score_q raised device flag 8 once: sweep + k_dense8 from then on
  std::memcpy(out, ctx->timers, sizeof(ctx->timers));
  return LT_OK;
}

int lt_get_timer_sums(lt_ctx *ctx, double out[24], int64_t *n_runs, int reset) {
  LT_FINISH(ctx);
  std::memcpy(out, ctx->timer_sums, sizeof(ctx->timer_sums));
  // the stage timers are sampled in pipelined runs (lt_ctx.h): their sums are scaled to the number of runs
  if (ctx->timer_stage_runs > 0 && ctx->timer_stage_runs < ctx->timer_runs) {
    const double f = (double)ctx->timer_runs / (double)ctx->timer_stage_runs;
    for (int k : {3, 4, 5, 6, 13, 14, 15}) out[k] *= f;
  }
  if (n_runs) *n_runs = ctx->timer_runs;
  if (reset) {
    std::memset(ctx->timer_sums, 0, sizeof(ctx->timer_sums));
    ctx->timer_runs = 0;
    ctx->timer_stage_runs = 0;
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

int64_t lt_fn_compressed_block_words(int64_t n) { return n < 0 ? -1 : (int64_t)lt::cb_words(n); }
int lt_fn_pack_match_rows_compressed(const int32_t *rows, int64_t n, uint32_t *out, uint32_t stats[4], int level) {
  if (n < 0 || (n > 0 && (!rows || !out)) || !stats) return LT_ERR_ARGUMENT;
  if ((reinterpret_cast<uintptr_t>(out) & 7u) != 0) return LT_ERR_ARGUMENT;  // the bit words are 64-bit
  const lt::RowStats rs = lt::pack_rows_cb(rows, n, out, level);
  stats[0] = rs.mx_line;
  stats[1] = rs.mx_ng;
  stats[2] = (uint32_t)rs.unsorted;
  stats[3] = (uint32_t)rs.irregular;
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
