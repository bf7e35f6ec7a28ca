// lt_tracks.cpp -- the steps that follow ComputeLineTracks inside limap.runners.line_triangulation
// (runners/line_triangulation.py:171-200): limap.merging.filter_tracks_by_reprojection, remerge,
// filter_tracks_by_sensitivity, filter_tracks_by_overlap -- SURVEY.md 8(f) rank 2.
//
// Reference: merging/merging_utils.cc:27-155 (CheckReprojection, FilterSupportingLines,
// CheckSensitivity, FilterTracksBySensitivity, FilterTracksByOverlap), merging/merging.cc:513-644
// (RemergeLineTracks), merging/merging.py:24-42 (the fixed-point loop).
//
// The per-support tests are a few 10^3..10^6 independent projections: they run on the host, shared out
// over the persistent thread team of lt_pool.h, through the same lt_geom.h code as the kernels.  The all-pairs check_connection of the
// remerge (O(T^2), the next hotspot on big scenes) runs on the GPU (k_track_connect); the union-find
// over its edges and the aggregation stay on the host like the rest of the tail.

#include "lt_ctx.h"
#include "lt_pool.h"
#include "lt_tail.h"

#include <atomic>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <set>

using namespace lt;

namespace {

struct Member {
  int img_id, line_id, node_id;
  double score;
  double l2d[4];
  Cand l3d;
};
struct TrackFull {
  double line[7];  // start, end, uncertainty
  bool active = true;
  // `line` is aggregate(m, agg_k) bit for bit (the aggregator is a pure function of the members in their order and of
  // num_outliers; lt_tail.h), or -1 when that is not known: lets the filters and the remerge skip re-aggregating tracks
  // whose members did not change -- the reference recomputes them and gets the same bits
  int agg_k = -1;
  std::vector<Member> m;
};

d2 project(const Cam &c, const double *p) { return cam_project(c, mk3(p[0], p[1], p[2])); }

void reaggregate(TrackFull &t, int num_outliers) {
  static thread_local AggScratch scratch;
  static thread_local std::vector<double> scores;
  scores.resize(t.m.size());
  for (size_t k = 0; k < t.m.size(); ++k) scores[k] = t.m[k].score;
  aggregate_impl([&](int i) -> const Cand & { return t.m[(size_t)i].l3d; }, scores.data(), (int)t.m.size(), num_outliers,
                 t.line, scratch);
  t.agg_k = num_outliers;
}

int distinct(std::vector<int> &v) {  // number of different values (the size of the reference's std::set)
  std::sort(v.begin(), v.end());
  return (int)(std::unique(v.begin(), v.end()) - v.begin());
}

double multiplier(double score_th) { return 1.0 / std::sqrt(-std::log(score_th) * 2.0); }

// Tracks per piece of the per-track loops below, shared out over the persistent host team (lt_pool.h) -- not OpenMP
// regions: a team that has gone to sleep costs 0.25-0.55 ms to start on the GPU box's host, five times per chain.
constexpr long long kTrackGrain = 8;

}  // namespace

struct lt_trackset {
  std::vector<TrackFull> tracks;
};

extern "C" {

lt_trackset *lt_ts_from_ctx(lt_ctx *ctx) {
  lt_host::SpinPool::get(lt_host::row_workers()).wake();  // the filters follow: the team leaves its sleep meanwhile
  lt_trackset *ts = new lt_trackset();
  const lt_host::TrackStore &src = ctx->tracks;
  ts->tracks.resize(src.size());
  lt_host::pool_for((long long)src.size(), 32, [&](long long t0_, long long t1_) {
  for (size_t t = (size_t)t0_; t < (size_t)t1_; ++t) {
    TrackFull &dst = ts->tracks[t];
    std::memcpy(dst.line, src.line7.data() + 7 * t, sizeof(dst.line));
    dst.agg_k = ctx->cfg.num_outliers_aggregator;  // lt_compute_tracks aggregated exactly these members with it
    const size_t a = (size_t)src.off[t], n = (size_t)src.off[t + 1] - a;
    dst.m.resize(n);
    for (size_t k = 0; k < n; ++k) {
      Member &mm = dst.m[k];
      mm.img_id = src.img_ids[a + k]; mm.line_id = src.line_ids[a + k]; mm.node_id = src.node_ids[a + k];
      mm.score = src.scores[a + k];
      long long g = src.gnodes[a + k];
      for (int c = 0; c < 4; ++c) mm.l2d[c] = ctx->h_segs_ptr[4 * g + c] + ctx->h_segs_add;  // (v + 0.5 as at Init before round 6)
      mm.l3d = ctx->best_c[g];
    }
  }
  });
  return ts;
}

lt_trackset *lt_ts_create(int64_t T, const double *line7, const uint8_t *active, const int64_t *off,
                          const int32_t *img, const int32_t *lid, const int32_t *nid, const double *score,
                          const double *line2d4, const double *line3d10) {
  lt_host::SpinPool::get(lt_host::row_workers()).wake();
  lt_trackset *ts = new lt_trackset();
  ts->tracks.resize((size_t)T);
  for (int64_t t = 0; t < T; ++t) {
    TrackFull &dst = ts->tracks[t];
    std::memcpy(dst.line, line7 + 7 * t, 56);
    dst.active = active ? active[t] != 0 : true;
    for (int64_t e = off[t]; e < off[t + 1]; ++e) {
      Member mm;
      mm.img_id = img[e]; mm.line_id = lid[e]; mm.node_id = nid ? nid[e] : 0; mm.score = score ? score[e] : 0.0;
      std::memcpy(mm.l2d, line2d4 + 4 * e, 32);
      const double *o = line3d10 + 10 * e;
      for (int k = 0; k < 3; ++k) { mm.l3d.s[k] = o[k]; mm.l3d.e[k] = o[3 + k]; }
      mm.l3d.depth[0] = o[6]; mm.l3d.depth[1] = o[7]; mm.l3d.unc = o[8]; mm.l3d.score3 = o[9];
      for (int k = 0; k < 4; ++k) mm.l3d.seg[k] = 0.0;
      dst.m.push_back(mm);
    }
  }
  return ts;
}

void lt_ts_destroy(lt_trackset *ts) { delete ts; }
int64_t lt_ts_num_tracks(lt_trackset *ts) { return (int64_t)ts->tracks.size(); }
int64_t lt_ts_num_members(lt_trackset *ts) {
  int64_t n = 0;
  for (auto &t : ts->tracks) n += (int64_t)t.m.size();
  return n;
}

int lt_ts_get(lt_trackset *ts, double *line7, uint8_t *active, int64_t *off, int32_t *img, int32_t *lid,
              int32_t *nid, double *score, double *line2d4, double *line3d10) {
  int64_t e = 0, ti = 0;
  off[0] = 0;
  for (auto &t : ts->tracks) {
    std::memcpy(line7 + 7 * ti, t.line, 56);
    active[ti] = t.active ? 1 : 0;
    for (const Member &mm : t.m) {
      img[e] = mm.img_id; lid[e] = mm.line_id; nid[e] = mm.node_id; score[e] = mm.score;
      std::memcpy(line2d4 + 4 * e, mm.l2d, 32);
      double *o = line3d10 + 10 * e;
      for (int k = 0; k < 3; ++k) { o[k] = mm.l3d.s[k]; o[3 + k] = mm.l3d.e[k]; }
      o[6] = mm.l3d.depth[0]; o[7] = mm.l3d.depth[1]; o[8] = mm.l3d.unc; o[9] = mm.l3d.score3;
      ++e;
    }
    off[++ti] = e;
  }
  return LT_OK;
}

static int cam_of(lt_ctx *ctx, int img_id, const Cam **out) {
  auto it = ctx->id2idx.find(img_id);
  if (it == ctx->id2idx.end()) return fail(ctx, LT_ERR_ARGUMENT, "unknown image id " + std::to_string(img_id));
  *out = &ctx->h_cams[it->second];
  return LT_OK;
}

// FilterSupportingLines (merging_utils.cc:51-83) with CheckReprojection (:27-49)
int lt_ts_filter_by_reprojection(lt_ctx *ctx, lt_trackset *ts, double th_angular2d, double th_perp2d,
                                 int num_outliers) {
  if (!ctx->inited) return fail(ctx, LT_ERR_STATE, "filter before Init");
  const long long nT = (long long)ts->tracks.size();
  for (auto &t : ts->tracks)
    for (auto &mm : t.m) {
      const Cam *c;
      int rc = cam_of(ctx, mm.img_id, &c);
      if (rc) return rc;
    }
  std::vector<TrackFull> out((size_t)nT);
  std::vector<char> keep((size_t)nT, 0);
  lt_host::pool_for(nT, kTrackGrain, [&](long long t0_, long long t1_) {
  for (long long ti = t0_; ti < t1_; ++ti) {
    TrackFull &t = ts->tracks[ti];
    TrackFull nt;
    size_t n_kept = 0;
    static thread_local std::vector<char> ok;
    ok.assign(t.m.size(), 0);
    for (size_t k = 0; k < t.m.size(); ++k) {
      const Member &mm = t.m[k];
      const Cam &c = ctx->h_cams[ctx->id2idx.at(mm.img_id)];
      L2 det{mk2(mm.l2d[0], mm.l2d[1]), mk2(mm.l2d[2], mm.l2d[3])};
      L2 proj{project(c, t.line), project(c, t.line + 3)};
      double angle = angle_between(det, proj);
      if (angle > th_angular2d) continue;
      double ds, de;
      perp_oneway(det, proj, &ds, &de);  // dist_endpoints_perpendicular_oneway (line_dists.h:113-120)
      if (dmax(ds, de) > th_perp2d) continue;
      ok[k] = 1;
      ++n_kept;
    }
    if (n_kept == 0) continue;
    if (n_kept == t.m.size()) {
      // every support stays: the new track has the old one's members; its line is their aggregate -- which the old
      // line already is when it was aggregated with the same num_outliers
      nt.m = std::move(t.m);
      if (t.agg_k == num_outliers) {
        std::memcpy(nt.line, t.line, sizeof(nt.line));
        nt.agg_k = t.agg_k;
      } else {
        reaggregate(nt, num_outliers);
      }
    } else {
      nt.m.reserve(n_kept);
      for (size_t k = 0; k < t.m.size(); ++k)
        if (ok[k]) nt.m.push_back(t.m[k]);
      reaggregate(nt, num_outliers);
    }
    nt.active = true;  // a fresh LineTrack in the reference
    out[ti] = std::move(nt);
    keep[ti] = 1;
  }
  });
  std::vector<TrackFull> packed;
  for (long long ti = 0; ti < nT; ++ti)
    if (keep[ti]) packed.push_back(std::move(out[ti]));
  ts->tracks.swap(packed);
  return LT_OK;
}

// FilterTracksBySensitivity (merging_utils.cc:85-128); Line3d::sensitivity (linebase.cc:100-107)
int lt_ts_filter_by_sensitivity(lt_ctx *ctx, lt_trackset *ts, double th_angular3d, int min_supports) {
  if (!ctx->inited) return fail(ctx, LT_ERR_STATE, "filter before Init");
  const long long nT = (long long)ts->tracks.size();
  std::vector<char> keep((size_t)nT, 0);
  std::atomic<int> bad{0};
  lt_host::pool_for(nT, kTrackGrain, [&](long long t0_, long long t1_) {
  for (long long ti = t0_; ti < t1_; ++ti) {
    const TrackFull &t = ts->tracks[ti];
    d3 s = mk3(t.line[0], t.line[1], t.line[2]), e = mk3(t.line[3], t.line[4], t.line[5]);
    d3 dir3 = unit(sub(e, s));
    static thread_local std::vector<int> imgs;  // (std::set<int> in the reference: only its size is used)
    imgs.clear();
    for (const Member &mm : t.m) {
      auto it = ctx->id2idx.find(mm.img_id);
      if (it == ctx->id2idx.end()) {
        bad.store(1, std::memory_order_relaxed);
        continue;
      }
      const Cam &c = ctx->h_cams[it->second];
      d2 ps = cam_project(c, s), pe = cam_project(c, e);
      d3 ray = cam_ray(c, d2{0.5 * (ps.x + pe.x), 0.5 * (ps.y + pe.y)});
      double sens = 90 - acos(fabs(dot(dir3, ray))) * 180.0 / kPi;
      if (!(sens > th_angular3d)) imgs.push_back(mm.img_id);
    }
    keep[ti] = distinct(imgs) >= min_supports;
  }
  });
  if (bad.load()) return fail(ctx, LT_ERR_ARGUMENT, "track references an unknown image id");
  std::vector<TrackFull> packed;
  for (long long ti = 0; ti < nT; ++ti)
    if (keep[ti]) packed.push_back(std::move(ts->tracks[ti]));
  ts->tracks.swap(packed);
  return LT_OK;
}

// FilterTracksByOverlap (merging_utils.cc:130-155): overlap of the projection onto the detection
int lt_ts_filter_by_overlap(lt_ctx *ctx, lt_trackset *ts, double th_overlap, int min_supports) {
  if (!ctx->inited) return fail(ctx, LT_ERR_STATE, "filter before Init");
  const long long nT = (long long)ts->tracks.size();
  std::vector<char> keep((size_t)nT, 0);
  std::atomic<int> bad{0};
  lt_host::pool_for(nT, kTrackGrain, [&](long long t0_, long long t1_) {
  for (long long ti = t0_; ti < t1_; ++ti) {
    const TrackFull &t = ts->tracks[ti];
    static thread_local std::vector<int> imgs;
    imgs.clear();
    for (const Member &mm : t.m) {
      auto it = ctx->id2idx.find(mm.img_id);
      if (it == ctx->id2idx.end()) {
        bad.store(1, std::memory_order_relaxed);
        continue;
      }
      const Cam &c = ctx->h_cams[it->second];
      L2 proj{project(c, t.line), project(c, t.line + 3)};
      L2 det{mk2(mm.l2d[0], mm.l2d[1]), mk2(mm.l2d[2], mm.l2d[3])};
      if (overlap_oneway(proj, det) >= th_overlap) imgs.push_back(mm.img_id);
    }
    keep[ti] = distinct(imgs) >= min_supports;
  }
  });
  if (bad.load()) return fail(ctx, LT_ERR_ARGUMENT, "track references an unknown image id");
  std::vector<TrackFull> packed;
  for (long long ti = 0; ti < nT; ++ti)
    if (keep[ti]) packed.push_back(std::move(ts->tracks[ti]));
  ts->tracks.swap(packed);
  return LT_OK;
}

// One pass of RemergeLineTracks (merging/merging.cc:513-644).  linker = l3_* fields of `linker_cfg`.
int lt_ts_remerge_once(lt_ctx *ctx, lt_trackset *ts, const lt_config *linker_cfg, int num_outliers) {
  HIPCHK(ctx, hipSetDevice(ctx->device));
  const int T = (int)ts->tracks.size();
  if (T == 0) return LT_OK;
  LinkCfg3 l3;
  l3.score_th = linker_cfg->l3_score_th; l3.th_angle = linker_cfg->l3_th_angle; l3.th_overlap = linker_cfg->l3_th_overlap;
  l3.th_smartoverlap = linker_cfg->l3_th_smartoverlap; l3.th_smartangle = linker_cfg->l3_th_smartangle;
  l3.th_perp = linker_cfg->l3_th_perp; l3.th_innerseg = linker_cfg->l3_th_innerseg;
  l3.th_scaleinv = linker_cfg->l3_th_scaleinv; l3.mult = multiplier(linker_cfg->l3_score_th);
  l3.use_smartangle = linker_cfg->l3_use_smartangle;
  // set_to_spatial_merging (line_linker.h:123-129)
  l3.use_angle = 1; l3.use_overlap = 1; l3.use_perp = 0; l3.use_innerseg = 1; l3.use_scaleinv = 0;
  double th = l3.th_angle * (1.0 + 1e-6) + 1e-6;
  double cos_guard = (th < 90.0) ? std::cos(th * kPi / 180.0) : -1.0;

  // host image of the device input: [7 T doubles: the track lines | T bytes: active flags], one copy
  std::vector<double> &inbuf = ctx->h_rm_in;
  const size_t in_bytes = 56 * (size_t)T + (size_t)T;
  inbuf.resize((in_bytes + 7) / 8);
  unsigned char *active = reinterpret_cast<unsigned char *>(inbuf.data() + 7 * (size_t)T);
  int n_active = 0;
  for (int t = 0; t < T; ++t) {
    std::memcpy(&inbuf[7 * (size_t)t], ts->tracks[t].line, 56);
    active[t] = ts->tracks[t].active ? 1 : 0;
    n_active += active[t];
  }
  // the device buffers and the edge list live in the context: a remerge to its fixed point calls this several times
  DevBuf &d_in = ctx->d_rm_line, &d_edges = ctx->d_rm_edges;
  hipStream_t st = ctx->stream;
  std::vector<unsigned long long> &edges = ctx->h_rm_edges;
  edges.clear();
  // d_edges = [edge count | edges ...]: count and the first kFirst edges come back in ONE copy behind the kernel
  constexpr unsigned long long kFirst = 4095;
  unsigned long long capacity = std::max<unsigned long long>(1ull << 16, 32ull * (unsigned long long)T);
  int rc = LT_OK;
  std::vector<unsigned long long> &back = ctx->h_rm_back;
  back.resize((size_t)kFirst + 1);
  for (int attempt = 0; attempt < 8; ++attempt) {
    if (!d_in.ensure(inbuf.size() * 8) || !d_edges.ensure((capacity + 1) * 8)) {
      rc = fail(ctx, LT_ERR_HIP, "hipMalloc failed in remerge");
      break;
    }
    if (hipMemcpyAsync(d_in.p, inbuf.data(), inbuf.size() * 8, hipMemcpyHostToDevice, st) != hipSuccess ||
        hipMemsetAsync(d_edges.p, 0, 8, st) != hipSuccess) {
      rc = fail(ctx, LT_ERR_HIP, "HIP copy failed in remerge");
      break;
    }
    launch_track_connect(st, T, d_in.as<double>(), reinterpret_cast<const unsigned char *>(d_in.as<double>() + 7 * (size_t)T),
                         n_active == T ? 1 : 0, l3, cos_guard, d_edges.as<unsigned long long>() + 1, capacity,
                         d_edges.as<unsigned long long>());
    if (hipMemcpyAsync(back.data(), d_edges.p, (kFirst + 1) * 8, hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess) {
      rc = fail(ctx, LT_ERR_HIP, "HIP failure in k_track_connect");
      break;
    }
    const unsigned long long n = back[0];
    if (n > capacity) {  // rare: more edges than reserved, run again with room for all of them
      capacity = n + 1024;
      continue;
    }
    edges.assign(back.begin() + 1, back.begin() + 1 + (size_t)std::min(n, kFirst));
    if (n > kFirst) {
      edges.resize((size_t)n);
      if (hipMemcpy(edges.data() + kFirst, d_edges.as<unsigned long long>() + 1 + kFirst, (size_t)(n - kFirst) * 8,
                    hipMemcpyDeviceToHost) != hipSuccess)
        rc = fail(ctx, LT_ERR_HIP, "HIP copy failed in remerge");
    }
    break;
  }
  if (rc) return rc;
  // std::set<pair<size_t,size_t>> order + dedupe
  std::sort(edges.begin(), edges.end());
  edges.erase(std::unique(edges.begin(), edges.end()), edges.end());
  std::vector<int> parent((size_t)T, -1);
  std::vector<size_t> gsize((size_t)T, 1);
  for (unsigned long long e : edges) {
    int r1 = uf_root((int)(e >> 32), parent), r2 = uf_root((int)(e & 0xFFFFFFFFull), parent);
    if (r1 == r2) continue;
    if (gsize[r1] < gsize[r2]) {
      parent[r1] = r2;
      gsize[r2] += gsize[r1];
      gsize[r1] = 0;
    } else {
      parent[r2] = r1;
      gsize[r1] += gsize[r2];
      gsize[r2] = 0;
    }
  }
  std::vector<long> labels((size_t)T, -1);
  long n_groups = 0;
  for (int t = 0; t < T; ++t)
    if (parent[t] == -1) labels[t] = n_groups++;
  for (int t = 0; t < T; ++t)
    if (labels[t] == -1) labels[t] = labels[uf_root(t, parent)];
  std::vector<TrackFull> out((size_t)n_groups);
  std::vector<int> counter((size_t)n_groups, 0);
  for (int t = 0; t < T; ++t) counter[labels[t]]++;
  for (int t = 0; t < T; ++t) {
    TrackFull &g = out[labels[t]];
    if (counter[labels[t]] == 1) {
      // a group of one: the new track has this track's members, and its line is their aggregate -- the line it has
      g.m = std::move(ts->tracks[t].m);
      g.agg_k = ts->tracks[t].agg_k;
      std::memcpy(g.line, ts->tracks[t].line, sizeof(g.line));
    } else {
      g.m.insert(g.m.end(), ts->tracks[t].m.begin(), ts->tracks[t].m.end());
    }
  }
  lt_host::pool_for((long long)n_groups, kTrackGrain, [&](long long g0_, long long g1_) {
    for (long long gi = g0_; gi < g1_; ++gi) {
      // the reference re-aggregates every group (merging.cc:629-640); for a group of one whose line already is the
      // aggregate of its members with this num_outliers that gives the same bits
      if (counter[gi] != 1 || out[gi].agg_k != num_outliers) reaggregate(out[gi], num_outliers);
      out[gi].active = counter[gi] != 1;
    }
  });
  ts->tracks.swap(out);
  return LT_OK;
}

}  // extern "C"
