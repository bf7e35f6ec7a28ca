// oracle/ref_driver.cpp -- C entry points over the REFERENCE'S OWN classes, compiled together with the
// unmodified sources under /root/reference/src/limap into oracle/_ref/liblimap_ref.so (recipe: oracle/Makefile,
// target `ref`).  TEST INFRASTRUCTURE: the checker the oracle (oracle/lt_oracle.cpp) is pinned against
// (tests/test_oracle_vs_ref.py); nothing under limap_amd/ may load it.
//
// The functions are the ora_* functions of lt_oracle.h with the prefix ref_ and the same argument meaning, so
// that oracle/ref.py can drive this library through the OracleTriangulator wrapper.  What runs underneath:
//   limap::triangulation::GlobalLineTriangulator (Init / TriangulateImage* / ComputeLineTracks) with its config
//   parsed from a py::dict by the reference's own constructors, limap::ImageCollection / CameraView / Line2d /
//   Line3d, limap::LineLinker2d/3d, limap::merging::{Aggregator, ComputeLineTrackLabels*, FilterSupportingLines,
//   FilterTracksBySensitivity, FilterTracksByOverlap, RemergeLineTracks}, limap::triangulation free functions,
//   limap::solvers::triangulation::triangulate_line_with_one_point.
// What does NOT come from the reference: Eigen, COLMAP, PoseLib (oracle/ref_shim/, see the headers there).
// pybind11 objects are created here, so the library must be called with the GIL held (ctypes.PyDLL).
#include "limap/base/graph.h"
#include "limap/base/image_collection.h"
#include "limap/base/line_linker.h"
#include "limap/base/linebase.h"
#include "limap/base/linetrack.h"
#include "limap/merging/aggregator.h"
#include "limap/merging/merging.h"
#include "limap/merging/merging_utils.h"
#include "limap/structures/pl_bipartite.h"
#include "limap/triangulation/functions.h"
#include "limap/triangulation/global_line_triangulator.h"
#include "limap/vplib/vpbase.h"

#include <omp.h>

#include <cstring>
#include <memory>

#include "lt_oracle.h"

using namespace limap;
namespace tri = limap::triangulation;

namespace {

struct RefTri : tri::GlobalLineTriangulator {  // opens the protected state for read-out
  using tri::GlobalLineTriangulator::GlobalLineTriangulator;
  using tri::GlobalLineTriangulator::run_clustering;
  using tri::GlobalLineTriangulator::tris_best_;
  using tri::GlobalLineTriangulator::valid_edges_;
  using tri::GlobalLineTriangulator::valid_tris_;
  using tri::BaseLineTriangulator::all_lines_2d_;
  using tri::BaseLineTriangulator::neighbors_;
  using tri::BaseLineTriangulator::tris_;
};

py::dict linker2d_dict(const ora_config &c) {
  py::dict d;
  d["score_th"] = c.l2_score_th; d["th_angle"] = c.l2_th_angle; d["th_overlap"] = c.l2_th_overlap;
  d["th_smartoverlap"] = c.l2_th_smartoverlap; d["th_smartangle"] = c.l2_th_smartangle; d["th_perp"] = c.l2_th_perp;
  d["th_innerseg"] = c.l2_th_innerseg;
  d["use_angle"] = c.l2_use_angle != 0; d["use_overlap"] = c.l2_use_overlap != 0;
  d["use_smartangle"] = c.l2_use_smartangle != 0; d["use_perp"] = c.l2_use_perp != 0;
  d["use_innerseg"] = c.l2_use_innerseg != 0;
  return d;
}
py::dict linker3d_dict(const ora_config &c) {
  py::dict d;
  d["score_th"] = c.l3_score_th; d["th_angle"] = c.l3_th_angle; d["th_overlap"] = c.l3_th_overlap;
  d["th_smartoverlap"] = c.l3_th_smartoverlap; d["th_smartangle"] = c.l3_th_smartangle; d["th_perp"] = c.l3_th_perp;
  d["th_innerseg"] = c.l3_th_innerseg; d["th_scaleinv"] = c.l3_th_scaleinv;
  d["use_angle"] = c.l3_use_angle != 0; d["use_overlap"] = c.l3_use_overlap != 0;
  d["use_smartangle"] = c.l3_use_smartangle != 0; d["use_perp"] = c.l3_use_perp != 0;
  d["use_innerseg"] = c.l3_use_innerseg != 0; d["use_scaleinv"] = c.l3_use_scaleinv != 0;
  return d;
}
// cfg["triangulation"] as line_triangulation.py hands it to GlobalLineTriangulator(dict).  debug_mode is always on:
// it only keeps tris_ / valid_tris_ alive for the read-out (global_line_triangulator.cc:156-159).
py::dict config_dict(const ora_config &c) {
  py::dict d;
  d["debug_mode"] = true;
  d["add_halfpix"] = c.add_halfpix != 0;
  d["use_vp"] = c.use_vp != 0;
  d["use_endpoints_triangulation"] = c.use_endpoints_triangulation != 0;
  d["disable_many_points_triangulation"] = c.disable_many_points_triangulation != 0;
  d["disable_one_point_triangulation"] = c.disable_one_point_triangulation != 0;
  d["disable_algebraic_triangulation"] = c.disable_algebraic_triangulation != 0;
  d["disable_vp_triangulation"] = c.disable_vp_triangulation != 0;
  d["min_length_2d"] = c.min_length_2d;
  d["line_tri_angle_threshold"] = c.line_tri_angle_threshold;
  d["IoU_threshold"] = c.IoU_threshold;
  d["sensitivity_threshold"] = c.sensitivity_threshold;
  d["var2d"] = c.var2d;
  d["fullscore_th"] = c.fullscore_th;
  d["max_valid_conns"] = c.max_valid_conns;
  d["min_num_outer_edges"] = c.min_num_outer_edges;
  const char *names[] = {"greedy", "exhaustive", "avg"};
  d["merging_strategy"] = std::string(c.merging_strategy >= 0 && c.merging_strategy <= 2 ? names[c.merging_strategy]
                                                                                         : "not-a-strategy");
  d["num_outliers_aggregator"] = c.num_outliers_aggregator;
  d["linker2d_config"] = linker2d_dict(c);
  d["linker3d_config"] = linker3d_dict(c);
  return d;
}

CameraView view_from_cam11(const double cam[11]) {
  Camera c(1 /* PINHOLE */, std::vector<double>{cam[0], cam[1], cam[2], cam[3]});
  CameraPose p(V4D(cam[4], cam[5], cam[6], cam[7]), V3D(cam[8], cam[9], cam[10]));
  return CameraView(c, p);
}
Line2d seg_to_line(const double s[4]) { return Line2d(V2D(s[0], s[1]), V2D(s[2], s[3])); }
Line3d line_from10(const double a[10]) {
  return Line3d(V3D(a[0], a[1], a[2]), V3D(a[3], a[4], a[5]), a[9], a[6], a[7], a[8]);
}
void line_to10(const Line3d &l, double a[10]) {
  for (int k = 0; k < 3; ++k) { a[k] = l.start[k]; a[3 + k] = l.end[k]; }
  a[6] = l.depths[0]; a[7] = l.depths[1]; a[8] = l.uncertainty; a[9] = l.score;
}

}  // namespace

struct ora_ctx {  // (the type name of lt_oracle.h; this is the reference-backed one)
  ora_config cfg;
  std::unique_ptr<RefTri> t;
  std::unique_ptr<ImageCollection> imagecols;
  std::map<int, std::vector<Line2d>> segs;
  std::vector<int> ids;  // ascending
  std::string err;
  int64_t n_conn = 0, graph_nodes = 0, graph_edges = 0;
  double t_tri = 0, t_tail = 0;
};
struct ora_trackset {
  std::vector<LineTrack> tracks;
};

#define REF_TRY(ctx, ...)                      \
  try {                                        \
    __VA_ARGS__;                               \
    return 0;                                  \
  } catch (const std::exception &e) {          \
    (ctx)->err = e.what();                     \
    return -1;                                 \
  }

extern "C" {

// the REFERENCE'S defaults (default-constructed config classes), not a transcription of them
void ref_config_default(ora_config *c) {
  tri::GlobalLineTriangulatorConfig g;
  std::memset(c, 0, sizeof(*c));
  c->debug_mode = g.debug_mode; c->add_halfpix = g.add_halfpix; c->use_vp = g.use_vp;
  c->use_endpoints_triangulation = g.use_endpoints_triangulation;
  c->disable_many_points_triangulation = g.disable_many_points_triangulation;
  c->disable_one_point_triangulation = g.disable_one_point_triangulation;
  c->disable_algebraic_triangulation = g.disable_algebraic_triangulation;
  c->disable_vp_triangulation = g.disable_vp_triangulation;
  c->min_length_2d = g.min_length_2d; c->line_tri_angle_threshold = g.line_tri_angle_threshold;
  c->IoU_threshold = g.IoU_threshold; c->sensitivity_threshold = g.sensitivity_threshold; c->var2d = g.var2d;
  c->fullscore_th = g.fullscore_th; c->max_valid_conns = g.max_valid_conns;
  c->min_num_outer_edges = g.min_num_outer_edges;
  c->merging_strategy = g.merging_strategy == "greedy" ? 0 : (g.merging_strategy == "exhaustive" ? 1 : 2);
  c->num_outliers_aggregator = g.num_outliers_aggregator;
  const LineLinker2dConfig &a = g.linker2d_config;
  c->l2_score_th = a.score_th; c->l2_th_angle = a.th_angle; c->l2_th_overlap = a.th_overlap;
  c->l2_th_smartoverlap = a.th_smartoverlap; c->l2_th_smartangle = a.th_smartangle; c->l2_th_perp = a.th_perp;
  c->l2_th_innerseg = a.th_innerseg; c->l2_use_angle = a.use_angle; c->l2_use_overlap = a.use_overlap;
  c->l2_use_smartangle = a.use_smartangle; c->l2_use_perp = a.use_perp; c->l2_use_innerseg = a.use_innerseg;
  const LineLinker3dConfig &b = g.linker3d_config;
  c->l3_score_th = b.score_th; c->l3_th_angle = b.th_angle; c->l3_th_overlap = b.th_overlap;
  c->l3_th_smartoverlap = b.th_smartoverlap; c->l3_th_smartangle = b.th_smartangle; c->l3_th_perp = b.th_perp;
  c->l3_th_innerseg = b.th_innerseg; c->l3_th_scaleinv = b.th_scaleinv; c->l3_use_angle = b.use_angle;
  c->l3_use_overlap = b.use_overlap; c->l3_use_smartangle = b.use_smartangle; c->l3_use_perp = b.use_perp != 0;
  c->l3_use_innerseg = b.use_innerseg; c->l3_use_scaleinv = b.use_scaleinv;
}

ora_ctx *ref_create(const ora_config *cfg, int /*faithful*/) {
  auto *ctx = new ora_ctx();
  ctx->cfg = *cfg;
  try {
    ctx->t = std::make_unique<RefTri>(tri::GlobalLineTriangulatorConfig(config_dict(*cfg)));
  } catch (const std::exception &e) {
    ctx->err = e.what();
  }
  return ctx;
}
void ref_destroy(ora_ctx *ctx) { delete ctx; }
const char *ref_last_error(ora_ctx *ctx) { return ctx->err.c_str(); }
void ref_set_num_threads(int n) { omp_set_num_threads(n); }
int ref_get_max_threads(void) { return omp_get_max_threads(); }

int ref_set_ranges(ora_ctx *ctx, const double lo[3], const double hi[3]) {
  REF_TRY(ctx, { ctx->t->SetRanges(std::make_pair(V3D(lo[0], lo[1], lo[2]), V3D(hi[0], hi[1], hi[2]))); })
}
int ref_unset_ranges(ora_ctx *ctx) { REF_TRY(ctx, { ctx->t->UnsetRanges(); }) }

int ref_init(ora_ctx *ctx, int n_img, const int32_t *img_ids, const double *kvec, const double *qvec,
             const double *tvec, const int64_t *seg_off, const double *segs) {
  REF_TRY(ctx, {
    std::map<int, Camera> cameras;
    std::map<int, CameraImage> images;
    ctx->segs.clear();
    for (int i = 0; i < n_img; ++i) {
      const int id = img_ids[i];
      // one undistorted PINHOLE camera per image (fx, fy, cx, cy); CameraPose normalises the quaternion
      Camera cam(1, std::vector<double>{kvec[4 * i], kvec[4 * i + 1], kvec[4 * i + 2], kvec[4 * i + 3]}, id);
      cameras.insert(std::make_pair(id, cam));
      CameraPose pose(V4D(qvec[4 * i], qvec[4 * i + 1], qvec[4 * i + 2], qvec[4 * i + 3]),
                      V3D(tvec[3 * i], tvec[3 * i + 1], tvec[3 * i + 2]));
      images.insert(std::make_pair(id, CameraImage(id, pose)));
      const int64_t m = seg_off[i + 1] - seg_off[i];
      Eigen::MatrixXd arr(m, 4);
      for (int64_t l = 0; l < m; ++l)
        for (int k = 0; k < 4; ++k) arr(l, k) = segs[4 * (seg_off[i] + l) + k];
      ctx->segs[id] = GetLine2dVectorFromArray(arr);
    }
    ctx->imagecols = std::make_unique<ImageCollection>(cameras, images);
    ctx->ids = ctx->imagecols->get_img_ids();
    ctx->t->Init(ctx->segs, *ctx->imagecols);
    ctx->n_conn = 0;
  })
}

int ref_init_vp(ora_ctx *ctx, int n_img, const int32_t *img_ids, const int64_t *label_off, const int32_t *labels,
                const int64_t *vp_off, const double *vps) {
  REF_TRY(ctx, {
    std::map<int, vplib::VPResult> res;
    for (int i = 0; i < n_img; ++i) {
      std::vector<int> lab(labels + label_off[i], labels + label_off[i + 1]);
      std::vector<V3D> v;
      for (int64_t k = vp_off[i]; k < vp_off[i + 1]; ++k) v.push_back(V3D(vps[3 * k], vps[3 * k + 1], vps[3 * k + 2]));
      res.insert(std::make_pair(int(img_ids[i]), vplib::VPResult(lab, v)));
    }
    ctx->t->InitVPResults(res);
  })
}

int ref_set_bipartites(ora_ctx *ctx, int n_img, const int32_t *img_ids, const int64_t *pt_off, const int32_t *pt_ids,
                       const double *pt_xy, const int32_t *pt_p3d, const int64_t *line_off, const int64_t *lp_off,
                       const int32_t *lp_ptids) {
  REF_TRY(ctx, {
    std::map<int, structures::PL_Bipartite2d> all;
    for (int i = 0; i < n_img; ++i) {
      const int id = img_ids[i];
      structures::PL_Bipartite2d b;
      const auto &lines = ctx->segs.at(id);
      b.init_lines(lines);  // ids 0..M-1
      for (int64_t p = pt_off[i]; p < pt_off[i + 1]; ++p)
        b.add_point(Point2d(V2D(pt_xy[2 * p], pt_xy[2 * p + 1]), pt_p3d[p]), pt_ids[p]);
      for (int64_t l = line_off[i]; l < line_off[i + 1]; ++l)
        for (int64_t e = lp_off[l]; e < lp_off[l + 1]; ++e) b.add_edge(lp_ptids[e], int(l - line_off[i]));
      all.insert(std::make_pair(id, b));
    }
    ctx->t->SetBipartites2d(all);
  })
}

int ref_set_sfm_points(ora_ctx *ctx, int64_t n, const int32_t *ids, const double *xyz) {
  REF_TRY(ctx, {
    std::map<int, V3D> pts;
    for (int64_t i = 0; i < n; ++i) pts[ids[i]] = V3D(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]);
    ctx->t->SetSfMPoints(pts);
  })
}

int ref_triangulate_image(ora_ctx *ctx, int img_id, int n_nb, const int32_t *nb_ids, const int64_t *m_off,
                          const int32_t *m_pairs) {
  REF_TRY(ctx, {
    std::map<int, Eigen::MatrixXi> matches;
    for (int k = 0; k < n_nb; ++k) {
      const int64_t n = m_off[k + 1] - m_off[k];
      Eigen::MatrixXi m(n, 2);
      for (int64_t r = 0; r < n; ++r) {
        m(r, 0) = m_pairs[2 * (m_off[k] + r)];
        m(r, 1) = m_pairs[2 * (m_off[k] + r) + 1];
      }
      matches[nb_ids[k]] = m;
      // statistic only, the oracle's definition: connections of the lines triangulateOneNode does not skip (:166-167)
      const auto &own = ctx->t->all_lines_2d_.at(img_id);
      for (int64_t r = 0; r < n; ++r)
        if (m(r, 0) >= 0 && size_t(m(r, 0)) < own.size() && own[size_t(m(r, 0))].length() > ctx->cfg.min_length_2d)
          ++ctx->n_conn;
    }
    double t0 = omp_get_wtime();
    ctx->t->TriangulateImage(img_id, matches);
    ctx->t_tri += omp_get_wtime() - t0;
  })
}

int ref_triangulate_image_exhaustive(ora_ctx *ctx, int img_id, int n_nb, const int32_t *nb_ids) {
  REF_TRY(ctx, {
    std::vector<int> nb(nb_ids, nb_ids + n_nb);
    int64_t n_long = 0;  // statistic only, see ref_triangulate_image
    for (const Line2d &l : ctx->t->all_lines_2d_.at(img_id)) n_long += l.length() > ctx->cfg.min_length_2d ? 1 : 0;
    for (int k = 0; k < n_nb; ++k) ctx->n_conn += n_long * int64_t(ctx->segs.at(nb[k]).size());
    double t0 = omp_get_wtime();
    ctx->t->TriangulateImageExhaustiveMatch(img_id, nb);
    ctx->t_tri += omp_get_wtime() - t0;
  })
}

int ref_compute_tracks(ora_ctx *ctx) {
  REF_TRY(ctx, {
    double t0 = omp_get_wtime();
    ctx->t->ComputeLineTracks();
    ctx->t_tail += omp_get_wtime() - t0;
    Graph g;  // the graph of ComputeLineTracks is local to it: rebuilt here for its node / edge counts only
    ctx->t->run_clustering(&g);
    ctx->graph_nodes = int64_t(g.nodes.size());
    ctx->graph_edges = int64_t(g.undirected_edges.size());
    g.Clear();
  })
}

int64_t ref_num_nodes(ora_ctx *ctx) {
  int64_t n = 0;
  for (int id : ctx->ids) n += int64_t(ctx->t->CountLines(id));
  return n;
}

int ref_get_num_tris(ora_ctx *ctx, int32_t *out) {
  int64_t g = 0;
  for (int id : ctx->ids)
    for (auto &v : ctx->t->tris_.at(id)) out[g++] = int32_t(v.size());
  return 0;
}

int ref_get_best(ora_ctx *ctx, double *out_line10, double *out_score, int32_t *out_src2, uint8_t *out_has_best) {
  int64_t g = 0;
  for (int id : ctx->ids) {
    auto &best = ctx->t->tris_best_.at(id);
    auto &all = ctx->t->tris_.at(id);
    for (size_t l = 0; l < best.size(); ++l, ++g) {
      const bool hb = !all[l].empty();  // a node without candidates keeps its value-initialised TriTuple
      out_has_best[g] = hb;
      if (hb) {
        line_to10(std::get<0>(best[l]), out_line10 + 10 * g);
        out_score[g] = std::get<1>(best[l]);
        out_src2[2 * g] = std::get<2>(best[l]).first;
        out_src2[2 * g + 1] = std::get<2>(best[l]).second;
      } else {
        for (int k = 0; k < 10; ++k) out_line10[10 * g + k] = 0.0;
        out_score[g] = 0.0;
        out_src2[2 * g] = out_src2[2 * g + 1] = 0;
      }
    }
  }
  return 0;
}

int64_t ref_num_valid_edges(ora_ctx *ctx) {
  int64_t n = 0;
  for (int id : ctx->ids)
    for (auto &v : ctx->t->valid_edges_.at(id)) n += int64_t(v.size());
  return n;
}
int ref_get_valid_edges(ora_ctx *ctx, int64_t *out_off, int32_t *out_edges2) {
  int64_t g = 0, e = 0;
  out_off[0] = 0;
  for (int id : ctx->ids)
    for (auto &v : ctx->t->valid_edges_.at(id)) {
      for (auto &p : v) {
        out_edges2[2 * e] = p.first;
        out_edges2[2 * e + 1] = p.second;
        ++e;
      }
      out_off[++g] = e;
    }
  return 0;
}

int64_t ref_num_all_tris(ora_ctx *ctx) {
  int64_t n = 0;
  for (int id : ctx->ids)
    for (auto &v : ctx->t->tris_.at(id)) n += int64_t(v.size());
  return n;
}
int ref_get_all_tris(ora_ctx *ctx, int64_t *out_off, double *out_line10, double *out_score, int32_t *out_src2) {
  int64_t g = 0, t = 0;
  out_off[0] = 0;
  for (int id : ctx->ids)
    for (auto &v : ctx->t->tris_.at(id)) {
      for (auto &tr : v) {
        line_to10(std::get<0>(tr), out_line10 + 10 * t);
        out_score[t] = std::get<1>(tr);
        out_src2[2 * t] = std::get<2>(tr).first;
        out_src2[2 * t + 1] = std::get<2>(tr).second;
        ++t;
      }
      out_off[++g] = t;
    }
  return 0;
}

int64_t ref_num_tracks(ora_ctx *ctx) { return int64_t(ctx->t->GetTracks().size()); }
int64_t ref_num_track_members(ora_ctx *ctx) {
  int64_t n = 0;
  for (auto &tr : ctx->t->GetTracks()) n += int64_t(tr.count_lines());
  return n;
}
int ref_get_tracks(ora_ctx *ctx, double *out_line7, int64_t *out_off, int32_t *out_img_ids, int32_t *out_line_ids,
                   int32_t *out_node_ids, double *out_scores, double *out_line3d10) {
  int64_t e = 0, ti = 0;
  out_off[0] = 0;
  for (auto &tr : ctx->t->GetTracks()) {
    double *o = out_line7 + 7 * ti;
    for (int k = 0; k < 3; ++k) { o[k] = tr.line.start[k]; o[3 + k] = tr.line.end[k]; }
    o[6] = tr.line.uncertainty;
    for (size_t k = 0; k < tr.count_lines(); ++k, ++e) {
      out_img_ids[e] = tr.image_id_list[k];
      out_line_ids[e] = tr.line_id_list[k];
      out_node_ids[e] = tr.node_id_list[k];
      out_scores[e] = tr.score_list[k];
      const Line3d &l3 = tr.line3d_list[k];
      double *p = out_line3d10 + 10 * e;
      for (int q = 0; q < 3; ++q) {
        p[q] = l3.start[q];
        p[3 + q] = l3.end[q];
      }
      p[6] = l3.depths[0]; p[7] = l3.depths[1]; p[8] = l3.uncertainty; p[9] = l3.score;
    }
    out_off[++ti] = e;
  }
  return 0;
}

int ref_get_stats(ora_ctx *ctx, int64_t out[8]) {
  int64_t cand = 0, pairs = 0;
  for (int id : ctx->ids)
    for (auto &v : ctx->t->tris_.at(id)) {
      cand += int64_t(v.size());
      pairs += int64_t(v.size()) * int64_t(v.size());
    }
  out[0] = ctx->n_conn; out[1] = cand; out[2] = pairs; out[3] = ref_num_valid_edges(ctx);
  out[4] = ctx->graph_nodes; out[5] = ctx->graph_edges; out[6] = ref_num_tracks(ctx); out[7] = 0;
  return 0;
}
int ref_get_timers(ora_ctx *ctx, double out[4]) {
  out[0] = ctx->t_tri; out[1] = 0; out[2] = ctx->t_tail; out[3] = 0;  // generation and scoring run inside one call
  return 0;
}

// ---- post-triangulation filters + remerge ----
ora_trackset *ref_ts_from_ctx(ora_ctx *ctx) {
  auto *ts = new ora_trackset();
  ts->tracks = ctx->t->GetTracks();
  return ts;
}
void ref_ts_destroy(ora_trackset *ts) { delete ts; }
int64_t ref_ts_num_tracks(ora_trackset *ts) { return int64_t(ts->tracks.size()); }
int64_t ref_ts_num_members(ora_trackset *ts) {
  int64_t n = 0;
  for (auto &t : ts->tracks) n += int64_t(t.count_lines());
  return n;
}
int ref_ts_get(ora_trackset *ts, double *line7, uint8_t *active, int64_t *off, int32_t *img, int32_t *lid, int32_t *nid,
               double *score, double *line2d4, double *line3d10) {
  int64_t e = 0, ti = 0;
  off[0] = 0;
  for (auto &tr : ts->tracks) {
    double *o = line7 + 7 * ti;
    for (int k = 0; k < 3; ++k) { o[k] = tr.line.start[k]; o[3 + k] = tr.line.end[k]; }
    o[6] = tr.line.uncertainty;
    active[ti] = tr.active ? 1 : 0;
    for (size_t k = 0; k < tr.count_lines(); ++k, ++e) {
      img[e] = tr.image_id_list[k]; lid[e] = tr.line_id_list[k]; nid[e] = tr.node_id_list[k];
      score[e] = tr.score_list[k];
      line2d4[4 * e] = tr.line2d_list[k].start[0]; line2d4[4 * e + 1] = tr.line2d_list[k].start[1];
      line2d4[4 * e + 2] = tr.line2d_list[k].end[0]; line2d4[4 * e + 3] = tr.line2d_list[k].end[1];
      line_to10(tr.line3d_list[k], line3d10 + 10 * e);
    }
    off[++ti] = e;
  }
  return 0;
}
int ref_ts_filter_by_reprojection(ora_ctx *ctx, ora_trackset *ts, double th_angular2d, double th_perp2d, int num_outliers) {
  REF_TRY(ctx, {
    std::vector<LineTrack> out;
    merging::FilterSupportingLines(out, ts->tracks, *ctx->imagecols, th_angular2d, th_perp2d, num_outliers);
    ts->tracks = out;
  })
}
int ref_ts_filter_by_sensitivity(ora_ctx *ctx, ora_trackset *ts, double th_angular3d, int min_supports) {
  REF_TRY(ctx, {
    std::vector<LineTrack> out;
    merging::FilterTracksBySensitivity(out, ts->tracks, *ctx->imagecols, th_angular3d, min_supports);
    ts->tracks = out;
  })
}
int ref_ts_filter_by_overlap(ora_ctx *ctx, ora_trackset *ts, double th_overlap, int min_supports) {
  REF_TRY(ctx, {
    std::vector<LineTrack> out;
    merging::FilterTracksByOverlap(out, ts->tracks, *ctx->imagecols, th_overlap, min_supports);
    ts->tracks = out;
  })
}
int ref_ts_remerge_once(ora_ctx *ctx, ora_trackset *ts, const ora_config *linker_cfg, int num_outliers) {
  REF_TRY(ctx, {
    LineLinker3d l3(linker3d_dict(*linker_cfg));
    ts->tracks = merging::RemergeLineTracks(ts->tracks, l3, num_outliers);
  })
}

// ---- free functions ----
void ref_get_normal_direction(const double seg[4], const double cam[11], double out[3]) {
  V3D n = tri::getNormalDirection(seg_to_line(seg), view_from_cam11(cam));
  for (int k = 0; k < 3; ++k) out[k] = n[k];
}
static void m3_out(const M3D &m, double out[9]) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) out[3 * i + j] = m(i, j);
}
void ref_compute_essential_matrix(const double cam1[11], const double cam2[11], double out[9]) {
  m3_out(tri::compute_essential_matrix(view_from_cam11(cam1), view_from_cam11(cam2)), out);
}
void ref_compute_fundamental_matrix(const double cam1[11], const double cam2[11], double out[9]) {
  m3_out(tri::compute_fundamental_matrix(view_from_cam11(cam1), view_from_cam11(cam2)), out);
}
double ref_compute_epipolar_IoU(const double seg1[4], const double cam1[11], const double seg2[4], const double cam2[11]) {
  return tri::compute_epipolar_IoU(seg_to_line(seg1), view_from_cam11(cam1), seg_to_line(seg2), view_from_cam11(cam2));
}
int ref_triangulate_point(const double p1[2], const double cam1[11], const double p2[2], const double cam2[11],
                          double out[3]) {
  auto r = tri::triangulate_point(V2D(p1[0], p1[1]), view_from_cam11(cam1), V2D(p2[0], p2[1]), view_from_cam11(cam2));
  for (int k = 0; k < 3; ++k) out[k] = r.first[k];
  return r.second ? 1 : 0;
}
void ref_triangulate_line(const double seg1[4], const double cam1[11], const double seg2[4], const double cam2[11],
                          double out10[10]) {
  line_to10(tri::triangulate_line(seg_to_line(seg1), view_from_cam11(cam1), seg_to_line(seg2), view_from_cam11(cam2)), out10);
}
void ref_triangulate_line_by_endpoints(const double seg1[4], const double cam1[11], const double seg2[4],
                                       const double cam2[11], double out10[10]) {
  line_to10(tri::triangulate_line_by_endpoints(seg_to_line(seg1), view_from_cam11(cam1), seg_to_line(seg2),
                                               view_from_cam11(cam2)), out10);
}
void ref_cam_project(const double cam[11], const double p[3], double out[2]) {
  V2D q = view_from_cam11(cam).projection(V3D(p[0], p[1], p[2]));
  out[0] = q[0]; out[1] = q[1];
}
void ref_cam_ray_direction(const double cam[11], const double p2d[2], double out[3]) {
  V3D r = view_from_cam11(cam).ray_direction(V2D(p2d[0], p2d[1]));
  for (int k = 0; k < 3; ++k) out[k] = r[k];
}
void ref_get_direction_from_vp(const double vp[3], const double cam[11], double out[3]) {
  V3D r = tri::getDirectionFromVP(V3D(vp[0], vp[1], vp[2]), view_from_cam11(cam));
  for (int k = 0; k < 3; ++k) out[k] = r[k];
}
void ref_triangulate_line_with_direction(const double seg1[4], const double cam1[11], const double seg2[4],
                                         const double cam2[11], const double dir[3], double out10[10]) {
  line_to10(tri::triangulate_line_with_direction(seg_to_line(seg1), view_from_cam11(cam1), seg_to_line(seg2),
                                                 view_from_cam11(cam2), V3D(dir[0], dir[1], dir[2])), out10);
}
void ref_triangulate_line_with_one_point(const double seg1[4], const double cam1[11], const double seg2[4],
                                         const double cam2[11], const double point[3], double out10[10]) {
  line_to10(tri::triangulate_line_with_one_point(seg_to_line(seg1), view_from_cam11(cam1), seg_to_line(seg2),
                                                 view_from_cam11(cam2), V3D(point[0], point[1], point[2])), out10);
}
double ref_cam_projdepth(const double cam[11], const double p[3]) {
  return view_from_cam11(cam).pose.projdepth(V3D(p[0], p[1], p[2]));
}
void ref_cam_R(const double cam[11], double out[9]) { m3_out(view_from_cam11(cam).R(), out); }
void ref_cam_center(const double cam[11], double out[3]) {
  V3D c = view_from_cam11(cam).pose.center();
  for (int k = 0; k < 3; ++k) out[k] = c[k];
}
double ref_line3d_sensitivity(const double line10[10], const double cam[11]) {
  return line_from10(line10).sensitivity(view_from_cam11(cam));
}
double ref_line3d_uncertainty(const double line10[10], const double cam[11], double var2d) {
  return line_from10(line10).computeUncertainty(view_from_cam11(cam), var2d);
}
double ref_linker2d_score(const ora_config *cfg, const double seg1[4], const double seg2[4]) {
  LineLinker2d l2(linker2d_dict(*cfg));
  return l2.compute_score(seg_to_line(seg1), seg_to_line(seg2));
}
double ref_linker3d_score(const ora_config *cfg, int mode3d, const double a[10], const double b[10]) {
  LineLinker3d l3(linker3d_dict(*cfg));
  if (mode3d == 1) l3.config.set_to_shared_parent_scoring();
  if (mode3d == 2) l3.config.set_to_spatial_merging();
  if (mode3d == 3) l3.config.set_to_avgtest_merging();
  return l3.compute_score(line_from10(a), line_from10(b));
}
int ref_track_labels_greedy(int n_nodes, const int32_t *node_img, int64_t n_edges, const double *edge_sim,
                            const int32_t *edge_nodes2, int32_t *out_labels) {
  Graph g;
  std::vector<PatchNode *> nodes;
  for (int i = 0; i < n_nodes; ++i) nodes.push_back(g.FindOrCreateNode(node_img[i], size_t(i)));
  for (int64_t e = 0; e < n_edges; ++e) g.AddEdge(nodes[edge_nodes2[2 * e]], nodes[edge_nodes2[2 * e + 1]], edge_sim[e]);
  std::vector<Line3d> lines(static_cast<size_t>(n_nodes));
  std::vector<int> labels = merging::ComputeLineTrackLabelsGreedy(g, lines);
  for (int i = 0; i < n_nodes; ++i) out_labels[i] = labels[size_t(i)];
  g.Clear();
  return 0;
}
void ref_aggregate_line3d_list(int n, const double *lines10, const double *scores, int num_outliers, double out7[7]) {
  std::vector<Line3d> lines;
  std::vector<double> sc(scores, scores + n);
  for (int i = 0; i < n; ++i) lines.push_back(line_from10(lines10 + 10 * i));
  Line3d r = merging::Aggregator::aggregate_line3d_list(lines, sc, num_outliers);
  for (int k = 0; k < 3; ++k) { out7[k] = r.start[k]; out7[3 + k] = r.end[k]; }
  out7[6] = r.uncertainty;
}

// LineTrack::Write / LineTrack::Read (base/linetrack.cc:133-209, 211-270) over flat arrays: the reference's own file
// format code, for the byte-level comparison with limap_amd/io.py (tests/test_io_formats.py).  These two exist only
// in the reference-backed library (no ora_* counterpart).  flags: bit 0 node_id_list, bit 1 score_list, bit 2 line3d_list.
int ref_track_write(const char *filename, const double line6[6], int n, const int32_t *img_ids, const int32_t *line_ids,
                    const double *line2d4, int flags, const int32_t *node_ids, const double *scores, const double *line3d6) {
  LineTrack tr;
  tr.line = Line3d(V3D(line6[0], line6[1], line6[2]), V3D(line6[3], line6[4], line6[5]));
  for (int i = 0; i < n; ++i) {
    tr.image_id_list.push_back(img_ids[i]);
    tr.line_id_list.push_back(line_ids[i]);
    tr.line2d_list.push_back(Line2d(V2D(line2d4[4 * i], line2d4[4 * i + 1]), V2D(line2d4[4 * i + 2], line2d4[4 * i + 3])));
    if (flags & 1) tr.node_id_list.push_back(node_ids[i]);
    if (flags & 2) tr.score_list.push_back(scores[i]);
    if (flags & 4)
      tr.line3d_list.push_back(Line3d(V3D(line3d6[6 * i], line3d6[6 * i + 1], line3d6[6 * i + 2]),
                                      V3D(line3d6[6 * i + 3], line3d6[6 * i + 4], line3d6[6 * i + 5])));
  }
  try {
    tr.Write(filename);
  } catch (const std::exception &) {
    return 1;
  }
  return 0;
}
// returns the number of supporting lines (-1: error, -2: more than cap); the aux arrays are filled as far as the file
// has them (LineTrack::Read resizes all lists to n and returns early at "END")
int ref_track_read(const char *filename, double line6[6], int cap, int32_t *img_ids, int32_t *line_ids, double *line2d4,
                   int32_t *node_ids, double *scores, double *line3d6) {
  LineTrack tr;
  try {
    tr.Read(filename);
  } catch (const std::exception &) {
    return -1;
  }
  const int n = (int)tr.image_id_list.size();
  if (n > cap) return -2;
  for (int k = 0; k < 3; ++k) { line6[k] = tr.line.start[k]; line6[3 + k] = tr.line.end[k]; }
  for (int i = 0; i < n; ++i) {
    img_ids[i] = tr.image_id_list[i];
    line_ids[i] = tr.line_id_list[i];
    const Line2d &l = tr.line2d_list[i];
    line2d4[4 * i] = l.start[0]; line2d4[4 * i + 1] = l.start[1]; line2d4[4 * i + 2] = l.end[0]; line2d4[4 * i + 3] = l.end[1];
    node_ids[i] = (size_t)i < tr.node_id_list.size() ? tr.node_id_list[i] : 0;
    scores[i] = (size_t)i < tr.score_list.size() ? tr.score_list[i] : 0.0;
    for (int k = 0; k < 3; ++k) {
      line3d6[6 * i + k] = (size_t)i < tr.line3d_list.size() ? tr.line3d_list[i].start[k] : 0.0;
      line3d6[6 * i + 3 + k] = (size_t)i < tr.line3d_list.size() ? tr.line3d_list[i].end[k] : 0.0;
    }
  }
  return n;
}

// ImageCollection::as_dict() (base/image_collection.cc:158-171, camera.cc:265-294) of a collection built like ref_init
// builds it, plus ImageCollection(py::dict) -> arrays for the way back: what limap's imagecols.npy holds.  Returns a new
// reference to the dict (call through ctypes.PyDLL with restype py_object).
PyObject *ref_imagecols_as_dict(int n_img, const int32_t *img_ids, const double *kvec, const double *qvec, const double *tvec) {
  try {
    std::map<int, Camera> cameras;
    std::map<int, CameraImage> images;
    for (int i = 0; i < n_img; ++i) {
      const int id = img_ids[i];
      Camera cam(1, std::vector<double>{kvec[4 * i], kvec[4 * i + 1], kvec[4 * i + 2], kvec[4 * i + 3]}, i);
      cameras.insert(std::make_pair(i, cam));
      CameraPose pose(V4D(qvec[4 * i], qvec[4 * i + 1], qvec[4 * i + 2], qvec[4 * i + 3]),
                      V3D(tvec[3 * i], tvec[3 * i + 1], tvec[3 * i + 2]));
      images.insert(std::make_pair(id, CameraImage(i, pose)));
    }
    ImageCollection ic(cameras, images);
    py::dict d = ic.as_dict();
    return d.release().ptr();
  } catch (const std::exception &) {
    Py_RETURN_NONE;
  }
}
// ImageCollection(py::dict) as the reference parses it: out arrays in ascending image id; returns the image count
int ref_imagecols_from_dict(PyObject *dict, int cap, int32_t *img_ids, double *kvec, double *qvec, double *tvec) {
  try {
    ImageCollection ic(py::reinterpret_borrow<py::dict>(dict));
    std::vector<int> ids = ic.get_img_ids();
    if ((int)ids.size() > cap) return -2;
    for (size_t i = 0; i < ids.size(); ++i) {
      CameraView v = ic.camview(ids[i]);
      img_ids[i] = ids[i];
      M3D K = v.K();
      kvec[4 * i] = K(0, 0); kvec[4 * i + 1] = K(1, 1); kvec[4 * i + 2] = K(0, 2); kvec[4 * i + 3] = K(1, 2);
      for (int k = 0; k < 4; ++k) qvec[4 * i + k] = v.pose.qvec[k];
      for (int k = 0; k < 3; ++k) tvec[3 * i + k] = v.pose.tvec[k];
    }
    return (int)ids.size();
  } catch (const std::exception &) {
    return -1;
  }
}

// Hash of the tree files this library was compiled against (stand-in headers, eigen_svd_ref.h, lt_oracle.h, this file):
// oracle/Makefile passes it (oracle/ref.py --hash), tests/test_oracle_vs_ref.py compares it with the tree's, so a prebuilt
// _ref that is older than the headers it embeds cannot pass for current.
#ifndef REF_SOURCE_HASH
#define REF_SOURCE_HASH "unknown"
#endif
const char *ref_source_hash() { return REF_SOURCE_HASH; }

}  // extern "C"
