/*
 * lt_oracle.h -- C interface of the CPU ORACLE for the line-triangulation hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This library is a plain C++17 (no Eigen) CPU restatement of
 * the reference algorithm (cvg/limap, src/limap/triangulation + the slices of src/limap/base
 * and src/limap/merging it calls).  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it.  The product (limap_amd/) never links, imports or executes it.
 *
 * PINNED against the reference's own sources since round 2: oracle/_ref (oracle/Makefile `ref`,
 * oracle/ref_driver.cpp) compiles the unmodified hot-path files of /root/reference/src/limap against
 * stand-in Eigen / COLMAP / PoseLib headers (oracle/ref_shim/ -- those libraries are not on disk), and
 * tests/test_oracle_vs_ref.py holds this restatement to it bit for bit; the reference's own tests hold no
 * golden vectors for this path, so tests/golden/*.npz are outputs of the reference
 * run here.  Still assumptions: Eigen's internal evaluation order at the ulp level (shared by the stand-in),
 * the SVD sign, PoseLib's quartic root finder.  Analytic known-answer tests: tests/test_oracle_kat.py.
 */
#ifndef LT_ORACLE_H
#define LT_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Mirrors BaseLineTriangulatorConfig + GlobalLineTriangulatorConfig +
 * LineLinker2dConfig + LineLinker3dConfig
 * (reference: triangulation/base_line_triangulator.h:20-43,
 *  triangulation/global_line_triangulator.h:11-24, base/line_linker.h:18-52,88-151).
 * Defaults are the C++ defaults of the reference (NOT the yaml values). */
typedef struct ora_config {
  /* base triangulator */
  int32_t debug_mode;
  int32_t add_halfpix;
  int32_t use_vp;                       /* VP-guided proposals (needs ora_init_vp) */
  int32_t use_endpoints_triangulation;
  int32_t disable_many_points_triangulation;
  int32_t disable_one_point_triangulation;
  int32_t disable_algebraic_triangulation;
  int32_t disable_vp_triangulation;
  double min_length_2d;
  double line_tri_angle_threshold;
  double IoU_threshold;
  double sensitivity_threshold;
  double var2d;
  /* global triangulator */
  double fullscore_th;
  int32_t max_valid_conns;
  int32_t min_num_outer_edges;
  int32_t merging_strategy;             /* 0 greedy, 1 exhaustive, 2 avg (merging/merging.cc:18-368) */
  int32_t num_outliers_aggregator;
  /* linker 2d */
  double l2_score_th, l2_th_angle, l2_th_overlap, l2_th_smartoverlap, l2_th_smartangle,
      l2_th_perp, l2_th_innerseg;
  int32_t l2_use_angle, l2_use_overlap, l2_use_smartangle, l2_use_perp, l2_use_innerseg;
  int32_t _pad0;
  /* linker 3d */
  double l3_score_th, l3_th_angle, l3_th_overlap, l3_th_smartoverlap, l3_th_smartangle,
      l3_th_perp, l3_th_innerseg, l3_th_scaleinv;
  int32_t l3_use_angle, l3_use_overlap, l3_use_smartangle, l3_use_perp, l3_use_innerseg,
      l3_use_scaleinv;
} ora_config;

typedef struct ora_ctx ora_ctx;

void ora_config_default(ora_config *cfg);

/* faithful != 0 reproduces the reference's cost structure (by-value CameraView copies with
 * heap allocations, per-call R()/K_inv() recomputation); 0 hoists nothing either -- the
 * arithmetic is identical, only the copies are skipped. */
ora_ctx *ora_create(const ora_config *cfg, int faithful);
void ora_destroy(ora_ctx *ctx);
const char *ora_last_error(ora_ctx *ctx);
/* OpenMP threads used by the two parallel loops (default: min(host cores, 16) set by oracle.py;
 * the reference uses OMP's default = all cores) */
void ora_set_num_threads(int n);
int ora_get_max_threads(void);

int ora_set_ranges(ora_ctx *ctx, const double lo[3], const double hi[3]);
int ora_unset_ranges(ora_ctx *ctx);

/* kvec = (fx, fy, cx, cy) per image; qvec = (w,x,y,z) world->cam; tvec; segs = (x1,y1,x2,y2).
 * seg_off has n_img+1 entries. */
int ora_init(ora_ctx *ctx, int n_img, const int32_t *img_ids, const double *kvec,
             const double *qvec, const double *tvec, const int64_t *seg_off, const double *segs);

/* matches for neighbour k are rows m_off[k]..m_off[k+1] of m_pairs (line_id, ng_line_id).
 * Neighbours are processed in ascending nb id order like the reference's std::map. */
/* SetBipartites2d (base_line_triangulator.h:71-76): per image its 2D points (id, xy, point3D_id; CSR pt_off)
 * and per line (CSR line_off over images, lp_off over lines) the ids of the neighbouring points.
 * SetSfMPoints (:77): point3D_id -> xyz.  Enable the many-points proposal (lines 183-236); the one-point
 * proposal is not restated (disable_one_point_triangulation must be set). */
int ora_set_bipartites(ora_ctx *ctx, int n_img, const int32_t *img_ids, const int64_t *pt_off, const int32_t *pt_ids,
                       const double *pt_xy, const int32_t *pt_p3d, const int64_t *line_off, const int64_t *lp_off,
                       const int32_t *lp_ptids);
int ora_set_sfm_points(ora_ctx *ctx, int64_t n, const int32_t *ids, const double *xyz);
/* InitVPResults (base_line_triangulator.h:47-49): per image the VP label of every line (-1 = none) and
 * the VP vectors (vplib/vpbase.h:18-47), CSR over the images of ora_init */
int ora_init_vp(ora_ctx *ctx, int n_img, const int32_t *img_ids, const int64_t *label_off, const int32_t *labels,
                const int64_t *vp_off, const double *vps);
int ora_triangulate_image(ora_ctx *ctx, int img_id, int n_nb, const int32_t *nb_ids,
                          const int64_t *m_off, const int32_t *m_pairs);
int ora_triangulate_image_exhaustive(ora_ctx *ctx, int img_id, int n_nb, const int32_t *nb_ids);
int ora_compute_tracks(ora_ctx *ctx);

/* ---- getters (node order = images ascending id, lines ascending) ---- */
int64_t ora_num_nodes(ora_ctx *ctx);
/* per node: n candidates generated (counted even when debug_mode==0) */
int ora_get_num_tris(ora_ctx *ctx, int32_t *out_n_tris);
/* per node best: line[10] = start3,end3,depths2,uncertainty,score3d(line.score);
 * score = multi-view score; src = (ng_img_id, ng_line_id); has_best = 0 if node had no candidate */
int ora_get_best(ora_ctx *ctx, double *out_line10, double *out_score, int32_t *out_src2,
                 uint8_t *out_has_best);
/* valid edges CSR: off[n_nodes+1]; edges (neighbor_index, ng_line_id) pairs */
int64_t ora_num_valid_edges(ora_ctx *ctx);
int ora_get_valid_edges(ora_ctx *ctx, int64_t *out_off, int32_t *out_edges2);
/* all candidates (debug_mode only): CSR off[n_nodes+1], line10, score, src2 */
int64_t ora_num_all_tris(ora_ctx *ctx);
int ora_get_all_tris(ora_ctx *ctx, int64_t *out_off, double *out_line10, double *out_score,
                     int32_t *out_src2);
/* tracks */
int64_t ora_num_tracks(ora_ctx *ctx);
int64_t ora_num_track_members(ora_ctx *ctx);
int ora_get_tracks(ora_ctx *ctx, double *out_line7 /* start3,end3,uncertainty */,
                   int64_t *out_off /* T+1 */, int32_t *out_img_ids, int32_t *out_line_ids,
                   int32_t *out_node_ids, double *out_scores,
                   double *out_line3d10 /* per support: start3, end3, depths2, uncertainty, score */);
/* stats: [0] connections tested, [1] candidates, [2] candidate pairs visited in scoring,
 * [3] valid edges, [4] graph nodes, [5] graph edges, [6] tracks */
int ora_get_stats(ora_ctx *ctx, int64_t out[8]);
/* wall-clock split of the last run in seconds: [0] generation, [1] scoring, [2] tail */
int ora_get_timers(ora_ctx *ctx, double out[4]);

/* ---- post-triangulation filters + remerge (merging/merging_utils.cc:27-155,
 * merging/merging.cc:513-644; called from runners/line_triangulation.py:171-200) on a copy of the
 * context's tracks.  Cameras are the ones given to ora_init. ---- */
typedef struct ora_trackset ora_trackset;
ora_trackset *ora_ts_from_ctx(ora_ctx *ctx);
void ora_ts_destroy(ora_trackset *ts);
int64_t ora_ts_num_tracks(ora_trackset *ts);
int64_t ora_ts_num_members(ora_trackset *ts);
int ora_ts_get(ora_trackset *ts, double *line7, uint8_t *active, int64_t *off, int32_t *img, int32_t *lid,
               int32_t *nid, double *score, double *line2d4, double *line3d10);
int ora_ts_filter_by_reprojection(ora_ctx *ctx, ora_trackset *ts, double th_angular2d, double th_perp2d,
                                  int num_outliers);
int ora_ts_filter_by_sensitivity(ora_ctx *ctx, ora_trackset *ts, double th_angular3d, int min_supports);
int ora_ts_filter_by_overlap(ora_ctx *ctx, ora_trackset *ts, double th_overlap, int min_supports);
int ora_ts_remerge_once(ora_ctx *ctx, ora_trackset *ts, const ora_config *linker_cfg, int num_outliers);

/* ---- free functions (mirror triangulation/bindings.cc:22-31) on raw arrays ----
 * cam = kvec[4] | qvec[4] | tvec[3]  (11 doubles), seg = x1,y1,x2,y2 */
void ora_get_normal_direction(const double seg[4], const double cam[11], double out[3]);
void ora_compute_essential_matrix(const double cam1[11], const double cam2[11], double out[9]);
void ora_compute_fundamental_matrix(const double cam1[11], const double cam2[11], double out[9]);
double ora_compute_epipolar_IoU(const double seg1[4], const double cam1[11], const double seg2[4],
                                const double cam2[11]);
/* returns 1 on success (cheirality passed) */
int ora_triangulate_point(const double p1[2], const double cam1[11], const double p2[2],
                          const double cam2[11], double out[3]);
/* out10 = start3,end3,depths2,uncertainty(-1),score ; score == -1 marks the failure sentinel */
void ora_triangulate_line(const double seg1[4], const double cam1[11], const double seg2[4],
                          const double cam2[11], double out10[10]);
void ora_triangulate_line_by_endpoints(const double seg1[4], const double cam1[11],
                                       const double seg2[4], const double cam2[11],
                                       double out10[10]);
/* camera helpers */
void ora_cam_project(const double cam[11], const double p[3], double out[2]);
void ora_cam_ray_direction(const double cam[11], const double p2d[2], double out[3]);
void ora_get_direction_from_vp(const double vp[3], const double cam[11], double out[3]);  /* functions.cc:37-42 */
void ora_triangulate_line_with_direction(const double seg1[4], const double cam1[11], const double seg2[4],
                                         const double cam2[11], const double dir[3],
                                         double out10[10]);                              /* functions.cc:385-442 */
/* one-point proposal: 1 (default) = the reference's generated solver evaluated term by term (bit-identical to
   oracle/_ref), 0 = the restated optimisation problem (what the device code follows; equal to 1e-6 relative) */
void ora_set_one_point_solver(int generated);
int ora_get_one_point_solver(void);
void ora_triangulate_line_with_one_point(const double seg1[4], const double cam1[11], const double seg2[4],
                                         const double cam2[11], const double point[3],
                                         double out10[10]);                              /* functions.cc:325-383 */
double ora_cam_projdepth(const double cam[11], const double p[3]);
void ora_cam_R(const double cam[11], double out[9]);
void ora_cam_center(const double cam[11], double out[3]);
/* line helpers; line10 layout as above */
double ora_line3d_sensitivity(const double line10[10], const double cam[11]);
double ora_line3d_uncertainty(const double line10[10], const double cam[11], double var2d);
/* linkers.  mode3d: 0 = as configured, 1 = shared-parent scoring, 2 = spatial merging,
 * 3 = avgtest merging (line_linker.h:115-137) */
double ora_linker2d_score(const ora_config *cfg, const double seg1[4], const double seg2[4]);
double ora_linker3d_score(const ora_config *cfg, int mode3d, const double line1_10[10],
                          const double line2_10[10]);
/* greedy union-find labels (merging/merging.cc:18-103): nodes carry an image id, edges
 * (sim, node1, node2); out_labels[n_nodes] (-1 = not in a track) */
int ora_track_labels_greedy(int n_nodes, const int32_t *node_img, int64_t n_edges,
                            const double *edge_sim, const int32_t *edge_nodes2,
                            int32_t *out_labels);
/* aggregator (merging/aggregator.cc:53-101): lines10[n], scores[n] -> out7 (start,end,unc) */
void ora_aggregate_line3d_list(int n, const double *lines10, const double *scores,
                               int num_outliers, double out7[7]);

#ifdef __cplusplus
}
#endif
#endif /* LT_ORACLE_H */
