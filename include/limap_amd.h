/*
 * limap_amd.h -- C ABI of the MI355X-native line-triangulation backend (liblimap_amd.so).
 *
 * Drop-in boundary for the hot path of cvg/limap's `limap.triangulation.GlobalLineTriangulator`
 * (reference pybind surface: src/limap/triangulation/bindings.cc:19-32,78-119; only production
 * caller: src/limap/runners/line_triangulation.py:102-168).  Plain pointers and sizes, no
 * exceptions across the boundary: every call returns 0 on success or a negative code, and
 * lt_last_error(ctx) holds the message the reference would have thrown.  One context per thread.
 *
 * All arithmetic is FP64 like the reference.  Image ids are arbitrary int32 values; internally
 * images are ordered by ascending id (the reference iterates std::map<int, ...>).  A "node" is an
 * (image, line) pair; global node index = seg_off[image index] + line id.
 */
#ifndef LIMAP_AMD_H
#define LIMAP_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LT_OK 0
#define LT_ERR_RUNTIME (-1)  /* std::runtime_error in the reference (bad matches, bad strategy) */
#define LT_ERR_ARGUMENT (-2) /* THROW_CHECK / std::out_of_range in the reference */
#define LT_ERR_HIP (-3)      /* HIP runtime failure */
#define LT_ERR_STATE (-4)    /* call-order violation (e.g. triangulate before init) */

/* Replaces GlobalLineTriangulatorConfig(py::dict) = BaseLineTriangulatorConfig
 * (triangulation/base_line_triangulator.h:20-43, .cc:16-31) + GlobalLineTriangulatorConfig
 * (triangulation/global_line_triangulator.h:11-24, .cc:18-29) + LineLinker2dConfig /
 * LineLinker3dConfig (base/line_linker.h:18-52,88-151, .cc:21-34,164-179), field for field.
 * lt_config_default() fills the reference's C++ defaults. */
typedef struct lt_config {
  int32_t debug_mode;
  int32_t add_halfpix;
  int32_t use_vp;                            /* VP-guided proposals (needs lt_init_vp) */
  int32_t use_endpoints_triangulation;
  int32_t disable_many_points_triangulation; /* many-points proposal (needs lt_set_bipartites) */
  int32_t disable_one_point_triangulation;
  int32_t disable_algebraic_triangulation;
  int32_t disable_vp_triangulation;
  double min_length_2d;
  double line_tri_angle_threshold;
  double IoU_threshold;
  double sensitivity_threshold;
  double var2d;
  double fullscore_th;
  int32_t max_valid_conns;
  int32_t min_num_outer_edges;
  int32_t merging_strategy; /* 0 "greedy", 1 "exhaustive", 2 "avg" (merging/merging.cc:18-368); any other
                               value -> LT_ERR_RUNTIME from lt_compute_tracks, where the reference throws
                               (global_line_triangulator.cc:314-316) */
  int32_t num_outliers_aggregator;
  double l2_score_th, l2_th_angle, l2_th_overlap, l2_th_smartoverlap, l2_th_smartangle,
      l2_th_perp, l2_th_innerseg;
  int32_t l2_use_angle, l2_use_overlap, l2_use_smartangle, l2_use_perp, l2_use_innerseg;
  int32_t _pad0;
  double l3_score_th, l3_th_angle, l3_th_overlap, l3_th_smartoverlap, l3_th_smartangle,
      l3_th_perp, l3_th_innerseg, l3_th_scaleinv;
  int32_t l3_use_angle, l3_use_overlap, l3_use_smartangle, l3_use_perp, l3_use_innerseg,
      l3_use_scaleinv;
} lt_config;

typedef struct lt_ctx lt_ctx;

void lt_config_default(lt_config *cfg);
int lt_abi_version(void);          /* bumped on any incompatible change of this header */
uint64_t lt_sizeof_config(void);   /* sizeof(lt_config) the library was built with */

/* GlobalLineTriangulator(cfg) -- bindings.cc:79-80,99.  device = HIP device ordinal.
 * Returns NULL (and writes a message to stderr) if no usable GPU / HIP runtime is present:
 * there is NO CPU fallback. */
lt_ctx *lt_create(const lt_config *cfg, int device);
void lt_destroy(lt_ctx *ctx);
const char *lt_last_error(lt_ctx *ctx);
/* run the kernels on a caller-owned hipStream_t (e.g. torch's current stream); NULL = own stream */
int lt_set_stream(lt_ctx *ctx, void *hip_stream);

/* SetRanges / UnsetRanges -- base_line_triangulator.h:61-65, bindings.cc:94-95 */
int lt_set_ranges(lt_ctx *ctx, const double lo[3], const double hi[3]);
int lt_unset_ranges(lt_ctx *ctx);

/* Init(all_2d_segs, imagecols) -- base_line_triangulator.cc:45-63, global_line_triangulator.cc:31-57.
 * kvec = (fx,fy,cx,cy) of the undistorted pinhole camera, qvec = (w,x,y,z), tvec; seg_off[n_img+1]
 * offsets into segs[.][4] = (x1,y1,x2,y2).  Host pointers; data is snapshotted (the reference
 * keeps a raw pointer to the caller's ImageCollection -- base_line_triangulator.cc:50). */
int lt_init(lt_ctx *ctx, int n_img, const int32_t *img_ids, const double *kvec, const double *qvec,
            const double *tvec, const int64_t *seg_off, const double *segs);
/* InitVPResults(vpresults) -- base_line_triangulator.h:47-49, bindings.cc:89.  Per image (any subset and
 * order of the ids given to lt_init) the VP label of every line (-1 = none; vplib/vpbase.h:35,42) and its
 * vanishing points (homogeneous image coordinates, vps[.][3]); CSR: label_off / vp_off [n_img + 1].
 * Used by the VP-guided proposals of triangulateOneNode (base_line_triangulator.cc:250-281) when
 * cfg.use_vp && !cfg.disable_vp_triangulation (both triangulation modes).  Call after lt_init. */
int lt_init_vp(lt_ctx *ctx, int n_img, const int32_t *img_ids, const int64_t *label_off, const int32_t *labels,
               const int64_t *vp_off, const double *vps);
/* SetBipartites2d(all_bpt2ds) / SetSfMPoints(points) -- base_line_triangulator.h:71-77, bindings.cc:90-91.
 * Per image (CSR pt_off) its 2D points: id, xy, point3D_id; per line (CSR line_off over the images, lp_off
 * over the lines -- every line of the image must be listed) the ids of its neighbouring points
 * (structures::PL_Bipartite2d::neighbor_points).  SfM points: point3D_id -> xyz; with none given the shared
 * points are triangulated from the two views.  Enables the many-points proposal of triangulateOneNode
 * (base_line_triangulator.cc:183-236: line fit through the shared 3D points + Pluecker projection) in
 * matched and exhaustive mode, and the one-point proposal (:238-248, one candidate per shared point, any number of
 * shared points per connection, as in the reference; see lt_fn_triangulate_line_with_one_point for the solver).
 * Call after lt_init. */
int lt_set_bipartites(lt_ctx *ctx, int n_img, const int32_t *img_ids, const int64_t *pt_off, const int32_t *pt_ids,
                      const double *pt_xy, const int32_t *pt_p3d, const int64_t *line_off, const int64_t *lp_off,
                      const int32_t *lp_ptids);
int lt_set_sfm_points(lt_ctx *ctx, int64_t n, const int32_t *ids, const double *xyz);
/* Same with kvec/qvec/tvec/segs already resident in HBM (e.g. the output of the RCCL all-gather),
 * images given in ascending id order. */
int lt_init_device(lt_ctx *ctx, int n_img, const int32_t *img_ids, const void *d_kvec,
                   const void *d_qvec, const void *d_tvec, const int64_t *seg_off,
                   const void *d_segs);

/* Re-read the scene arrays from HBM (same image set and segment counts as the last init) and
 * rebuild the per-camera / per-segment invariants on the context's stream, keeping the buffered
 * or uploaded images: the per-step entry of the multi-GPU path, called after the all-gather. */
int lt_refresh_scene_device(lt_ctx *ctx, const void *d_kvec, const void *d_qvec, const void *d_tvec,
                            const void *d_segs);

/* Multi-GPU per-step path without unpack copies: describe the scene as n_chunks chunks (one per
 * rank of the all-gather), chunk c holding images [img_begin[c], img_begin[c+1]) (indices in
 * ascending-id order) as four device arrays kvec | qvec | tvec | segs.  lt_set_scene_chunks records
 * the (persistent) buffer addresses once; lt_refresh_scene_chunks rebuilds the invariants from them
 * on the context's stream after every all-gather.  While a job is uploaded (lt_upload) only the images
 * that job references -- triangulated here, or a neighbour -- get their segment records rebuilt (a rank
 * of an N-GPU job needs ~1/N of the gathered scene); cameras are always rebuilt for all images. */
int lt_set_scene_chunks(lt_ctx *ctx, int n_chunks, const int32_t *img_begin, const void *const *d_kvec,
                        const void *const *d_qvec, const void *const *d_tvec, const void *const *d_segs);
int lt_refresh_scene_chunks(lt_ctx *ctx);

/* TriangulateImage(img_id, matches) -- base_line_triangulator.cc:71-109, bindings.cc:83.
 * Rows m_off[k]..m_off[k+1] of m_pairs[.][2] = (line_id, ng_line_id) belong to neighbour
 * nb_ids[k].  Calls are buffered; the GPU runs at the next lt_flush / lt_compute_tracks / getter
 * (observable behaviour is unchanged: results are only readable through those). */
int lt_triangulate_image(lt_ctx *ctx, int img_id, int n_nb, const int32_t *nb_ids,
                         const int64_t *m_off, const int32_t *m_pairs);
/* Same, with the (K,2) int32 row array of every neighbour given by its own pointer -- the natural
 * form of the std::map<int, Eigen::MatrixXi> argument; saves the caller a concatenation. */
int lt_triangulate_image_rows(lt_ctx *ctx, int img_id, int n_nb, const int32_t *nb_ids,
                              const int32_t *const *rows, const int64_t *n_rows);
/* The TriangulateImage loop of the caller (runners/line_triangulation.py:160-167: `for img_id in imagecols.get_img_ids():
 * Triangulator.TriangulateImage(img_id, matches)`) as ONE call: image k has the neighbours nb_ids[nb_off[k] .. nb_off[k+1])
 * and, for neighbour entry e in that range, the (n_rows[e], 2) int32 row array rows[e].  Same buffering, same validation
 * and the same errors as n calls of lt_triangulate_image_rows in the given order -- but one pass over all rows (one
 * parallel region over the (image, neighbour) blocks instead of one per call: the per-call form spends 2.1 ms of a
 * 5 ms end-to-end run on 100 calls of 0.8 MB each).  Atomic: on an error nothing of the call is kept.  No reference
 * counterpart (the reference's per-image call does the work itself). */
int lt_triangulate_all_rows(lt_ctx *ctx, int n_images, const int32_t *img_ids, const int64_t *nb_off, const int32_t *nb_ids,
                            const int32_t *const *rows, const int64_t *n_rows);
/* TriangulateImageExhaustiveMatch(img_id, neighbors) -- base_line_triangulator.cc:111-136 */
int lt_triangulate_image_exhaustive(lt_ctx *ctx, int img_id, int n_nb, const int32_t *nb_ids);

/* Staged execution of the buffered images (lt_flush = upload + run + download). */
int lt_upload(lt_ctx *ctx);     /* host staging -> HBM (matches, neighbour tables) */
int lt_run_device(lt_ctx *ctx); /* kernels only, inputs resident in HBM; repeatable */
/* The same run enqueued without waiting for it.  If the previous lt_run_device_async is still in flight it is
 * completed AFTER the new run has been enqueued, and ITS status is the return value (a streaming caller keeps
 * the device busy across the host's end-of-run bookkeeping); lt_sync completes the run in flight and returns
 * its status.  Every other entry point that touches results or inputs completes it first.  No reference
 * counterpart (the reference's TriangulateImage is synchronous host code).
 * Two configurations make the call SYNCHRONOUS in part: with extra proposals (VP, points) stage B runs twice and the host
 * waits for the candidate count between the two runs (it sizes the staging exactly; this also drains a run that was still
 * in flight); and when the bound-sized arrays of a batch would exceed 48 GB the exact count is fetched before placement. */
int lt_run_device_async(lt_ctx *ctx);
int lt_sync(lt_ctx *ctx);
int lt_download(lt_ctx *ctx);   /* per-node results -> host */
int lt_flush(lt_ctx *ctx);

/* ComputeLineTracks() -- global_line_triangulator.cc:353-359 */
int lt_compute_tracks(lt_ctx *ctx);
/* The same in two halves, for a caller that streams steps (rank 0 of a multi-GPU job; no reference counterpart):
 * _begin enqueues the device half of the tail behind the resident run -- valid-edge keys, sort, similarities, the graph
 * nodes' records into page-locked memory -- and returns; the caller may then enqueue the NEXT run (lt_run_device_async);
 * _end waits for the tail's own event, not for that run, and does the host half (graph, union-find, aggregation:
 * global_line_triangulator.cc:234-351) while the device works on the next step.  Needs the device form of the tail
 * (results of the run resident on the device; the node filter of min_num_outer_edges > 0 runs on the device too, over a
 * single context's run as over imported shards); lt_compute_tracks() == _begin + _end. */
int lt_compute_tracks_begin(lt_ctx *ctx);
int lt_compute_tracks_end(lt_ctx *ctx);

/* CountImages / CountLines -- base_line_triangulator.h:84-87 */
int64_t lt_count_images(lt_ctx *ctx);
int64_t lt_count_lines(lt_ctx *ctx, int img_id);
int64_t lt_num_nodes(lt_ctx *ctx);

/* Per-node results, node order = images ascending id x lines.
 * line10 = start3,end3,depths2,uncertainty,line.score ; score = multi-view support score;
 * src2 = (ng_img_id, ng_line_id) ; has_best = 0 for nodes without any candidate
 * (GetBestScoredTriNode / GetAllBestTris -- global_line_triangulator.cc:496-541). */
int lt_get_best(lt_ctx *ctx, double *out_line10, double *out_score, int32_t *out_src2,
                uint8_t *out_has_best);
int lt_get_num_tris(lt_ctx *ctx, int32_t *out_n_tris);
/* valid_edges_ (global_line_triangulator.cc:138-142) as CSR: (neighbour index, ng_line_id) */
int64_t lt_num_valid_edges(lt_ctx *ctx);
int lt_get_valid_edges(lt_ctx *ctx, int64_t *out_off, int32_t *out_edges2);
/* valid_flags_ (filterNodeByNumOuterEdges, global_line_triangulator.cc:168-232): 1 for nodes that keep at
 * least min_num_outer_edges valid edges to surviving nodes.  The reference fills it inside run_clustering
 * (:236), so this needs lt_compute_tracks first (LT_ERR_STATE otherwise); GetAllValidBestTris (:502-514). */
int lt_get_valid_flags(lt_ctx *ctx, uint8_t *out_flags);
/* All scored candidates of the last device run (GetScoredTrisNode; kept regardless of
 * debug_mode until the next run): CSR off[n_nodes+1], line10, score, src2. */
int64_t lt_num_all_tris(lt_ctx *ctx);
int lt_get_all_tris(lt_ctx *ctx, int64_t *out_off, double *out_line10, double *out_score,
                    int32_t *out_src2);
/* GetTracks() -- tracks as CSR over members (LineTrack fields, base/linetrack.h:33-42):
 * line7 = start3,end3,uncertainty; line3d10 = per support the Line3d of line3d_list in full: start3, end3, depths2,
 * uncertainty, score -- the post-triangulation steps read the uncertainties (merging/merging.cc:513-644 re-aggregates
 * from them), a (start, end) pair alone changes their outcome */
int64_t lt_num_tracks(lt_ctx *ctx);
int64_t lt_num_track_members(lt_ctx *ctx);
int lt_get_tracks(lt_ctx *ctx, double *out_line7, int64_t *out_off, int32_t *out_img_ids,
                  int32_t *out_line_ids, int32_t *out_node_ids, double *out_scores,
                  double *out_line3d10);

/* Multi-GPU tail: the rank that triangulated an image exports its per-node results (neighbour list,
 * best candidate per line, valid edges); the rank that runs ComputeLineTracks imports them for the
 * images it did not triangulate itself.  lt_image_results_size returns the image's line count and
 * its number of valid edges (array sizes for the export). */
int64_t lt_image_results_size(lt_ctx *ctx, int img_id, int64_t *n_edges);
int lt_export_image_results(lt_ctx *ctx, int img_id, int32_t *out_nb_ids /*[255]*/, int32_t *out_n_nb,
                            double *out_line10, double *out_score, int32_t *out_src2, int32_t *out_n_tris,
                            int64_t *out_edge_off, int32_t *out_edges2);
int lt_import_image_results(lt_ctx *ctx, int img_id, int n_nb, const int32_t *nb_ids, const double *line10,
                            const double *score, const int32_t *src2, const int32_t *n_tris,
                            const int64_t *edge_off, const int32_t *edges2);
/* The same for n images in ONE call and two flat blobs -- what a streamed job moves per chunk (limap_amd/stream.py:
 * BASELINE configs[4], runners/rome16k/triangulation.py:15-45) and what the ranks of a multi-GPU job gather to rank 0.
 * ints: n, then per image  img_id, n_nb, m (lines), ne (valid edges), nb_ids[n_nb], src[m][2], n_tris[m], edge_cnt[m],
 * edges[ne][2];  dbls: per image  line10[m][10], score[m]  (the layout of limap_amd.dist.pack_image_results, so blobs packed
 * either way are interchangeable).  lt_export_images_size returns the two lengths; lt_import_images_packed checks the
 * blob against n_ints / n_dbls and every id and count in it before it touches the context (LT_ERR_ARGUMENT otherwise). */
int lt_export_images_size(lt_ctx *ctx, int n, const int32_t *img_ids, int64_t *n_ints, int64_t *n_dbls);
int lt_export_images_packed(lt_ctx *ctx, int n, const int32_t *img_ids, int32_t *ints, double *dbls);
int lt_import_images_packed(lt_ctx *ctx, const int32_t *ints, int64_t n_ints, const double *dbls, int64_t n_dbls);

/* ---- shards of a multi-GPU run, device to device (no reference counterpart: the reference is one process).
 * Images are sharded over the ranks in id order, so a rank's nodes are one range [g_lo, g_hi) of the global node index
 * (node = first node of its image + line id).  A shard travels as two blobs -- lt_shard_node_bytes() bytes per node
 * (best candidate, score, source, candidate count of global_line_triangulator.cc:145-153, as arrays one behind the
 * other) and 8 bytes per valid edge (undirected node-pair keys of run_clustering, :243-290) -- written and read by
 * device copies; the pointers may be device or host memory.  Order of calls: every rank lt_shard_count; lt_shard_build
 * (total_keys = the sum over the ranks on the rank that merges, the own count elsewhere); the other ranks lt_shard_export;
 * the merging rank lt_shard_import once per other rank, then lt_compute_tracks (which needs the device form of the
 * tail).  With min_num_outer_edges > 0 the keys of a shard are DIRECTED (source node << kb | target node): the merging rank
 * runs filterNodeByNumOuterEdges (global_line_triangulator.cc:168-232) over the merged list before it sorts the undirected
 * form.  lt_shard_import checks on the device that every imported key
 * names two nodes of this scene as (min << kb | max) -- LT_ERR_ARGUMENT otherwise; the node blobs are taken as they are. */
int lt_shard_node_bytes(void);
int lt_shard_count(lt_ctx *ctx, int64_t *n_keys);
int lt_shard_build(lt_ctx *ctx, int64_t total_keys);
int lt_shard_export(lt_ctx *ctx, int64_t g_lo, int64_t g_hi, void *nodes_blob, void *keys_blob);
int lt_shard_import(lt_ctx *ctx, int64_t g_lo, int64_t g_hi, const void *nodes_blob, int64_t n_keys, const void *keys_blob);

/* ---- post-triangulation steps of limap.runners.line_triangulation (:171-200), SURVEY 8(f) rank 2:
 * limap.merging.filter_tracks_by_reprojection / remerge / filter_tracks_by_sensitivity /
 * filter_tracks_by_overlap (merging/merging_utils.cc:27-155, merging/merging.cc:513-644).
 * A track set is a host container of LineTracks (base/linetrack.h:21-50); cameras are those of the
 * context's Init.  member arrays: img/lid/nid int32, score f64, line2d4 = x1 y1 x2 y2,
 * line3d10 = start3 end3 depths2 uncertainty score; line7 = start3 end3 uncertainty. ---- */
typedef struct lt_trackset lt_trackset;
lt_trackset *lt_ts_from_ctx(lt_ctx *ctx); /* copy of GetTracks() with the auxiliary lists filled */
lt_trackset *lt_ts_create(int64_t n_tracks, const double *line7, const uint8_t *active, const int64_t *off,
                          const int32_t *img, const int32_t *lid, const int32_t *nid, const double *score,
                          const double *line2d4, const double *line3d10);
void lt_ts_destroy(lt_trackset *ts);
int64_t lt_ts_num_tracks(lt_trackset *ts);
int64_t lt_ts_num_members(lt_trackset *ts);
int lt_ts_get(lt_trackset *ts, double *line7, uint8_t *active, int64_t *off, int32_t *img, int32_t *lid,
              int32_t *nid, double *score, double *line2d4, double *line3d10);
/* _FilterSupportLines (merging_utils.cc:51-83) */
int lt_ts_filter_by_reprojection(lt_ctx *ctx, lt_trackset *ts, double th_angular2d, double th_perp2d,
                                 int num_outliers);
/* _FilterTracksBySensitivity (merging_utils.cc:105-128) */
int lt_ts_filter_by_sensitivity(lt_ctx *ctx, lt_trackset *ts, double th_angular3d, int min_supports);
/* _FilterTracksByOverlap (merging_utils.cc:130-155) */
int lt_ts_filter_by_overlap(lt_ctx *ctx, lt_trackset *ts, double th_overlap, int min_supports);
/* one pass of _RemergeLineTracks (merging/merging.cc:513-644); the LineLinker3d is read from the
 * l3_* fields of linker_cfg; the all-pairs connection test runs on the GPU */
int lt_ts_remerge_once(lt_ctx *ctx, lt_trackset *ts, const lt_config *linker_cfg, int num_outliers);

/* Counters of the last device run: [0] connections tested, [1] candidates, [2] ordered candidate
 * pairs swept by the scoring kernel (sum n_tris^2), [3] valid edges, [4] graph nodes,
 * [5] graph edges, [6] tracks, [7] nodes. */
int lt_get_stats(lt_ctx *ctx, int64_t out[8]);
/* HIP-event timings (ms) of the last lt_run_device on the context's stream:
 * [0] whole run, [1], [2] unused (0), [3] generation (incl. the per-pair records), [4] placement of the
 * candidates, [5] scoring (incl. its per-candidate records), [6] selection, [7] unused (0); host: [8] upload, [9] download,
 * [10] tail (lt_compute_tracks); [11] candidate pairs that reached the dense evaluation in k_score3;
 * [12] host ms spent inside lt_triangulate_image* buffering the match rows of the batch;
 * single-kernel durations (HIP events around the launch): [13] k_gates, [14] k_tri_rows (only with LT_FINE_TIMERS=2
 * in the environment at the time of the run), [15] k_score3 (default; LT_FINE_TIMERS=0 turns every per-kernel event off);
 * [16] connections that passed the stage-A gates (k_gates);
 * one-pass exhaustive mode: [17] staging slots needed (fullest region x regions), [18] staging slots provided;
 * [19] 1 when this context scores with the fused kernel because the split form's pair store overflowed once, else 0;
 * [20] 1 when stage A of the last TriangulateImage job ran in the line-slot form (k_gates_ln: one lane per line), else 0;
 * [21] 1 when this context scores in the two-kernel form because the one-kernel form (k_score_q) raised its flag once, else 0;
 * [22], [23] of the last lt_compute_tracks ([10]): its device half + graph, its edge order + union-find (the rest of [10] is
 * the track members and their aggregation).
 * [15] spans the whole scoring stage (one kernel, k_score_q, by default for TriangulateImage jobs; two in the fallback form).
 * SAMPLING: an event between two kernels costs a ~5 us bubble in the stream, so a run enqueued BEHIND one still in flight
 * (lt_run_device_async back to back) carries the stage events -- [3]-[6], [13]-[15] -- only every LT_TIMER_SAMPLE-th time
 * (environment, read per run; default 8, 1 = every run); in between those slots keep the values of the last run that
 * did.  A run that starts on an idle context always carries them; [0] is measured for every run. */
int lt_get_timers(lt_ctx *ctx, double out[24]);
/* The same slots summed over every lt_run_device since the last reset ([8]-[10], [12] are not summed), and
 * the number of runs ([16] is not summed either: lt_get_timers counts it on demand with a device readback,
 * which is why a caller that times many runs should read the sums once instead of lt_get_timers per run).
 * reset != 0 clears the sums after reading.  The sampled stage slots (above) are summed over the sampled runs and
 * scaled to the number of runs. */
int lt_get_timer_sums(lt_ctx *ctx, double out[24], int64_t *n_runs, int reset);

/* The library keeps released device blocks and page-locked staging blocks in a process-wide cache
 * (contexts are typically created once per scene; hipMalloc / hipHostMalloc / hipFree are the slow part
 * of that).  This returns the cached memory of all devices to the driver; live contexts are unaffected.
 * No reference counterpart (the reference holds everything in host std::maps). */
void lt_release_cached_memory(void);

/* Puts `blocks` page-locked staging blocks of `bytes` each (rounded up to the cache's size class) into that cache, so
 * that the first scene of a process does not pin its match-row staging inside TriangulateImage (pinning costs
 * ~0.1 ms per MB).  Used by limap_amd.warmup(); LT_OK or LT_ERR_HIP.  No reference counterpart. */
int lt_reserve_host(uint64_t bytes, int blocks);

/* ---- free functions of limap.triangulation (bindings.cc:22-31) on raw arrays, run on the GPU
 * one query per call (convenience / parity checks; the batch path is the API above).
 * cam = kvec[4] | qvec[4] | tvec[3]; seg = x1,y1,x2,y2; line10 as above. */
int lt_fn_get_normal_direction(lt_ctx *ctx, const double seg[4], const double cam[11], double out[3]);
/* get_direction_from_VP(vp, view): functions.cc:37-42 */
int lt_fn_get_direction_from_vp(lt_ctx *ctx, const double vp[3], const double cam[11], double out[3]);
/* triangulate_point(p1, view1, p2, view2) -> (point, ok): functions.cc:100-117 */
int lt_fn_triangulate_point(lt_ctx *ctx, const double p1[2], const double cam1[11], const double p2[2],
                            const double cam2[11], double out[3], int *ok);
/* triangulate_line_with_direction(l1, view1, l2, view2, direction): functions.cc:385-442 */
int lt_fn_triangulate_line_with_direction(lt_ctx *ctx, const double seg1[4], const double cam1[11],
                                          const double seg2[4], const double cam2[11], const double direction[3],
                                          double out_line10[10]);
/* triangulate_line_with_one_point(l1, view1, l2, view2, point): functions.cc:325-383.  The reference's
 * solver (solvers/triangulation, a generated quartic + PoseLib's root finder) is restated from the
 * optimisation problem it solves, so this proposal agrees to rounding, not bit for bit. */
int lt_fn_triangulate_line_with_one_point(lt_ctx *ctx, const double seg1[4], const double cam1[11],
                                          const double seg2[4], const double cam2[11], const double point[3],
                                          double out_line10[10]);
int lt_fn_compute_fundamental_matrix(lt_ctx *ctx, const double cam1[11], const double cam2[11],
                                     double out[9]);
int lt_fn_compute_epipolar_IoU(lt_ctx *ctx, const double seg1[4], const double cam1[11],
                               const double seg2[4], const double cam2[11], double *out);
int lt_fn_triangulate_line(lt_ctx *ctx, const double seg1[4], const double cam1[11],
                           const double seg2[4], const double cam2[11], int by_endpoints,
                           double out_line10[10]);

/* merging::Aggregator::aggregate_line3d_list(lines, scores, num_outliers) (merging/aggregator.cc:53-101; takebest
 * :8-29 below four lines) as ComputeLineTracks and the track post-processing apply it (call sites
 * global_line_triangulator.cc:348, merging/merging_utils.cc:77, merging/merging.cc:509,635).  Host code of the tail: no
 * context, no device.  lines10 = n x line10 (uncertainty at [8]); out7 = start, end, uncertainty.  The orientation of
 * the result (which end is `start`) follows the sign rule documented in DESIGN.md section 5. */
int lt_fn_aggregate_line3d_list(int n, const double *lines10, const double *scores, int num_outliers, double out7[7]);

/* The host pass TriangulateImage / TriangulateAll make over one (image, neighbour) block of match rows -- the (n, 2)
 * int32 matrix `matches[ng_img_id]` of base_line_triangulator.cc:82-98 -- exposed for tests: out[r] = rows[r][0] |
 * rows[r][1] << 16 (the staged form), stats = {largest rows[:,0], largest rows[:,1] (as unsigned: a negative id wraps),
 * 1 if any rows[r][0] < rows[r-1][0]}.  The out-of-index error of :87-94 is raised from these maxima.  level: 0 = the
 * widest vector path the CPU has, 1 = scalar, 2 = AVX2, 3 = AVX-512 (a level the CPU lacks falls back to the widest).
 * No context, no device. */
int lt_fn_pack_match_rows(const int32_t *rows, int64_t n, uint32_t *out, uint32_t stats[3], int level);
/* the same pass into the COMPRESSED block form the rows cross PCIe in since round 4 (17 bits per row: the neighbour line
 * of every row + one "a new line starts here" bit; limap_amd/csrc/lt_rows.h), for blocks sorted by line id with steps of
 * 0 / +1 -- what limap's matchers write.  out: lt_fn_compressed_block_words(n) words, 8-byte aligned; stats[3] != 0: the
 * block is not of that shape (it is then staged in the plain form above). */
int64_t lt_fn_compressed_block_words(int64_t n);
int lt_fn_pack_match_rows_compressed(const int32_t *rows, int64_t n, uint32_t *out, uint32_t stats[4], int level);

#ifdef __cplusplus
}
#endif
#endif /* LIMAP_AMD_H */
