// lt_device.h -- types and launch wrappers shared by lt_kernels.hip and lt_api.cpp.
#pragma once

#include "lt_geom.h"

#include <hip/hip_runtime.h>
#include <stddef.h>

// error codes raised by device-side validation (mirrored in include/limap_amd.h)
#define LT_ERR_MATCH_LINE_RANGE 1
#define LT_ERR_MATCH_NGLINE_RANGE 2

namespace lt {

struct SceneChunk {  // images [img_begin, next chunk's img_begin) live in these buffers
  const double *k, *q, *t, *s;
  long long seg_begin;  // global index of the chunk's first segment
  int img_begin, pad_;
};
void launch_build_scene_chunked(hipStream_t st, int n_img, long long n_segs, int n_chunks, const SceneChunk *ch,
                                const long long *seg_off, double halfpix, Cam *cams, Seg *segs,
                                const int *img_list, int n_list, long long max_segs_per_img, void *gates);

void launch_build_cams(hipStream_t st, int n, const double *k, const double *q, const double *t, Cam *cams);
void launch_build_segs(hipStream_t st, long long n_segs, int n_img, const long long *seg_off,
                       const double *segs, double halfpix, const Cam *cams, Seg *out, void *gates);
void launch_build_pairs(hipStream_t st, int n_blk, const int *blk_img, const int *blk_nb, const Cam *cams,
                        PairRec *out, int *err_flag, unsigned long long *pair_counter,
                        unsigned long long *scan_status, int n_status, unsigned *blk_surv);
size_t sort_temp_bytes(long long P, int end_bit);
int launch_sort(hipStream_t st, void *temp, size_t temp_bytes, long long P, const unsigned *keys_in,
                unsigned *keys_out, const unsigned *vals_in, unsigned *vals_out, int end_bit);
void launch_node_offsets(hipStream_t st, long long P, long long G, const unsigned *skeys, long long *conn_off);
size_t scan_temp_bytes_u32_to_i64(long long n);
size_t scan_temp_bytes_popc(long long n);
int launch_scan_popc(hipStream_t st, void *temp, size_t temp_bytes, long long n, const unsigned long long *masks,
                     long long *out);
int launch_scan_u32_to_i64(hipStream_t st, void *temp, size_t temp_bytes, long long n, const unsigned *in,
                           long long *out);
void launch_gen_exhaustive(hipStream_t st, bool fill, long long n_items, const GenCfg &cfg,
                           const long long *item_off, long long G, const int *node_img, const long long *nb_off,
                           const int *blk_nb, const long long *seg_off, const Cam *cams, const Seg *segs,
                           const PairRec *pairs, unsigned long long *masks, const long long *mask_pos,
                           CRec *out_r, double *out_unc, const double *seg_vp, const unsigned char *seg_has_vp,
                           const int *blk_chunk_off, int max_nb, int max_chunks, const void *gates);
int ex_regions();
void launch_gates_exhaustive(hipStream_t st, int n_blk, int max_chunks, long long n_items, const GenCfg &cfg,
                             const long long *item_off, const int *blk_img, const int *blk_nb, const long long *seg_off,
                             const Cam *cams, const Seg *segs, const PairRec *pairs, unsigned long long *masks,
                             const int *blk_chunk_off, const void *gates, unsigned long long *ent_out,
                             unsigned long long *ctr, unsigned region_cap, int *err_flag);
void launch_tri_exhaustive(hipStream_t st, unsigned long long *ent, const unsigned long long *ctr,
                           unsigned region_cap, const GenCfg &cfg, long long n_items, const long long *item_off,
                           const int *blk_img, const int *blk_nb, const long long *nb_off, const long long *seg_off,
                           const Cam *cams, const Seg *segs, const PairRec *pairs, const int *blk_chunk_off,
                           unsigned long long *masks, CRec *st_r, double *st_unc, unsigned *st_node, float *st_z);
void launch_place_exhaustive(hipStream_t st, const unsigned long long *ctr, unsigned region_cap,
                             const unsigned long long *ent, const unsigned *st_node, const long long *item_off,
                             const int *blk_chunk_off, const unsigned long long *masks, const long long *mask_pos,
                             long long n_items, const long long *tri_off, long long G, unsigned *perm,
                             long long *fill_out);
void launch_fill_exhaustive(hipStream_t st, int n_blk, long long n_items, const GenCfg &cfg, const long long *item_off,
                            const int *blk_img, const int *blk_nb, const long long *nb_off, const long long *seg_off,
                            const Cam *cams, const Seg *segs, const PairRec *pairs, const unsigned long long *masks,
                            const long long *mask_pos, CRec *out_r, double *out_unc, const int *blk_chunk_off);
void launch_gen_exhaustive_pts(hipStream_t st, bool fill, long long n_items, const GenCfg &cfg,
                               const long long *item_off, long long G, const int *node_img, const long long *nb_off,
                               const int *blk_nb, const long long *seg_off, const Cam *cams, const Seg *segs,
                               const PairRec *pairs, unsigned short *cnt8, unsigned *item_cnt,
                               const long long *mask_pos, CRec *out_r, double *out_unc, const double *seg_vp,
                               const unsigned char *seg_has_vp, const long long *seg_pt_off, const void *seg_pts,
                               const double *sfm_xyz, int *err_flag, int many_on, int one_on,
                               const int *blk_chunk_off, int max_nb, int max_chunks, const void *gates);
void launch_popc(hipStream_t st, long long n, const unsigned long long *masks, unsigned *cnt, int n_masks);
void launch_tri_offsets_ex(hipStream_t st, long long G, const long long *item_off, const long long *mask_pos,
                           long long n_items, long long total, long long cap, long long *tri_off, int *err_flag);

void launch_select(hipStream_t st, long long G, const long long *tri_off, const double *score, double th,
                   int max_valid, long long *best_idx, unsigned *edge_flag, unsigned *n_valid, const CRec *cand,
                   const double *cand_unc, Cand *best_c, double *best_score, int *best_src2, int *n_tris,
                   bool wide, const int *err_flag, const unsigned long long *pair_counter, long long *result3,
                   const unsigned *perm);
void launch_edge_fill(hipStream_t st, long long G, const long long *tri_off, const unsigned *edge_flag,
                      const long long *edge_off, const CRec *cand, int *edges2, const unsigned *perm);

// device half of ComputeLineTracks (lt_kernels_tail.hip)
size_t tail_rec_bytes();
size_t tail_sort_temp_bytes(long long E, int end_bit);
void launch_tail_keys(hipStream_t st, long long G, const long long *tri_off, const unsigned *edge_flag,
                      const long long *edge_off, const CRec *cand, const long long *seg_off, int kb,
                      unsigned long long *keys, const unsigned *perm, int directed);
int launch_tail_sort(hipStream_t st, void *temp, size_t temp_bytes, long long E, const unsigned long long *keys_in,
                     unsigned long long *keys_out, int end_bit);
void launch_tail_sims(hipStream_t st, long long E, const unsigned long long *skeys, const int *n_tris, const Cand *best_c,
                      const LinkCfg3 &cfg, int kb, double *sims, unsigned *mark, unsigned *keep, const unsigned char *flags);
void launch_check_keys(hipStream_t st, long long n, const unsigned long long *keys, int kb, long long G, int *bad, int directed);
void launch_outer_pass_keys(hipStream_t st, long long E, const unsigned long long *keys, int kb, long long G, unsigned *counts,
                            int min_outer, unsigned char *flags, int *changed);
void launch_keys_undirect(hipStream_t st, long long E, unsigned long long *keys, int kb);
void launch_outer_filter(hipStream_t st, long long G, const long long *tri_off, const unsigned *edge_flag, const CRec *cand,
                         const long long *seg_off, const unsigned *perm, int min_outer, unsigned char *flags, int *changed);
void launch_tail_compact(hipStream_t st, long long E, const unsigned long long *skeys, const double *sims,
                         const unsigned *keep, const long long *kpos, void *out_pairs, long long *n_out, const long long *npos,
                         int kb);
void launch_tail_gather(hipStream_t st, long long G, const unsigned *mark, const long long *pos, const Cand *best_c,
                        const double *best_score, const int *best_src2, void *recs, int *nodes, long long *n_out);

void launch_track_connect(hipStream_t st, int T, const double *line7, const unsigned char *active, int all_active,
                          const LinkCfg3 &cfg, double cos_guard, unsigned long long *edges,
                          unsigned long long capacity, unsigned long long *n_edges);

}  // namespace lt
