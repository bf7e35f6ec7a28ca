// lt_host.h -- declarations shared by the host-side translation units of the C ABI (lt_api*.cpp): launch wrappers of
// lt_kernels_v2.hip, tracing ranges, configuration records, and the helpers one unit defines and another uses.
#pragma once

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

namespace lt {
void launch_fn_query(hipStream_t st, const double *in30, int by_endpoints, double *out32);
// lt_kernels_v2.hip
int gen_slots(long long max_rows);
int gen_groups(long long max_rows);
int gen_slots_ln(int max_own_segs);
int gen_max_run();
void launch_rows_ln(hipStream_t st, int n_blk, int n_slots, const void *desc, const unsigned *stream,
                    const long long *blk_line_base, unsigned *rstart, int *blk_nruns, unsigned *run_len,
                    unsigned *slot_row0, unsigned short *tr, int *ln_flag);
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
                      const long long *group_base, int phase, int ln_slots, const unsigned short *tr,
                      const unsigned *run_len, const unsigned *slot_row0, unsigned *blk_surv, const unsigned *blk_rnd0,
                      unsigned *round_count, const int *blk_vorder, unsigned *tri_unit_ctr);
size_t seg_point_bytes();
void launch_node_prefix(hipStream_t st, long long G, const int *node_img, const long long *seg_off,
                        const long long *nb_off, const long long *blk_line_base, unsigned *cnt_bl,
                        unsigned *base_bl, unsigned *n_tris, long long *tri_off, unsigned long long *status,
                        int *err_flag, void *node_rec);
void launch_expand_rows(hipStream_t st, int n_blk, const void *desc, const unsigned *stream, const unsigned *ovf, unsigned *rows);
void launch_place(hipStream_t st, int n_blk, long long max_rows, const long long *m_off, const int *blk_img,
                  const long long *seg_off, const long long *blk_line_base, const unsigned *base_bl,
                  const unsigned *wave_count, const long long *tri_off, const CRec *st_r, const double *st_unc,
                  const unsigned *st_key, CRec *cand, double *cand_unc, unsigned *cand_node, const long long *group_base,
                  unsigned *perm, const unsigned *blk_surv, const unsigned *blk_rnd0, const unsigned *round_count);
void launch_pack_keys(hipStream_t st, int n_blk, long long max_rows, const long long *m_off,
                      const unsigned *wave_count, const long long *wave_pos, const unsigned *st_key,
                      unsigned *keys_c, unsigned *src_c, const long long *group_base);
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
                   bool f32, unsigned *perm, void *rng, bool perm_is_placement, unsigned *bucket_cnt,
                   unsigned *bucket_list, unsigned bucket_cap, const unsigned *place, unsigned *rec, const float *st_z,
                   int *err_flag, void *sp_slots, int sp_slot_cap, unsigned *sp_cnt, unsigned *sp_ovf, void *sp_pairs,
                   void *sp_desc, long long sp_chunks, hipEvent_t ev_after, const void *node_rec, unsigned *pc_cnt,
                   void *pc_list, unsigned pc_cap, int one_kernel);
size_t score_split_chunk_bytes();
size_t score_split_entry_bytes();
long long score_split_chunks(long long C);
int score3_tile_buckets();
}

// ---- roctx ranges (tracing) around the host-visible stages (lt_api.cpp) ----
namespace lt_trace {
struct Range {
  bool on;
  explicit Range(const char *name);
  ~Range();
};
}  // namespace lt_trace
#define LT_CONCAT2(a, b) a##b
#define LT_CONCAT(a, b) LT_CONCAT2(a, b)
#define LT_RANGE(name) lt_trace::Range LT_CONCAT(lt_range_, __LINE__)(name)

namespace lt_impl {
// lt_api.cpp
// Test and developer switches (LT_TEST_*, LT_GEN_NO_LDS_TABLE, LT_GEN_ROW_SLOTS, LT_SCORE_FUSED, LT_SCORE_SPLIT,
// LT_TAIL_HOST: each selects the plain / reference form of a fast path, for the tests that compare the two) are read from
// the environment ONLY in a process that opted in with LT_ENABLE_TEST_SWITCHES=1 (tests/conftest.py, tools/ set it): a
// production process cannot be steered into the slow forms by a stray variable.  Returns getenv(name) or nullptr.
const char *test_switch(const char *name);
int fine_level();  // LT_FINE_TIMERS
bool fine_timers();
bool fine_gen_timers();
double now_ms();
double multiplier(double score_th);
lt::LinkCfg2 make_l2(const lt_config &c);
lt::LinkCfg3 make_l3(const lt_config &c);
lt::GenCfg make_gen(const lt_ctx *ctx);
lt::ScoreCfg make_score(const lt_ctx *ctx);
int bits_for(long long n);
void build_job_tables(lt_ctx *ctx);
int upload_points(lt_ctx *ctx);
// lt_api_run.cpp
int finish_run(lt_ctx *ctx);  // completes a run lt_run_device_async left in flight
void define_best_of_other_images(lt_ctx *ctx);
int materialize_compact(lt_ctx *ctx);

template <class T>
int upload_vec(lt_ctx *ctx, DevBuf &buf, const std::vector<T> &v) {
  ENSURE(ctx, buf, sizeof(T) * std::max<size_t>(v.size(), 1));
  if (!v.empty())
    HIPCHK(ctx, hipMemcpyAsync(buf.p, v.data(), sizeof(T) * v.size(), hipMemcpyHostToDevice, ctx->stream));
  return LT_OK;
}
}  // namespace lt_impl

#define LT_FINISH(ctx)                       \
  do {                                       \
    int rc_fin_ = lt_impl::finish_run(ctx);  \
    if (rc_fin_) return rc_fin_;             \
  } while (0)
