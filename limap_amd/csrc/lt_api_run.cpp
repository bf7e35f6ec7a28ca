// lt_api_run.cpp -- C ABI, part 3: job tables + remaining rows to HBM (lt_upload), the device run (lt_run_device_async:
// generation, placement, scoring, selection -- base_line_triangulator.cc:111-333, global_line_triangulator.cc:71-166),
// its completion (finish_run) and the read-back of the per-node results (lt_download).
#include "lt_host.h"

using namespace lt;
using namespace lt_impl;

extern "C" {

int lt_upload(lt_ctx *ctx) {
  LT_RANGE("lt_upload (match rows + job tables -> HBM)");
  LT_FINISH(ctx);
  if (!ctx->inited) return fail(ctx, LT_ERR_STATE, "upload before Init");
  if (ctx->uploaded) return LT_OK;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  double t0 = now_ms();
  build_job_tables(ctx);
  ctx->blk_vorder_ok = false;
  int rc;
  {
    // images referenced by the job (lt_refresh_scene_chunks rebuilds only their segment records)
    std::vector<char> need((size_t)std::max(ctx->n_img, 1), 0);
    for (size_t j = 0; j < ctx->job_imgs.size(); ++j) {
      need[(size_t)ctx->job_imgs[j]] = 1;
      for (int nb : ctx->job_nbs[j]) need[(size_t)nb] = 1;
    }
    std::vector<int> list;
    ctx->max_needed_segs = 0;
    for (int i = 0; i < ctx->n_img; ++i)
      if (need[(size_t)i]) {
        list.push_back(i);
        ctx->max_needed_segs = std::max(ctx->max_needed_segs, ctx->seg_off[i + 1] - ctx->seg_off[i]);
      }
    ctx->n_needed = (int)list.size();
    if (list.empty()) list.push_back(0);
    if ((rc = upload_vec(ctx, ctx->d_needed, list))) return rc;
  }
  if ((rc = upload_vec(ctx, ctx->d_nb_off, ctx->h_nb_off))) return rc;
  if ((rc = upload_vec(ctx, ctx->d_blk_img, ctx->h_blk_img))) return rc;
  if ((rc = upload_vec(ctx, ctx->d_blk_nb, ctx->h_blk_nb))) return rc;
  if ((rc = upload_vec(ctx, ctx->d_blk_slot, ctx->h_blk_slot))) return rc;
  if ((rc = upload_vec(ctx, ctx->d_blk_order, ctx->h_blk_order))) return rc;
  if ((rc = upload_vec(ctx, ctx->d_blk_line_base, ctx->h_blk_line_base))) return rc;
  if (ctx->job_mode == 1) {
    // block order in the tables is image-index-major; the staging arrays are call-order-major:
    // re-pack rows so that block b of the table owns rows m_off[b]..m_off[b+1]
    std::vector<long long> call_first_blk(ctx->job_imgs.size() + 1, 0);
    for (size_t j = 0; j < ctx->job_imgs.size(); ++j) call_first_blk[j + 1] = call_first_blk[j] + (long long)ctx->job_nbs[j].size();
    std::vector<int> job_pos(ctx->n_img, -1);
    for (size_t j = 0; j < ctx->job_imgs.size(); ++j) job_pos[ctx->job_imgs[j]] = (int)j;
    std::vector<long long> m_off(ctx->n_blk + 1, 0);
    // per device block: where its staged form lies (stream / overflow) and where its rows go
    std::vector<lt::RowDesc> desc((size_t)std::max(ctx->n_blk, 1));
    {
      long long b = 0;
      for (int i = 0; i < ctx->n_img; ++i) {
        int j = job_pos[i];
        if (j < 0) continue;
        for (size_t k = 0; k < ctx->job_nbs[j].size(); ++k, ++b) {
          long long cb = call_first_blk[j] + (long long)k;
          const long long n = ctx->h_m_off[cb + 1] - ctx->h_m_off[cb];
          m_off[b + 1] = m_off[b] + n;
          desc[(size_t)b] = lt::RowDesc{ctx->h_c_off[(size_t)cb], ctx->h_ovf_off[(size_t)cb], m_off[b], (int)n, ctx->h_line0[(size_t)cb]};
        }
      }
    }
    ctx->P = m_off[ctx->n_blk];
    ctx->n_conn = ctx->P;
    ctx->max_rows = 0;
    for (int bq = 0; bq < ctx->n_blk; ++bq) ctx->max_rows = std::max(ctx->max_rows, m_off[bq + 1] - m_off[bq]);
    if (ctx->P >= (1ll << 32) - 1) return fail(ctx, LT_ERR_ARGUMENT, "too many match rows in one batch (>= 2^32-1)");
    ENSURE(ctx, ctx->d_m_pairs, sizeof(int) * (size_t)std::max<long long>(ctx->P, 1));
    // the stream travels in call order, whatever part of it was not sent while the calls were still buffering; the
    // blocks are expanded on the device into device block order (so out-of-order calls need no re-packing any more)
    const size_t total_w = ctx->h_m_pairs.size();
    if (sizeof(int) * std::max<size_t>(total_w, 1) > ctx->d_c_stream.cap) {
      HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
      ctx->streamed_ints = 0;  // the buffer is replaced: everything is sent again
      ENSURE(ctx, ctx->d_c_stream, sizeof(int) * std::max<size_t>(total_w, 1));
    }
    {
      const size_t sent = std::min(ctx->streamed_ints, total_w);
      if (total_w > sent)
        HIPCHK(ctx, hipMemcpyAsync(ctx->d_c_stream.as<int>() + sent, ctx->h_m_pairs.data() + sent, sizeof(int) * (total_w - sent),
                                   hipMemcpyHostToDevice, ctx->stream));
      ctx->streamed_ints = total_w;
    }
    if ((rc = upload_vec(ctx, ctx->d_ovf, ctx->h_ovf))) return rc;
    if ((rc = upload_vec(ctx, ctx->d_rowdesc, desc))) return rc;
    // Line-slot form (k_gates_ln): every block compressed (sorted, lines contiguous), neighbour tables within the LDS.
    // Whether a run of equal line ids is longer than the kernel's outcome bits is only known on the device: the form is
    // built optimistically and its flag read with the sync at the end of the upload.
    // (extra proposals -- VP, points -- yield a variable number of candidates per row: they keep the row-slot form)
    const bool extras_cfg = (ctx->cfg.use_vp && !ctx->cfg.disable_vp_triangulation) ||
                            (ctx->pts_ready && (!ctx->cfg.disable_many_points_triangulation || !ctx->cfg.disable_one_point_triangulation));
    bool try_ln = ctx->rows_sorted && ctx->n_blk > 0 && ctx->P > 0 && ctx->max_nb_segs <= 1024 && !extras_cfg &&
                  !test_switch("LT_GEN_ROW_SLOTS") && !test_switch("LT_GEN_NO_LDS_TABLE");
    for (size_t cb = 0; try_ln && cb < ctx->h_ovf_off.size(); ++cb)
      if (ctx->h_ovf_off[cb] >= 0) try_ln = false;
    ctx->rows_ln = false;
    ctx->ln_slots = 0;
    if (try_ln) {
      ctx->ln_slots = gen_slots_ln(ctx->max_own_segs);
      const size_t n_entries = (size_t)std::max<long long>(ctx->h_blk_line_base[ctx->n_blk], 1);
      ENSURE(ctx, ctx->d_run_len, 4 * n_entries);
      ENSURE(ctx, ctx->d_base_bl, 4 * n_entries);  // scratch here: the run starts (k_node_prefix rewrites it in every run)
      ENSURE(ctx, ctx->d_slot_row0, 4 * (size_t)ctx->n_blk * (size_t)ctx->ln_slots);
      ENSURE(ctx, ctx->d_blk_nruns, 4 * (size_t)ctx->n_blk);
      ENSURE(ctx, ctx->d_ln_flag, 4);
      HIPCHK(ctx, hipMemsetAsync(ctx->d_ln_flag.p, 0, 4, ctx->stream));
      launch_rows_ln(ctx->stream, ctx->n_blk, ctx->ln_slots, ctx->d_rowdesc.p, ctx->d_c_stream.as<unsigned>(),
                     ctx->d_blk_line_base.as<long long>(), ctx->d_base_bl.as<unsigned>(), ctx->d_blk_nruns.as<int>(),
                     ctx->d_run_len.as<unsigned>(), ctx->d_slot_row0.as<unsigned>(), ctx->d_m_pairs.as<unsigned short>(),
                     ctx->d_ln_flag.as<int>());
      // a slot of round counts for every round a block could have
      std::vector<unsigned> rnd0((size_t)ctx->n_blk + 1, 0u);
      for (int bq = 0; bq < ctx->n_blk; ++bq) rnd0[(size_t)bq + 1] = rnd0[(size_t)bq] + (unsigned)((m_off[bq + 1] - m_off[bq] + 63) / 64);
      ctx->n_round_slots = (long long)rnd0[(size_t)ctx->n_blk];
      if ((rc = upload_vec(ctx, ctx->d_blk_rnd0, rnd0))) return rc;
      int flag = 0;
      HIPCHK(ctx, hipMemcpyAsync(&flag, ctx->d_ln_flag.p, 4, hipMemcpyDeviceToHost, ctx->stream));
      HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
      ctx->rows_ln = flag == 0;
    }
    if (!ctx->rows_ln)
      launch_expand_rows(ctx->stream, ctx->n_blk, ctx->d_rowdesc.p, ctx->d_c_stream.as<unsigned>(), ctx->d_ovf.as<unsigned>(),
                         ctx->d_m_pairs.as<unsigned>());
    if ((rc = upload_vec(ctx, ctx->d_m_off, m_off))) return rc;
    // per-block records of the matched pipeline (row range, images, segment bases): a function of the job
    {  // k_gates_ln takes the blocks in (neighbour, image) order: see the kernel
      std::vector<int> vo((size_t)std::max(ctx->n_blk, 1), 0);
      for (int b = 0; b < ctx->n_blk; ++b) vo[(size_t)b] = b;
      std::stable_sort(vo.begin(), vo.begin() + ctx->n_blk, [&](int x, int y) {
        return ctx->h_blk_nb[(size_t)x] != ctx->h_blk_nb[(size_t)y] ? ctx->h_blk_nb[(size_t)x] < ctx->h_blk_nb[(size_t)y]
                                                                    : ctx->h_blk_img[(size_t)x] < ctx->h_blk_img[(size_t)y];
      });
      if ((rc = upload_vec(ctx, ctx->d_blk_vorder, vo))) return rc;
      ctx->blk_vorder_ok = true;
    }
    ENSURE(ctx, ctx->d_blkrec, blk_rec_bytes() * (size_t)std::max(ctx->n_blk, 1));
    launch_build_blk(ctx->stream, ctx->n_blk, ctx->d_m_off.as<long long>(), ctx->d_blk_img.as<int>(),
                     ctx->d_blk_nb.as<int>(), ctx->d_blk_slot.as<int>(), ctx->d_seg_off.as<long long>(),
                     ctx->d_blk_line_base.as<long long>(), ctx->d_blkrec.p);
  } else if (ctx->job_mode == 2) {
    // work items: per node, per neighbour block, chunks of 64 neighbour lines
    ctx->h_item_off.assign(ctx->G + 1, 0);
    // per block: chunks of the earlier neighbour blocks of the same image (the item index of
    // (node, block, chunk) is item_off[node] + blk_chunk_off[block] + chunk -- no search on the device)
    std::vector<int> blk_chunk_off((size_t)std::max(ctx->n_blk, 1), 0);
    ctx->max_chunks = 1;
    long long items = 0, conns = 0;
    for (int i = 0; i < ctx->n_img; ++i) {
      long long per_node = 0, conn_node = 0;
      for (long long b = ctx->h_nb_off[i]; b < ctx->h_nb_off[i + 1]; ++b) {
        int i2 = ctx->h_blk_nb[b];
        long long M2 = ctx->seg_off[i2 + 1] - ctx->seg_off[i2];
        blk_chunk_off[(size_t)b] = (int)per_node;
        ctx->max_chunks = std::max(ctx->max_chunks, (int)((M2 + 63) / 64));
        per_node += (M2 + 63) / 64;
        conn_node += M2;
      }
      for (long long g = ctx->seg_off[i]; g < ctx->seg_off[i + 1]; ++g) {
        ctx->h_item_off[g] = items;
        items += per_node;
        conns += conn_node;
      }
    }
    ctx->h_item_off[ctx->G] = items;
    ctx->P = items;
    ctx->n_conn = conns;
    if ((rc = upload_vec(ctx, ctx->d_item_off, ctx->h_item_off))) return rc;
    if ((rc = upload_vec(ctx, ctx->d_blk_chunk_off, blk_chunk_off))) return rc;
  } else {
    ctx->P = 0;
    ctx->n_conn = 0;
  }
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  ctx->uploaded = true;
  ctx->ran = false;
  ctx->timers[8] = now_ms() - t0;
  return LT_OK;
}

// Completes the run that lt_run_device_async left in flight: waits for its end marker, reads the error flag,
// the candidate count and the pair statistic from the pinned slots of its set, and its event timings.
extern "C++" {
namespace lt_impl {
int finish_run(lt_ctx *ctx) {
  LT_RANGE("lt_sync (end of run: result scalars, event timings)");
  if (!ctx->run_pending) return LT_OK;
  ctx->run_pending = false;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  hipEvent_t *ev = ctx->pend_set ? ctx->ev_b : ctx->ev;
  long long *hp = ctx->h_pinned ? ctx->h_pinned + 8 * ctx->pend_set : nullptr;
  int derr = 0;
  if (hp) {
    HIPCHK(ctx, hipEventSynchronize(ctx->pend_ev_end));
    derr = (int)hp[1];
    ctx->stat_pairs_eval = hp[2];
    ctx->C_last = ctx->pend_count_on_device ? hp[0] : ctx->pend_C;
  } else {
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    HIPCHK(ctx, hipMemcpy(&derr, ctx->d_err.p, sizeof(int), hipMemcpyDeviceToHost));
    unsigned long long pe = 0;
    ctx->C_last = ctx->pend_C;
    if (ctx->pend_count_on_device)
      HIPCHK(ctx, hipMemcpy(&ctx->C_last, ctx->d_tri_off.as<long long>() + ctx->G, 8, hipMemcpyDeviceToHost));
    if (ctx->C_last > 0) HIPCHK(ctx, hipMemcpy(&pe, ctx->d_pair_counter.p, 8, hipMemcpyDeviceToHost));
    ctx->stat_pairs_eval = (long long)pe;
  }
  ctx->stat_survivors = -1;  // summed on demand (lt_get_timers)
  if (derr == 5) {
    // the staging capacity of the one-pass exhaustive mode did not hold: repeat the job in the two-pass form (exact
    // sizes).  When this is the earlier of two runs in flight the later one -- same inputs -- is repeated at its own end.
    if (ctx->ex_retry_depth > 0 || !ctx->ex_staged_set[ctx->pend_set])
      return fail(ctx, LT_ERR_RUNTIME, "internal: candidate staging overflow outside the one-pass exhaustive mode");
    ctx->ex_two_pass = true;
    // the counters kept counting beyond the capacity: the next run gets what this one would have needed
    if (hp && hp[3] > 0 && ctx->n_conn > 0)
      ctx->ex_frac = 1.4 * (double)hp[3] * (double)ex_regions() / (double)ctx->n_conn;
    if (ctx->in_run_async) return LT_OK;
    ctx->ex_retry_depth = 1;
    int rc2 = lt_run_device_async(ctx);
    if (!rc2) rc2 = finish_run(ctx);
    ctx->ex_retry_depth = 0;
    return rc2;
  }
  if (derr == 8) {
    // k_score_q found workgroups of one queue on different XCDs, or a wait for a swept tile timed out (the GPU shared with
    // another persistent kernel): this context scores in the two-kernel form from now on, the run is repeated
    if (ctx->score_two_kernels)
      return fail(ctx, LT_ERR_RUNTIME, "internal: device flag 8 outside the one-kernel scoring form");
    ctx->score_two_kernels = true;
    if (ctx->in_run_async) return LT_OK;
    const int depth = ctx->ex_retry_depth;
    ctx->ex_retry_depth = depth + 1;
    int rc2 = lt_run_device_async(ctx);
    if (!rc2) rc2 = finish_run(ctx);
    ctx->ex_retry_depth = depth;
    return rc2;
  }
  if (derr == 7) {
    // the chunk store of the split scoring form did not hold the pairs that passed the sweep: this context scores with the
    // fused kernel from now on (no store), the run is repeated
    if (ctx->ex_retry_depth > 1)
      return fail(ctx, LT_ERR_RUNTIME, "internal: pair chunk overflow outside the split scoring form");
    ctx->score_fused = true;
    if (ctx->in_run_async) return LT_OK;
    const int depth = ctx->ex_retry_depth;
    ctx->ex_retry_depth = depth + 1;
    int rc2 = lt_run_device_async(ctx);
    if (!rc2) rc2 = finish_run(ctx);
    ctx->ex_retry_depth = depth;
    return rc2;
  }
  ctx->timers[17] = ctx->timers[18] = 0.0;
  if (hp && ctx->ex_staged_set[ctx->pend_set]) {
    ctx->timers[17] = (double)hp[3] * (double)ex_regions();
    ctx->timers[18] = (double)ctx->ex_region_cap * (double)ex_regions();
  }
  if (ctx->job_mode == 2 && derr == 0 && ctx->n_conn > 0) {
    // this run's need of staging slots -> capacity of the next one: 1.4 x the fullest region (one-pass form), or an
    // estimate from the candidate count (two-pass form: slots = listed connections + block padding, ~1.5 per candidate)
    if (hp && ctx->ex_staged_set[ctx->pend_set])
      ctx->ex_frac = std::max(1.4 * (double)hp[3] * (double)ex_regions() / (double)ctx->n_conn, 1e-4);
    else if (ctx->ex_frac <= 0.0)
      ctx->ex_frac = std::max(2.2 * (double)ctx->C_last / (double)ctx->n_conn, 1e-4);
    ctx->ex_two_pass = false;
  }
  if (derr == 4) return fail(ctx, LT_ERR_RUNTIME, "internal: the scan of the node counts did not complete");
  if (derr == 3)
    return fail(ctx, LT_ERR_RUNTIME, "more than 65 000 candidates on one connection of the exhaustive mode (16-bit counts)");
  if (derr == 2)
    return fail(ctx, LT_ERR_RUNTIME, "map::at: a point shared by two lines has a point3D_id that is not among the SfM points");
  if (derr != 0) return fail(ctx, LT_ERR_RUNTIME, "IndexError! Out-of-index matches detected on the device");
  // coarse stages from five events (every hipEventRecord between kernels costs ~1.5 us of device time):
  // [3] generation incl. the pair records = ev0..ev3, [4] placement = ev3..ev4, [5] scoring incl. its per-candidate
  // records = ev4..ev5, [6] selection = ev5..ev7; [1], [2], [7] are no longer separate stages
  float ms;
  ctx->timers[1] = ctx->timers[2] = ctx->timers[7] = 0.0;
  const int eg = ctx->pend_ev_gen_end, ep = ctx->pend_ev_place_end;
  // start of the run: its own event, or the end marker of the run it was enqueued behind (lt_run_device_async); end of
  // the run: the end marker behind the result record, if there are result slots
  const hipEvent_t e_start = ctx->pend_ev_start ? ctx->pend_ev_start : ev[0];
  const hipEvent_t e_end = hp ? ctx->pend_ev_end : ev[7];
  HIPCHK(ctx, hipEventElapsedTime(&ms, e_start, e_end));
  ctx->timers[0] = ms;
  const bool sampled = ctx->pend_sampled;  // this run carried the stage events (lt_ctx.h); otherwise [3]-[6], [13]-[15] keep
                                           // the values of the last run that did
  if (sampled) {
  const hipEvent_t kA[4] = {e_start, ev[eg], ev[ep], ev[5]}, kB[4] = {ev[eg], ev[ep], ev[5], e_end};
  const int kT[4] = {3, 4, 5, 6};
  for (int k = 0; k < 4; ++k) {
    HIPCHK(ctx, hipEventElapsedTime(&ms, kA[k], kB[k]));
    ctx->timers[kT[k]] = ms;
  }
  // single-kernel durations of the matched pipeline: [13] k_gates, [14] k_tri_rows, [15] k_score3
  ctx->timers[13] = ctx->timers[14] = ctx->timers[15] = 0.0;
  if (ctx->pend_fine_gen && ctx->job_mode == 1 && ctx->n_blk > 0 && ctx->max_rows > 0) {
    if (hipEventElapsedTime(&ms, ev[8], ev[9]) == hipSuccess) ctx->timers[13] = ms;
    if (hipEventElapsedTime(&ms, ev[9], ev[10]) == hipSuccess) ctx->timers[14] = ms;
  }
  if (ctx->pend_fine_score && ctx->C_last > 0 && hipEventElapsedTime(&ms, ev[11], ev[5]) == hipSuccess) ctx->timers[15] = ms;
  (void)hipGetLastError();
  ++ctx->timer_stage_runs;
  }
  ctx->timers[11] = (double)ctx->stat_pairs_eval;
  for (int k = 0; k < 24; ++k) {  // [16] (survivors) is counted on demand by lt_get_timers, not per run
    const bool stage = (k >= 3 && k <= 6) || (k >= 13 && k <= 15);
    if (k != 8 && k != 9 && k != 10 && k != 12 && k != 16 && (sampled || !stage)) ctx->timer_sums[k] += ctx->timers[k];
  }
  ++ctx->timer_runs;
  return LT_OK;
}
}  // namespace lt_impl
}  // extern "C++"
int lt_sync(lt_ctx *ctx) { return finish_run(ctx); }

// Enqueues the whole run and returns.  A run still in flight from the previous call is completed AFTER the
// new one has been enqueued (its errors are the return value), so a caller that streams batches keeps the
// device busy across the host's end-of-run bookkeeping.  Two sets of events / pinned result slots alternate.
int lt_run_device_async(lt_ctx *ctx) {
  LT_RANGE("lt_run_device (enqueue: generation, placement, scoring, selection)");
  if (!ctx->uploaded) return fail(ctx, LT_ERR_STATE, "lt_run_device before lt_upload");
  if (!ctx->h_pinned) LT_FINISH(ctx);  // no pinned result slots: nothing may stay in flight
  ctx->shard_keys = ctx->shard_own_keys = -1;  // a new run: nothing of an earlier merge is pending
  // host copies of the job's images from an earlier read-back are superseded by this run: forget them, so that a query
  // after the device form of the tail (which refreshes the graph's nodes only) cannot see what an earlier configuration
  // left behind
  if (ctx->inited && !ctx->best_c_set.empty())
    for (int idx : ctx->job_imgs)
      if (ctx->best_c_set[(size_t)idx] == 1) {
        ctx->best_c_set[(size_t)idx] = 0;
        for (long long g = ctx->seg_off[idx]; g < ctx->seg_off[idx + 1]; ++g) {
          if ((size_t)g < ctx->has_best.size()) ctx->has_best[(size_t)g] = 0;
          if ((size_t)g < ctx->valid_edges.cnt.size()) ctx->valid_edges.cnt[(size_t)g] = 0;
        }
      }
  const int set = ctx->run_pending ? (ctx->pend_set ^ 1) : 0;
  if (ctx->h_pinned) std::memset(ctx->h_pinned + 8 * set, 0, 32);  // this set's result slots (its previous run is finished)
  if (set == 1 && !ctx->ev_b[0])
    for (auto &e : ctx->ev_b) HIPCHK(ctx, hipEventCreate(&e));
  hipEvent_t *ev = set ? ctx->ev_b : ctx->ev;
  long long *hp = ctx->h_pinned ? ctx->h_pinned + 8 * set : nullptr;
  ctx->ex_staged_set[set] = false;
  // stage events: every run that starts on an idle context, every LT_TIMER_SAMPLE-th of the runs enqueued behind one in
  // flight (lt_ctx.h)
  int sample_n = 8;
  if (const char *e = getenv("LT_TIMER_SAMPLE")) sample_n = std::max(1, atoi(e));
  if (!(ctx->run_pending && hp)) ctx->async_seq = 0;
  const bool sampled = (ctx->async_seq++ % (unsigned)sample_n) == 0u;
  const bool fine_gen = sampled && fine_gen_timers(), fine_score = sampled && fine_timers();
  HIPCHK(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  const long long G = ctx->G, P = ctx->P;
  {
    int rcp = upload_points(ctx);
    if (rcp) return rcp;
  }
  GenCfg gcfg = make_gen(ctx);
  // like the VP proposals, the point-guided ones do not depend on the algebraic gates
  if (ctx->pts_ready && (!ctx->cfg.disable_many_points_triangulation || !ctx->cfg.disable_one_point_triangulation))
    gcfg.force_undecided = 1;
  const ScoreCfg scfg = make_score(ctx);
  // LT_TEST_SCORE_F64: the sweep's early exit in double precision (the default is the bounded single-precision form)
  const bool score_f32 = !test_switch("LT_TEST_SCORE_F64");
  // conservative square of the scale-invariant endpoint gate (see k_score3)
  const bool guards_on = scfg.l3.th_scaleinv > 0.0 && scfg.l3.score_th > 0.0 && scfg.l3.score_th < 1.0 &&
                         !test_switch("LT_TEST_NO_SCORE_GUARDS");
  const double guard2 = guards_on ? (scfg.l3.th_scaleinv * (1.0 + 1e-6)) * (scfg.l3.th_scaleinv * (1.0 + 1e-6)) : 1e300;
  ENSURE(ctx, ctx->d_err, sizeof(int));
  ENSURE(ctx, ctx->d_pair_counter, 8);
  ENSURE(ctx, ctx->d_result3, 32);
  ENSURE(ctx, ctx->d_pairs, sizeof(PairRec) * (size_t)std::max(ctx->n_blk, 1));
  ENSURE(ctx, ctx->d_tri_off, sizeof(long long) * (size_t)(G + 1));
  // Enqueued behind a run still in flight: that run's end marker is this run's start event.  (Every other entry point
  // that puts work on the stream completes the pending run first -- LT_FINISH --, so nothing sits between the two.)
  hipEvent_t ev_start = ev[0];
  if (ctx->run_pending && hp && ctx->pend_ev_end && !test_switch("LT_TEST_OWN_START_EVENT")) ev_start = ctx->pend_ev_end;
  else HIPCHK(ctx, hipEventRecord(ev[0], st));
  // also zeroes the error flag, the pair statistic and the look-back state of k_node_prefix's scan
  // (+ the tile cost-class counters of k_cand_meta / k_score3 behind the scan's words: zeroed by the same kernel)
  const int n_status_scan = (int)((G + 1 + 255) / 256) + 1;
  // + the staging counters of the one-pass exhaustive mode; all counters 128 bytes apart
  const int n_status = n_status_scan + score3_tile_buckets() * 16 + ex_regions() * 16 + 8 * 16;  // ... + k_tri_rounds' unit counters
  ENSURE(ctx, ctx->d_scan_status, 8 * (size_t)n_status);
  const bool ln_job = ctx->job_mode == 1 && ctx->rows_ln && ctx->rows_sorted;
  if (ln_job) ENSURE(ctx, ctx->d_blk_surv, 4 * (size_t)std::max(ctx->n_blk, 1));
  launch_build_pairs(st, ctx->n_blk, ctx->d_blk_img.as<int>(), ctx->d_blk_nb.as<int>(), ctx->d_cams.as<Cam>(),
                     ctx->d_pairs.as<PairRec>(), ctx->d_err.as<int>(),
                     ctx->d_pair_counter.as<unsigned long long>(), ctx->d_scan_status.as<unsigned long long>(),
                     n_status, ln_job ? ctx->d_blk_surv.as<unsigned>() : nullptr);

  bool node_rec_valid = false;  // k_node_prefix of this run wrote the per-node scoring records (d_node_rec)
  long long C_known = -1;  // candidate count once it is known on the host
  long long C_bound = 0;   // what sizes the compact arrays: the count, or an upper bound while it stays on the device
  int ev_gen_end = 3, ev_place_end = 4;  // events that close the generation / placement stage (see finish_run)
  long long C_run = 0;  // what finish_run reports as the run's candidate count unless the device copy does
  if (ctx->job_mode == 1) {
    const size_t Pn = (size_t)std::max<long long>(P, 1);
    const bool fast = ctx->rows_sorted;
    {
      // the line-slot form carries no extra proposals: if they were switched on after the upload, the rows go back to
      // the row-slot form now
      const bool vp_now = ctx->cfg.use_vp && !ctx->cfg.disable_vp_triangulation;
      const bool pts_now = ctx->pts_ready && (!ctx->cfg.disable_many_points_triangulation || !ctx->cfg.disable_one_point_triangulation);
      if (ctx->rows_ln && (vp_now || pts_now)) {
        launch_expand_rows(st, ctx->n_blk, ctx->d_rowdesc.p, ctx->d_c_stream.as<unsigned>(), ctx->d_ovf.as<unsigned>(),
                           ctx->d_m_pairs.as<unsigned>());
        ctx->rows_ln = false;
      }
    }
    const bool ln = ctx->rows_ln && fast;
    const int ln_slots = ln ? ctx->ln_slots : 0;
    if (ln) ENSURE(ctx, ctx->d_round_count, 4 * (size_t)std::max<long long>(ctx->n_round_slots, 1));
    const long long n_waves = (long long)ctx->n_blk * gen_groups(ctx->max_rows);  // candidate lists (row-slot form)
    const long long n_slots_all = (long long)ctx->n_blk * (ln ? ln_slots : gen_slots(ctx->max_rows));  // survivor lists
    const long long n_entries = ctx->h_blk_line_base[ctx->n_blk];
    // ---- generation in row order; valid candidates appended in row order to per-wave lists ----
    // VP-guided proposals: up to three candidates per match row (vp of l1, vp of l2, algebraic)
    const bool vp_on = ctx->cfg.use_vp && !ctx->cfg.disable_vp_triangulation;
    if (vp_on && !ctx->vp_ready) return fail(ctx, LT_ERR_STATE, "use_vp is set but InitVPResults was not called");
    // point-guided proposals (SetBipartites2d): the many-points line fit (base_line_triangulator.cc:183-236) and the
    // one-point proposal (:238-248, one candidate per shared point; lt_devfn.h: one_point_candidate)
    const bool pts_any = ctx->pts_ready && (!ctx->cfg.disable_many_points_triangulation || !ctx->cfg.disable_one_point_triangulation);
    const bool many_on = pts_any && !ctx->cfg.disable_many_points_triangulation;
    const bool one_on = pts_any && !ctx->cfg.disable_one_point_triangulation;
    const bool pts_on = pts_any;
    // Without extra proposals a match row yields at most one candidate: the staging has one slot per row.  With them a
    // connection yields a variable number (many-points, one per shared point -- no limit, as in the reference --, vp(l1),
    // vp(l2), algebraic): stage B runs twice, counting then storing, and the staging gets the exact size in between.
    const bool extras = vp_on || pts_on;
    long long staged_total = -1;  // extras: candidates of the batch, known on the host after the counting run
    if (!extras) {
      ENSURE(ctx, ctx->d_st_c, sizeof(CRec) * Pn); ENSURE(ctx, ctx->d_st_l, sizeof(double) * Pn);
      ENSURE(ctx, ctx->d_st_key, 4 * (Pn + 64));  // (+ 64: k_place_rounds reads whole rounds)
    }
    ENSURE(ctx, ctx->d_wave_count, 4 * (size_t)(n_waves + 1));
    if (extras) ENSURE(ctx, ctx->d_wave_pos, 8 * (size_t)(n_waves + 1));
    const long long *group_base = extras ? ctx->d_wave_pos.as<long long>() : nullptr;
    ENSURE(ctx, ctx->d_ntris_u, 4 * (size_t)(G + 1));
    if (fast) {
      // per-(block, line) counters: k_node_prefix zeroes every counter it reads, so the array only has to
      // be cleared when it is new or when the previous run did not get that far
      const size_t nb = 4 * (size_t)std::max<long long>(n_entries, 1);
      const void *before = ctx->d_cnt_bl.p;
      ENSURE(ctx, ctx->d_cnt_bl, nb);
      ENSURE(ctx, ctx->d_base_bl, nb);
      if (ctx->d_cnt_bl.p != before || !ctx->cnt_bl_clean || ctx->cnt_bl_bytes != nb) {
        HIPCHK(ctx, hipMemsetAsync(ctx->d_cnt_bl.p, 0, ctx->d_cnt_bl.cap, st));
        ctx->cnt_bl_bytes = nb;
      }
      ctx->cnt_bl_clean = false;
    }
    const bool no_lds_table = test_switch("LT_GEN_NO_LDS_TABLE") != nullptr;  // developer / test switch
    // LDS tables of k_gates: the neighbour's gate records (T2) and the image's own segments (T1), 80 B
    // per segment each; two workgroups per CU need both within 80 KB, one workgroup within 160 KB
    int lds_segs = (!no_lds_table && ctx->max_nb_segs <= 1024) ? ctx->max_nb_segs : 0;
    int lds_segs1 = (!no_lds_table && ctx->max_own_segs <= 1024) ? ctx->max_own_segs : 0;
    // both tables only while two workgroups still fit a CU (80 KB each): beyond that the own segments come
    // from L2 -- measured at 700 / 1000 segments per image: k_gates -16 % / -14 % against one workgroup per CU
    if (lds_segs + lds_segs1 > 1024) lds_segs1 = 0;
    if (ln) {  // the line-slot form keeps no table of the image's own segments
      lds_segs = std::max(ctx->max_nb_segs, 1);
      lds_segs1 = 0;
    }
    {
      ENSURE(ctx, ctx->d_st_row, 8 * Pn);
      ENSURE(ctx, ctx->d_surv_count, 4 * (size_t)(n_slots_all + 1));
      if (!ctx->d_seg_gates.p) return fail(ctx, LT_ERR_STATE, "segment gate records missing (Init not run?)");
      auto gen = [&](int phase) {
      launch_gen_split(st, ctx->n_blk, ctx->max_rows, gcfg, ctx->d_m_off.as<long long>(), ctx->d_m_pairs.as<int>(),
                       ctx->d_blk_img.as<int>(), ctx->d_blk_nb.as<int>(), ctx->d_blk_slot.as<int>(),
                       ctx->d_seg_off.as<long long>(), ctx->d_cams.as<Cam>(), ctx->d_segs.as<Seg>(),
                       ctx->d_pairs.as<PairRec>(), ctx->d_blk_line_base.as<long long>(), ctx->d_st_c.as<CRec>(),
                       ctx->d_st_l.as<double>(), ctx->d_st_key.as<unsigned>(), ctx->d_wave_count.as<unsigned>(),
                       fast ? ctx->d_cnt_bl.as<unsigned>() : nullptr, lds_segs, lds_segs1, ctx->d_st_row.p,
                       ctx->d_surv_count.as<unsigned>(), G, ctx->d_seg_gates.p, ctx->d_blkrec.p, fine_gen ? &ev[8] : nullptr,
                       vp_on ? ctx->d_seg_vp.as<double>() : nullptr,
                       vp_on ? ctx->d_seg_has_vp.as<unsigned char>() : nullptr,
                       pts_on ? ctx->d_seg_pt_off.as<long long>() : nullptr, pts_on ? ctx->d_seg_pts.p : nullptr,
                       (pts_on && ctx->sfm_given) ? ctx->d_sfm_xyz.as<double>() : nullptr, ctx->d_err.as<int>(),
                       many_on ? 1 : 0, one_on ? 1 : 0, group_base, phase, ln_slots, ctx->d_m_pairs.as<unsigned short>(),
                       ctx->d_run_len.as<unsigned>(), ctx->d_slot_row0.as<unsigned>(), ctx->d_blk_surv.as<unsigned>(),
                       ctx->d_blk_rnd0.as<unsigned>(), ctx->d_round_count.as<unsigned>(),
                       (ctx->blk_vorder_ok && !test_switch("LT_TEST_GATES_IMAGE_MAJOR")) ? ctx->d_blk_vorder.as<int>() : nullptr,
                       test_switch("LT_TEST_TRI_STATIC") ? nullptr
                           : (unsigned *)(ctx->d_scan_status.as<unsigned long long>() + n_status_scan + score3_tile_buckets() * 16 + ex_regions() * 16));
      };
      if (!extras) {
        gen(0);
      } else {
        gen(1);  // k_gates + the counting run of stage B
        HIPCHK(ctx, hipMemsetAsync(ctx->d_wave_count.as<unsigned>() + n_waves, 0, 4, st));
        size_t tmp = scan_temp_bytes_u32_to_i64(n_waves + 1);
        ENSURE(ctx, ctx->d_scan_tmp, std::max<size_t>(tmp, 16));
        if (launch_scan_u32_to_i64(st, ctx->d_scan_tmp.p, tmp, n_waves + 1, ctx->d_wave_count.as<unsigned>(),
                                   ctx->d_wave_pos.as<long long>()) != 0)
          return fail(ctx, LT_ERR_HIP, "rocprim scan failed");
        HIPCHK(ctx, hipMemcpyAsync(&staged_total, ctx->d_wave_pos.as<long long>() + n_waves, 8, hipMemcpyDeviceToHost, st));
        HIPCHK(ctx, hipStreamSynchronize(st));
        if (staged_total >= (1ll << 32) - 1)
          return fail(ctx, LT_ERR_ARGUMENT, "too many candidates in one batch (>= 2^32-1): triangulate the images in smaller batches");
        const size_t Sn = (size_t)std::max<long long>(staged_total, 1);
        ENSURE(ctx, ctx->d_st_c, sizeof(CRec) * Sn); ENSURE(ctx, ctx->d_st_l, sizeof(double) * Sn);
        ENSURE(ctx, ctx->d_st_key, 4 * Sn);
        gen(2);  // the storing run, lists back to back
      }
    }
    // with the per-kernel events on, the one after k_tri_rows also ends the generation stage; without them NO event is
    // recorded here (an event between two kernels opens a 5-6 us gap in the stream; round 4): the generation stage's
    // timer then runs to the end of the placement and the placement's reads 0 -- bench.py prices both in its extra
    // steps with LT_FINE_TIMERS=2
    bool gen_event_pending = false;
    if (fine_gen && ctx->n_blk > 0 && ctx->max_rows > 0) ev_gen_end = 10;
    else gen_event_pending = true;
    long long *hC = hp;  // this set's slot 0
    long long hC_fallback = 0;
    if (!hC) hC = &hC_fallback;
    if (fast) {
      // rows of every block are sorted by line id: sort-free placement
      ENSURE(ctx, ctx->d_node_rec, 16 * (size_t)std::max<long long>(G, 1));
      launch_node_prefix(st, G, ctx->d_node_img.as<int>(), ctx->d_seg_off.as<long long>(),
                         ctx->d_nb_off.as<long long>(), ctx->d_blk_line_base.as<long long>(),
                         ctx->d_cnt_bl.as<unsigned>(), ctx->d_base_bl.as<unsigned>(), ctx->d_ntris_u.as<unsigned>(),
                         ctx->d_tri_off.as<long long>(), ctx->d_scan_status.as<unsigned long long>(),
                         ctx->d_err.as<int>(), ctx->d_node_rec.p);  // tri_off = exclusive scan of the counts, in the same kernel
      node_rec_valid = true;
      ctx->cnt_bl_clean = true;
      // Nothing below needs the candidate count on the host (the kernels read tri_off[G]; the grids of
      // k_place / k_score3 do not depend on it) except the SIZE of the compact arrays.  While the trivial
      // bound -- one candidate per staging slot -- fits kCountFreeBytes, the arrays get that size and the
      // whole run is enqueued without a host round trip (the count then arrives with the error flag);
      // otherwise (or with LT_TEST_SYNC_COUNT) one 8-byte copy + stream sync fetches the exact count.
      const long long bound = P;
      // (288 GB of HBM: a sixth of it may go to bound-sized arrays before a run pays a host round trip for its count --
      // at 2e8 match rows the bound costs 20 GB, and the round trip kept lt_run_device_async from returning before the
      // generation stage had finished, i.e. from being asynchronous at all)
      constexpr long long kCountFreeBytes = 48ll << 30;
      // per candidate of the bound: permutation or moved record (LT_TEST_PLACE_COPY), score, flag, node, scoring prologue
      // record, and the pair slots of the split scoring form (256 entries of 16 bytes per tile of 64)
      const bool perm_bound = !test_switch("LT_TEST_PLACE_COPY");
      const long long per_cand = (perm_bound ? 4 : (long long)(sizeof(CRec) + 8)) + 8 + 4 + 4 + (long long)cand_meta_bytes() +
                                 (long long)score_split_entry_bytes() * 256 / 64;
      if (extras) {
        C_known = staged_total;  // the counting run of stage B already brought the count to the host
      } else if (ctx->h_pinned && bound * per_cand <= kCountFreeBytes && !test_switch("LT_TEST_SYNC_COUNT")) {
        C_known = -1;
        C_bound = bound;
      } else {
        HIPCHK(ctx, hipMemcpyAsync(hC, ctx->d_tri_off.as<long long>() + G, 8, hipMemcpyDeviceToHost, st));
        HIPCHK(ctx, hipStreamSynchronize(st));
        C_known = *hC;
      }
    } else {
      // generic rows: stable radix sort of the candidates by node (input is in row order)
      ENSURE(ctx, ctx->d_wave_pos, 8 * (size_t)(n_waves + 1));
      HIPCHK(ctx, hipMemsetAsync(ctx->d_wave_count.as<unsigned>() + n_waves, 0, 4, st));
      size_t tmp = scan_temp_bytes_u32_to_i64(n_waves + 1);
      ENSURE(ctx, ctx->d_scan_tmp, std::max<size_t>(tmp, 16));
      if (launch_scan_u32_to_i64(st, ctx->d_scan_tmp.p, tmp, n_waves + 1, ctx->d_wave_count.as<unsigned>(),
                                 ctx->d_wave_pos.as<long long>()) != 0)
        return fail(ctx, LT_ERR_HIP, "rocprim scan failed");
      HIPCHK(ctx, hipMemcpyAsync(hC, ctx->d_wave_pos.as<long long>() + n_waves, 8, hipMemcpyDeviceToHost, st));
      HIPCHK(ctx, hipStreamSynchronize(st));
      C_known = *hC;
    }
    // fast path: no record is moved -- k_place writes the permutation only
    const bool perm_mode = fast && !test_switch("LT_TEST_PLACE_COPY");
    ctx->perm_mode = perm_mode;
    ctx->compact_valid = !perm_mode;
    if (C_known < 0) {
      // the bound is generous: if the device cannot give that much, fetch the exact count after all
      const size_t Bn = (size_t)std::max<long long>(C_bound, 1);
      const bool got = (perm_mode ? ctx->d_place_perm.ensure(4 * Bn)
                                  : (ctx->d_cand.ensure(sizeof(CRec) * Bn) && ctx->d_lite.ensure(sizeof(double) * Bn))) &&
                       ctx->d_score.ensure(8 * Bn) && ctx->d_edge_flag.ensure(4 * Bn) && ctx->d_cand_node.ensure(4 * Bn) &&
                       ctx->d_cand_meta.ensure(cand_meta_bytes() * Bn);
      if (!got) {
        (void)hipGetLastError();
        HIPCHK(ctx, hipMemcpyAsync(hC, ctx->d_tri_off.as<long long>() + G, 8, hipMemcpyDeviceToHost, st));
        HIPCHK(ctx, hipStreamSynchronize(st));
        C_known = *hC;
      }
    }
    if (C_known >= 0) C_bound = C_known;
    const size_t Cn = (size_t)std::max<long long>(C_bound, 1);
    if (perm_mode) {
      ENSURE(ctx, ctx->d_place_perm, 4 * Cn);
    } else {
      ENSURE(ctx, ctx->d_cand, sizeof(CRec) * Cn); ENSURE(ctx, ctx->d_lite, sizeof(double) * Cn);
    }
    ENSURE(ctx, ctx->d_score, 8 * Cn); ENSURE(ctx, ctx->d_edge_flag, 4 * Cn); ENSURE(ctx, ctx->d_cand_node, 4 * Cn);
    ctx->cand_cap = (long long)Cn;
    if (fast) {
      launch_place(st, ctx->n_blk, ctx->max_rows, ctx->d_m_off.as<long long>(), ctx->d_blk_img.as<int>(),
                   ctx->d_seg_off.as<long long>(), ctx->d_blk_line_base.as<long long>(),
                   ctx->d_base_bl.as<unsigned>(), ctx->d_wave_count.as<unsigned>(), ctx->d_tri_off.as<long long>(),
                   ctx->d_st_c.as<CRec>(), ctx->d_st_l.as<double>(), ctx->d_st_key.as<unsigned>(),
                   ctx->d_cand.as<CRec>(), ctx->d_lite.as<double>(), ctx->d_cand_node.as<unsigned>(), group_base,
                   perm_mode ? ctx->d_place_perm.as<unsigned>() : nullptr, ctx->d_blk_surv.as<unsigned>(),
                   ctx->d_blk_rnd0.as<unsigned>(), ln ? ctx->d_round_count.as<unsigned>() : nullptr);
    } else {
      ENSURE(ctx, ctx->d_keys, 4 * Cn); ENSURE(ctx, ctx->d_rows, 4 * Cn);
      ENSURE(ctx, ctx->d_skeys, 4 * Cn); ENSURE(ctx, ctx->d_srows, 4 * Cn);
      launch_pack_keys(st, ctx->n_blk, ctx->max_rows, ctx->d_m_off.as<long long>(),
                       ctx->d_wave_count.as<unsigned>(), ctx->d_wave_pos.as<long long>(),
                       ctx->d_st_key.as<unsigned>(), ctx->d_keys.as<unsigned>(), ctx->d_rows.as<unsigned>(), group_base);
      if (C_known > 0) {
        int end_bit = bits_for(G + 1);
        size_t tmp = sort_temp_bytes(C_known, end_bit);
        ENSURE(ctx, ctx->d_sort_tmp, std::max<size_t>(tmp, 16));
        if (launch_sort(st, ctx->d_sort_tmp.p, tmp, C_known, ctx->d_keys.as<unsigned>(), ctx->d_skeys.as<unsigned>(),
                        ctx->d_rows.as<unsigned>(), ctx->d_srows.as<unsigned>(), end_bit) != 0)
          return fail(ctx, LT_ERR_HIP, "rocprim radix sort failed");
      }
      launch_node_offsets(st, C_known, G, ctx->d_skeys.as<unsigned>(), ctx->d_tri_off.as<long long>());
      launch_permute(st, C_known, ctx->d_skeys.as<unsigned>(), ctx->d_srows.as<unsigned>(), ctx->d_st_c.as<CRec>(),
                     ctx->d_st_l.as<double>(), ctx->d_cand.as<CRec>(), ctx->d_lite.as<double>(),
                     ctx->d_cand_node.as<unsigned>());
    }
    // ... and the one in front of k_score3 ends the placement stage (it then includes k_cand_meta)
    if (fine_score && C_bound > 0) ev_place_end = 11;
    else if (sampled) HIPCHK(ctx, hipEventRecord(ev[4], st));
    if (gen_event_pending) ev_gen_end = ev_place_end;
  } else if (ctx->job_mode == 2) {
    ctx->perm_mode = false;
    ctx->compact_valid = true;
    const size_t In = (size_t)std::max<long long>(P, 1);
    // VP-guided proposals: three survivor ballots per work item (algebraic, vp of l1, vp of l2)
    const bool vp_on = ctx->cfg.use_vp && !ctx->cfg.disable_vp_triangulation;
    if (vp_on && !ctx->vp_ready) return fail(ctx, LT_ERR_STATE, "use_vp is set but InitVPResults was not called");
    const double *seg_vp = vp_on ? ctx->d_seg_vp.as<double>() : nullptr;
    const unsigned char *seg_has_vp = vp_on ? ctx->d_seg_has_vp.as<unsigned char>() : nullptr;
    const int n_masks = vp_on ? 3 : 1;
    // point-guided proposals: a variable number of candidates per connection -> per-connection counts
    // (one byte each) instead of ballots, see k_gen_exhaustive_pts
    const bool pts_any = ctx->pts_ready && (!ctx->cfg.disable_many_points_triangulation || !ctx->cfg.disable_one_point_triangulation);
    const int many_on = (pts_any && !ctx->cfg.disable_many_points_triangulation) ? 1 : 0;
    const int one_on = (pts_any && !ctx->cfg.disable_one_point_triangulation) ? 1 : 0;
    const double *sfm_xyz = (pts_any && ctx->sfm_given) ? ctx->d_sfm_xyz.as<double>() : nullptr;
    // (+ one ballot word: the plain mode scans the popcounts of the ballots directly, the word behind the last is 0)
    ENSURE(ctx, ctx->d_masks, pts_any ? 128 * In : 8 * (In * n_masks + 1)); ENSURE(ctx, ctx->d_mask_cnt, 4 * (In + 1));
    ENSURE(ctx, ctx->d_mask_pos, 8 * (In + 1));
    // Plain exhaustive mode (no VP / point proposals): pass 1 with the neighbour lines held in registers (k_gates_ex;
    // LT_TEST_EX_PASS1_BLOCK keeps the wave-per-(node, neighbour) form the VP variant uses), and, while the staging
    // capacity holds, in its ONE-PASS form: pass 1 only lists the connections that pass the cheap gates (k_gates_ex<true>),
    // k_tri_ex evaluates the list densely and writes the survivors to staging slots, a permutation orders them -- no
    // second triangulation pass, no host round trip for the candidate count.  The capacity is a
    // fraction of the connections (1/6 until a run of this context has measured its need, then 1.4 x that); a run that
    // overflows it (device error flag 5) is repeated in the two-pass form by finish_run.  LT_TEST_EX_TWO_PASS: always
    // two passes.
    const bool plain = !pts_any && !vp_on;
    bool staged = plain && ctx->h_pinned && !ctx->ex_two_pass && P > 0 && !test_switch("LT_TEST_EX_TWO_PASS") &&
                  !test_switch("LT_TEST_EX_PASS1_BLOCK");
    long long ex_cap = 0;
    unsigned region_cap = 0;
    unsigned long long *ex_ctr = ctx->d_scan_status.as<unsigned long long>() + n_status_scan + score3_tile_buckets() * 16;
    if (staged) {
      double frac = ctx->ex_frac > 0.0 ? ctx->ex_frac : 1.0 / 6.0;
      long long slack = 65536;
      if (const char *f = test_switch("LT_TEST_EX_CAP_FRAC")) {  // test switch: force a (too small) capacity
        frac = atof(f);
        slack = 0;
      }
      const long long want = (long long)((double)ctx->n_conn * frac) + slack;
      const long long nreg = ex_regions();
      const long long rc8 = ((want + nreg - 1) / nreg + 63) & ~63ll;
      ex_cap = nreg * rc8;
      // ~210 bytes per slot over all arrays: beyond 64 GB (or the 32-bit slot index) the two-pass form, whose arrays
      // have the exact size
      if (ex_cap >= (1ll << 32) - 1 || ex_cap * 210 > (64ll << 30)) staged = false;
      else {
        region_cap = (unsigned)rc8;
        const size_t Bn = (size_t)ex_cap;
        const bool got = ctx->d_st_c.ensure(sizeof(CRec) * Bn) && ctx->d_st_l.ensure(sizeof(double) * Bn) &&
                         ctx->d_st_key.ensure(4 * Bn) && ctx->d_place_perm.ensure(4 * Bn) && ctx->d_score.ensure(8 * Bn) &&
                         ctx->d_edge_flag.ensure(4 * Bn) && ctx->d_cand_node.ensure(4 * Bn) &&
                         ctx->d_cand_meta.ensure(cand_meta_bytes() * Bn) && ctx->d_ex_rec.ensure(4 * Bn) &&
                         ctx->d_ex_ent.ensure(8 * Bn) && ctx->d_ex_z.ensure(4 * Bn);
        if (!got) {
          (void)hipGetLastError();
          staged = false;
        }
      }
    }
    ctx->ex_staged_set[set] = staged;
    if (pts_any) {
      launch_gen_exhaustive_pts(st, false, P, gcfg, ctx->d_item_off.as<long long>(), G, ctx->d_node_img.as<int>(),
                                ctx->d_nb_off.as<long long>(), ctx->d_blk_nb.as<int>(), ctx->d_seg_off.as<long long>(),
                                ctx->d_cams.as<Cam>(), ctx->d_segs.as<Seg>(), ctx->d_pairs.as<PairRec>(),
                                ctx->d_masks.as<unsigned short>(), ctx->d_mask_cnt.as<unsigned>(), nullptr, nullptr,
                                nullptr, seg_vp, seg_has_vp, ctx->d_seg_pt_off.as<long long>(), ctx->d_seg_pts.p,
                                sfm_xyz, ctx->d_err.as<int>(), many_on, one_on, ctx->d_blk_chunk_off.as<int>(),
                                ctx->max_nb, ctx->max_chunks, ctx->d_seg_gates.p);
    } else {
      if (plain && !test_switch("LT_TEST_EX_PASS1_BLOCK"))
        launch_gates_exhaustive(st, ctx->n_blk, ctx->max_chunks, P, gcfg, ctx->d_item_off.as<long long>(),
                                ctx->d_blk_img.as<int>(), ctx->d_blk_nb.as<int>(), ctx->d_seg_off.as<long long>(),
                                ctx->d_cams.as<Cam>(), ctx->d_segs.as<Seg>(), ctx->d_pairs.as<PairRec>(),
                                ctx->d_masks.as<unsigned long long>(), ctx->d_blk_chunk_off.as<int>(), ctx->d_seg_gates.p,
                                staged ? ctx->d_ex_ent.as<unsigned long long>() : nullptr, ex_ctr, region_cap,
                                ctx->d_err.as<int>());
      else
      launch_gen_exhaustive(st, false, P, gcfg, ctx->d_item_off.as<long long>(), G, ctx->d_node_img.as<int>(),
                            ctx->d_nb_off.as<long long>(), ctx->d_blk_nb.as<int>(), ctx->d_seg_off.as<long long>(),
                            ctx->d_cams.as<Cam>(), ctx->d_segs.as<Seg>(), ctx->d_pairs.as<PairRec>(),
                            ctx->d_masks.as<unsigned long long>(), nullptr, nullptr, nullptr, seg_vp, seg_has_vp,
                            ctx->d_blk_chunk_off.as<int>(), ctx->max_nb, ctx->max_chunks, ctx->d_seg_gates.p);
      if (staged) {
        HIPCHK(ctx, hipMemsetAsync(ctx->d_masks.p, 0, 8 * In, st));
        launch_tri_exhaustive(st, ctx->d_ex_ent.as<unsigned long long>(), ex_ctr, region_cap, gcfg, P,
                              ctx->d_item_off.as<long long>(), ctx->d_blk_img.as<int>(), ctx->d_blk_nb.as<int>(),
                              ctx->d_nb_off.as<long long>(), ctx->d_seg_off.as<long long>(), ctx->d_cams.as<Cam>(),
                              ctx->d_segs.as<Seg>(), ctx->d_pairs.as<PairRec>(), ctx->d_blk_chunk_off.as<int>(),
                              ctx->d_masks.as<unsigned long long>(), ctx->d_st_c.as<CRec>(), ctx->d_st_l.as<double>(),
                              ctx->d_st_key.as<unsigned>(), ctx->d_ex_z.as<float>());
      }
      if (!plain) launch_popc(st, P, ctx->d_masks.as<unsigned long long>(), ctx->d_mask_cnt.as<unsigned>(), n_masks);
    }
    if (plain) {
      // one ballot per item: the scan reads the ballots through a popcount iterator (no count pass, no count array)
      HIPCHK(ctx, hipMemsetAsync(ctx->d_masks.as<unsigned long long>() + P, 0, 8, st));
      size_t tmp = scan_temp_bytes_popc(P + 1);
      ENSURE(ctx, ctx->d_scan_tmp, std::max<size_t>(tmp, 16));
      if (launch_scan_popc(st, ctx->d_scan_tmp.p, tmp, P + 1, ctx->d_masks.as<unsigned long long>(),
                           ctx->d_mask_pos.as<long long>()) != 0)
        return fail(ctx, LT_ERR_HIP, "rocprim scan failed");
    } else {
      HIPCHK(ctx, hipMemsetAsync(ctx->d_mask_cnt.as<unsigned>() + P, 0, 4, st));
      size_t tmp = scan_temp_bytes_u32_to_i64(P + 1);
      ENSURE(ctx, ctx->d_scan_tmp, std::max<size_t>(tmp, 16));
      if (launch_scan_u32_to_i64(st, ctx->d_scan_tmp.p, tmp, P + 1, ctx->d_mask_cnt.as<unsigned>(),
                                 ctx->d_mask_pos.as<long long>()) != 0)
        return fail(ctx, LT_ERR_HIP, "rocprim scan failed");
    }
    if (staged) {
      launch_tri_offsets_ex(st, G, ctx->d_item_off.as<long long>(), ctx->d_mask_pos.as<long long>(), P, -1, ex_cap,
                            ctx->d_tri_off.as<long long>(), ctx->d_err.as<int>());
      if (sampled) HIPCHK(ctx, hipEventRecord(ev[3], st));
      launch_place_exhaustive(st, ex_ctr, region_cap, ctx->d_ex_ent.as<unsigned long long>(),
                              ctx->d_st_key.as<unsigned>(), ctx->d_item_off.as<long long>(),
                              ctx->d_blk_chunk_off.as<int>(), ctx->d_masks.as<unsigned long long>(),
                              ctx->d_mask_pos.as<long long>(), P, ctx->d_tri_off.as<long long>(), G,
                              ctx->d_place_perm.as<unsigned>(), (hp ? hp : ctx->d_result3.as<long long>()) + 3);
      ctx->ex_region_cap = region_cap;
      launch_cand_node(st, G, ctx->d_tri_off.as<long long>(), ctx->d_cand_node.as<unsigned>());
      ctx->perm_mode = true;
      ctx->compact_valid = false;
      ctx->cand_cap = ex_cap;
      C_known = -1;
      C_bound = ex_cap;
      if (sampled) HIPCHK(ctx, hipEventRecord(ev[4], st));
    } else {
    // the candidate count sizes the compacted arrays: one small host round trip
    long long total = 0;
    HIPCHK(ctx, hipMemcpyAsync(&total, ctx->d_mask_pos.as<long long>() + P, 8, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipStreamSynchronize(st));
    if (sampled) HIPCHK(ctx, hipEventRecord(ev[3], st));
    const size_t Cn = (size_t)std::max<long long>(total, 1);
    ENSURE(ctx, ctx->d_cand, sizeof(CRec) * Cn); ENSURE(ctx, ctx->d_lite, sizeof(double) * Cn);
    ENSURE(ctx, ctx->d_score, 8 * Cn); ENSURE(ctx, ctx->d_edge_flag, 4 * Cn);
    ctx->cand_cap = (long long)Cn;
    if (pts_any)
      launch_gen_exhaustive_pts(st, true, P, gcfg, ctx->d_item_off.as<long long>(), G, ctx->d_node_img.as<int>(),
                                ctx->d_nb_off.as<long long>(), ctx->d_blk_nb.as<int>(), ctx->d_seg_off.as<long long>(),
                                ctx->d_cams.as<Cam>(), ctx->d_segs.as<Seg>(), ctx->d_pairs.as<PairRec>(),
                                ctx->d_masks.as<unsigned short>(), ctx->d_mask_cnt.as<unsigned>(),
                                ctx->d_mask_pos.as<long long>(), ctx->d_cand.as<CRec>(), ctx->d_lite.as<double>(),
                                seg_vp, seg_has_vp, ctx->d_seg_pt_off.as<long long>(), ctx->d_seg_pts.p, sfm_xyz,
                                ctx->d_err.as<int>(), many_on, one_on, ctx->d_blk_chunk_off.as<int>(), ctx->max_nb,
                                ctx->max_chunks, ctx->d_seg_gates.p);
    else if (!vp_on && !test_switch("LT_TEST_EX_PASS2_BLOCK"))
      launch_fill_exhaustive(st, ctx->n_blk, P, gcfg, ctx->d_item_off.as<long long>(), ctx->d_blk_img.as<int>(),
                             ctx->d_blk_nb.as<int>(), ctx->d_nb_off.as<long long>(), ctx->d_seg_off.as<long long>(),
                             ctx->d_cams.as<Cam>(), ctx->d_segs.as<Seg>(), ctx->d_pairs.as<PairRec>(),
                             ctx->d_masks.as<unsigned long long>(), ctx->d_mask_pos.as<long long>(),
                             ctx->d_cand.as<CRec>(), ctx->d_lite.as<double>(), ctx->d_blk_chunk_off.as<int>());
    else
      launch_gen_exhaustive(st, true, P, gcfg, ctx->d_item_off.as<long long>(), G, ctx->d_node_img.as<int>(),
                            ctx->d_nb_off.as<long long>(), ctx->d_blk_nb.as<int>(), ctx->d_seg_off.as<long long>(),
                            ctx->d_cams.as<Cam>(), ctx->d_segs.as<Seg>(), ctx->d_pairs.as<PairRec>(),
                            ctx->d_masks.as<unsigned long long>(), ctx->d_mask_pos.as<long long>(),
                            ctx->d_cand.as<CRec>(), ctx->d_lite.as<double>(), seg_vp, seg_has_vp,
                            ctx->d_blk_chunk_off.as<int>(), ctx->max_nb, ctx->max_chunks, ctx->d_seg_gates.p);
    launch_tri_offsets_ex(st, G, ctx->d_item_off.as<long long>(), ctx->d_mask_pos.as<long long>(), P, total, -1,
                          ctx->d_tri_off.as<long long>(), ctx->d_err.as<int>());
    ENSURE(ctx, ctx->d_cand_node, 4 * Cn);
    launch_cand_node(st, G, ctx->d_tri_off.as<long long>(), ctx->d_cand_node.as<unsigned>());
    C_known = total;
    C_bound = total;
    if (sampled) HIPCHK(ctx, hipEventRecord(ev[4], st));
    }
  } else {
    ctx->perm_mode = false;
    ctx->compact_valid = true;
    HIPCHK(ctx, hipMemsetAsync(ctx->d_tri_off.p, 0, sizeof(long long) * (size_t)(G + 1), st));
    ENSURE(ctx, ctx->d_cand, sizeof(CRec)); ENSURE(ctx, ctx->d_lite, sizeof(double));
    ENSURE(ctx, ctx->d_score, 8); ENSURE(ctx, ctx->d_edge_flag, 4); ENSURE(ctx, ctx->d_cand_node, 4);
    C_known = 0;
    C_bound = 0;
    for (int k = 3; k <= 4; ++k)
      if (sampled) HIPCHK(ctx, hipEventRecord(ev[k], st));
  }

  // ---- scoring ----
  if (score3_lds_bytes(ctx->max_nb, score_f32) > 160 * 1024)
    return fail(ctx, LT_ERR_ARGUMENT, "too many neighbours for the scoring kernel's LDS budget");
  {
    if (ctx->h_nb_off[ctx->n_img] >= (1ll << 24))
      return fail(ctx, LT_ERR_ARGUMENT, "too many (image, neighbour) blocks in one batch (>= 2^24)");
    ENSURE(ctx, ctx->d_cand_meta, cand_meta_bytes() * (size_t)std::max<long long>(C_bound, 1));
    ENSURE(ctx, ctx->d_tile_order, 64 * 128);  // counters of the scoring stage, 128 B apart: draw queues of the fused kernel, overflow chunks, unit claims, k_score_q's XCD cells
    // tiles listed by cost class (LT_TEST_NO_TILE_CLASSES: natural tile order)
    // (matched mode only: the wide nodes of the exhaustive mode put every tile into the top class, whose one counter
    // per queue then serialises ~4e4 appends -- k_cand_meta 0.11 -> 0.50 ms -- for an order that changes nothing)
    const bool tile_classes = !test_switch("LT_TEST_NO_TILE_CLASSES") && ctx->job_mode == 1;
    const unsigned tile_cap = (unsigned)(((std::max<long long>(C_bound, 1) + 63) / 64 + 7) / 8);  // tiles of one draw queue
    if (tile_classes) ENSURE(ctx, ctx->d_tile_list, 16 * (size_t)tile_cap * (size_t)score3_tile_buckets());  // 16-byte entries
    // large nodes (exhaustive matching): depth-sorted sweep, see k_depth_order
    const bool score_sorted = score_f32 && ctx->job_mode == 2 && !test_switch("LT_TEST_SCORE_UNSORTED");
    if (score_sorted) {
      ENSURE(ctx, ctx->d_perm, 4 * (size_t)std::max<long long>(C_bound, 1));
      ENSURE(ctx, ctx->d_rng, 4 * (size_t)std::max<long long>(C_bound, 1));
    }
    // depth-sorted sweep over the staged records of the one-pass exhaustive mode: see k_depth_order
    const bool staged_sorted = score_sorted && ctx->perm_mode && ctx->job_mode == 2;
    C_run = C_bound;  // replaced by the exact count when that arrives with the error flag (finish_run)
    // split form (default for the single-precision sweep): the sweep writes pair chunks, k_dense8 evaluates them with
    // full rounds; LT_SCORE_FUSED=1 or a chunk store that overflowed once: the fused kernel
    // (the exhaustive mode's depth-sorted tiles carry ~25 pairs each and k_dense8's units there are a few tiles whatever they
    // hold: 1.16 + 1.44 ms against 2.54 fused in round 5, 1.15 + 1.13 against 2.36 since round 6 -- the fused kernel keeps its
    // 10 KB table of maxima per wave through the sweep, six waves to a CU.  LT_SCORE_SPLIT=1 forces the split form for
    // the natural tile order too, for the tests.)
    const bool split = score_f32 && !ctx->score_fused && !test_switch("LT_SCORE_FUSED") &&
                       (tile_classes || staged_sorted || test_switch("LT_SCORE_SPLIT"));
    long long sp_chunks = 0;
    int sp_slot_cap = ctx->job_mode == 2 ? 64 : 256;  // entries per tile slot (matched: p90 of the bench scene is 182 pairs)
    if (split) {
      const long long n_tiles_b = (std::max<long long>(C_bound, 1) + 63) / 64;
      sp_chunks = score_split_chunks(std::max<long long>(C_bound, 1));
      if (const char *e = test_switch("LT_TEST_SPLIT_CHUNKS")) sp_chunks = std::max(0, atoi(e));
      if (const char *e = test_switch("LT_TEST_SPLIT_SLOT")) sp_slot_cap = std::max(1, atoi(e));
      ENSURE(ctx, ctx->d_sp_slots, score_split_entry_bytes() * (size_t)sp_slot_cap * (size_t)n_tiles_b);
      ENSURE(ctx, ctx->d_sp_cnt, 4 * (size_t)n_tiles_b);
      ENSURE(ctx, ctx->d_sp_ovf, 4 * (size_t)n_tiles_b);
      ENSURE(ctx, ctx->d_sp_pairs, score_split_chunk_bytes() * (size_t)std::max<long long>(sp_chunks, 1));
      ENSURE(ctx, ctx->d_sp_desc, 8 * (size_t)std::max<long long>(sp_chunks, 1));
    }
    // the sweep lists its finished tiles by their true pair count for k_dense8 (LT_TEST_NO_PAIR_CLASSES: units in the order
    // of k_cand_meta's cost classes, as in round 5)
    const bool pair_classes = split && tile_classes && !test_switch("LT_TEST_NO_PAIR_CLASSES");
    if (pair_classes) {
      ENSURE(ctx, ctx->d_pc_cnt, 128 * (size_t)score3_tile_buckets());
      ENSURE(ctx, ctx->d_pc_list, 8 * (size_t)tile_cap * (size_t)score3_tile_buckets());
    }
    launch_score3(st, C_bound, G, ctx->d_tri_off.as<long long>(), ctx->d_cand_node.as<unsigned>(), ctx->d_cand_meta.p,
                  ctx->perm_mode ? ctx->d_st_c.as<CRec>() : ctx->d_cand.as<CRec>(), ctx->d_node_img.as<int>(),
                  ctx->d_nb_off.as<long long>(), ctx->d_blk_order.as<int>(), ctx->d_cams.as<Cam>(),
                  ctx->d_score.as<double>(), ctx->d_pair_counter.as<unsigned long long>(), ctx->max_nb, scfg,
                  guard2, fine_score ? ev[11] : nullptr, ctx->d_tile_order.as<unsigned>(), score_f32,
                  (ctx->perm_mode && !staged_sorted) ? ctx->d_place_perm.as<unsigned>()
                                                     : (score_sorted ? ctx->d_perm.as<unsigned>() : nullptr),
                  score_sorted ? ctx->d_rng.p : nullptr, ctx->perm_mode && !staged_sorted,
                  tile_classes ? (unsigned *)(ctx->d_scan_status.as<unsigned long long>() + n_status_scan) : nullptr,
                  tile_classes ? ctx->d_tile_list.as<unsigned>() : nullptr, tile_cap,
                  staged_sorted ? ctx->d_place_perm.as<unsigned>() : nullptr,
                  staged_sorted ? ctx->d_ex_rec.as<unsigned>() : nullptr,
                  staged_sorted ? ctx->d_ex_z.as<float>() : nullptr, ctx->d_err.as<int>(),
                  split ? ctx->d_sp_slots.p : nullptr, sp_slot_cap, ctx->d_sp_cnt.as<unsigned>(),
                  ctx->d_sp_ovf.as<unsigned>(), ctx->d_sp_pairs.p, ctx->d_sp_desc.p, sp_chunks, sampled ? ev[5] : nullptr,
                  node_rec_valid ? ctx->d_node_rec.p : nullptr, pair_classes ? ctx->d_pc_cnt.as<unsigned>() : nullptr,
                  pair_classes ? ctx->d_pc_list.p : nullptr, tile_cap,
                  // the one-kernel form k_score_q (default; LT_SCORE_TWO_KERNELS=1 or device flag 8 once: sweep kernel + k_dense8;
                  // its pair entries carry the neighbour word in 26 bits; 2 = LT_TEST_Q_LOSE_TILE, a tile is never published)
                  // (+ 4 = LT_TEST_DENSE_TABLES: k_dense8's per-tile tables in the exhaustive mode too, instead of k_dense_rows;
                  //  + 8 = LT_TEST_DENSE_FEW_ROWS: k_dense_rows with a table of 64 rows)
                  (((ctx->score_two_kernels || test_switch("LT_SCORE_TWO_KERNELS") || ctx->n_img >= (1 << 18))
                        ? 0 : (test_switch("LT_TEST_Q_LOSE_TILE") ? 2 : 1)) |
                   (test_switch("LT_TEST_DENSE_TABLES") ? 4 : 0) | (test_switch("LT_TEST_DENSE_FEW_ROWS") ? 8 : 0)));
    if (C_bound <= 0 && sampled) HIPCHK(ctx, hipEventRecord(ev[5], st));  // nothing to score: no kernel carries the event
  }
  ENSURE(ctx, ctx->d_best_idx, 8 * (size_t)std::max<long long>(G, 1));
  ENSURE(ctx, ctx->d_nvalid, 4 * (size_t)(G + 1));
  ENSURE(ctx, ctx->d_edge_off, 8 * (size_t)(G + 1));
  ENSURE(ctx, ctx->d_best_c, sizeof(Cand) * (size_t)std::max<long long>(G, 1));
  ENSURE(ctx, ctx->d_best_score, 8 * (size_t)std::max<long long>(G, 1));
  ENSURE(ctx, ctx->d_best_src, 8 * (size_t)std::max<long long>(G, 1));
  ENSURE(ctx, ctx->d_ntris, 4 * (size_t)std::max<long long>(G, 1));
  // per node: best candidate (gathered into the dense per-node arrays by the same kernel), valid-edge
  // flags and their number; the edge offsets (a scan) and the edge lists are produced at download time
  launch_select(st, G, ctx->d_tri_off.as<long long>(), ctx->d_score.as<double>(), scfg.fullscore_th,
                scfg.max_valid_conns, ctx->d_best_idx.as<long long>(), ctx->d_edge_flag.as<unsigned>(),
                ctx->d_nvalid.as<unsigned>(), ctx->perm_mode ? ctx->d_st_c.as<CRec>() : ctx->d_cand.as<CRec>(),
                ctx->perm_mode ? ctx->d_st_l.as<double>() : ctx->d_lite.as<double>(),
                ctx->d_best_c.as<Cand>(), ctx->d_best_score.as<double>(), ctx->d_best_src.as<int>(),
                ctx->d_ntris.as<int>(), /*wide=*/ctx->job_mode == 2, ctx->d_err.as<int>(),
                ctx->d_pair_counter.as<unsigned long long>(), hp ? hp : ctx->d_result3.as<long long>(),
                ctx->perm_mode ? ctx->d_place_perm.as<unsigned>() : nullptr);
  if (!hp) HIPCHK(ctx, hipEventRecord(ev[7], st));  // with result slots the end marker below also ends the run
  HIPCHK(ctx, hipGetLastError());
  // the device error flag, the candidate count and the pair statistic ride on the stream into this set's
  // pinned slots; finish_run reads them behind the end marker
  hipEvent_t ev_end = nullptr;
  if (hp) {
    // hp[0] candidate count, hp[1] error flag, hp[2] pair statistic: one record, WRITTEN BY k_select straight into this
    // set's page-locked slots (round 4: the 32-byte device-to-host copy was a 4 us blit kernel in every step); hp[3]: fullest
    // staging region of the one-pass exhaustive mode (k_place_ex).  The slots were zeroed when the run was enqueued (no
    // nodes: k_select does not run).
    if (!ctx->ev_end[0])
      for (auto &e : ctx->ev_end) HIPCHK(ctx, hipEventCreate(&e));
    ev_end = ctx->ev_end[ctx->run_seq++ % 3u];
    HIPCHK(ctx, hipEventRecord(ev_end, st));
  }
  int rc_prev = LT_OK;
  if (ctx->run_pending) {  // the previous run (the other set)
    ctx->in_run_async = true;
    rc_prev = finish_run(ctx);
    ctx->in_run_async = false;
  }
  ctx->run_pending = true;
  ctx->pend_ev_start = ev_start;
  ctx->pend_sampled = sampled;
  ctx->pend_ev_end = ev_end;
  ctx->pend_set = set;
  ctx->pend_count_on_device = C_known < 0;
  ctx->pend_fine_gen = fine_gen;
  ctx->pend_fine_score = fine_score;
  ctx->pend_C = C_run;
  ctx->pend_ev_gen_end = ev_gen_end;
  ctx->pend_ev_place_end = ev_place_end;
  ctx->ran = true;
  ctx->downloaded = false;
  ctx->host_view_valid = false;
  return rc_prev;
}

int lt_run_device(lt_ctx *ctx) {
  int rc = lt_run_device_async(ctx);
  if (rc) return rc;
  return finish_run(ctx);
}

// Images that have no results (yet) hold a value-initialised best candidate, like the reference's TriTuple.
extern "C++" {
namespace lt_impl {
void define_best_of_other_images(lt_ctx *ctx) {
  for (int i = 0; i < ctx->n_img; ++i)
    if (!ctx->best_c_set[(size_t)i]) {
      const long long a = ctx->seg_off[i], b = ctx->seg_off[i + 1];
      if (b > a) std::memset((void *)(ctx->best_c + a), 0, sizeof(Cand) * (size_t)(b - a));
      ctx->best_c_set[(size_t)i] = 2;  // defined, but no results: nothing the device does not know
    }
}
}  // namespace lt_impl
}  // extern "C++"

// The split host-side view (Cand / CandLite in candidate order) of the last run's candidates, for the debug
// read-outs: converted on demand from the 128-byte device records -- through the placement permutation when the
// records are still in the staging lists.
extern "C++" {
namespace lt_impl {
int materialize_compact(lt_ctx *ctx) {
  if (ctx->host_view_valid) return LT_OK;
  const long long C = ctx->C_last;
  const size_t Cn = (size_t)std::max<long long>(C, 1);
  ENSURE(ctx, ctx->d_hcand, sizeof(Cand) * Cn); ENSURE(ctx, ctx->d_hlite, sizeof(CandLite) * Cn);
  launch_host_view(ctx->stream, C, ctx->perm_mode ? ctx->d_place_perm.as<unsigned>() : nullptr,
                   ctx->perm_mode ? ctx->d_st_c.as<CRec>() : ctx->d_cand.as<CRec>(),
                   ctx->perm_mode ? ctx->d_st_l.as<double>() : ctx->d_lite.as<double>(), ctx->d_hcand.as<Cand>(),
                   ctx->d_hlite.as<CandLite>());
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  ctx->host_view_valid = true;
  return LT_OK;
}
}  // namespace lt_impl
}  // extern "C++"

int lt_download(lt_ctx *ctx) {
  LT_RANGE("lt_download (per-node results -> host)");
  LT_FINISH(ctx);
  if (!ctx->ran) return fail(ctx, LT_ERR_STATE, "lt_download before lt_run_device");
  if (ctx->downloaded) return LT_OK;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  double t0 = now_ms();
  hipStream_t st = ctx->stream;
  const long long G = ctx->G;
  // edges need their final positions: offsets (scan of the per-node counts) and lists are made now
  HIPCHK(ctx, hipMemsetAsync(ctx->d_nvalid.as<unsigned>() + G, 0, 4, st));
  {
    size_t tmp = scan_temp_bytes_u32_to_i64(G + 1);
    ENSURE(ctx, ctx->d_scan_tmp, std::max<size_t>(tmp, 16));
    if (launch_scan_u32_to_i64(st, ctx->d_scan_tmp.p, tmp, G + 1, ctx->d_nvalid.as<unsigned>(),
                               ctx->d_edge_off.as<long long>()) != 0)
      return fail(ctx, LT_ERR_HIP, "rocprim scan failed");
  }
  std::vector<long long> tri_off(G + 1), edge_off(G + 1);
  HIPCHK(ctx, hipMemcpyAsync(tri_off.data(), ctx->d_tri_off.p, 8 * (size_t)(G + 1), hipMemcpyDeviceToHost, st));
  HIPCHK(ctx, hipMemcpyAsync(edge_off.data(), ctx->d_edge_off.p, 8 * (size_t)(G + 1), hipMemcpyDeviceToHost, st));
  HIPCHK(ctx, hipStreamSynchronize(st));
  ctx->C = tri_off[G];
  ctx->E = edge_off[G];
  ENSURE(ctx, ctx->d_edges, 8 * (size_t)std::max<long long>(ctx->E, 1));
  launch_edge_fill(st, G, ctx->d_tri_off.as<long long>(), ctx->d_edge_flag.as<unsigned>(),
                   ctx->d_edge_off.as<long long>(), ctx->perm_mode ? ctx->d_st_c.as<CRec>() : ctx->d_cand.as<CRec>(),
                   ctx->d_edges.as<int>(), ctx->perm_mode ? ctx->d_place_perm.as<unsigned>() : nullptr);
  // one pooled page-locked block for all result arrays
  const size_t Gn = (size_t)std::max<long long>(G, 1), En = (size_t)std::max<long long>(ctx->E, 1);
  const size_t o_bc = 0, o_bs = o_bc + sizeof(Cand) * Gn, o_src = o_bs + 8 * Gn, o_nt = o_src + 8 * Gn,
               o_ed = (o_nt + 4 * Gn + 15) / 16 * 16, total = o_ed + 8 * En;
  lt_host::HostBlock hb = lt_host::host_block_acquire(total);
  if (!hb.p) return fail(ctx, LT_ERR_RUNTIME, "out of host memory for the results");
  struct Rel {
    lt_host::HostBlock b;
    ~Rel() { lt_host::host_block_release(b); }
  } rel{hb};
  char *base = (char *)hb.p;
  const Cand *bc = (const Cand *)(base + o_bc);
  const double *bs = (const double *)(base + o_bs);
  const int *bsrc = (const int *)(base + o_src);
  const int *nt = (const int *)(base + o_nt);
  const int *edges = (const int *)(base + o_ed);
  if (G > 0) {
    HIPCHK(ctx, hipMemcpyAsync(base + o_bc, ctx->d_best_c.p, sizeof(Cand) * (size_t)G, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipMemcpyAsync(base + o_bs, ctx->d_best_score.p, 8 * (size_t)G, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipMemcpyAsync(base + o_src, ctx->d_best_src.p, 8 * (size_t)G, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipMemcpyAsync(base + o_nt, ctx->d_ntris.p, 4 * (size_t)G, hipMemcpyDeviceToHost, st));
  }
  if (ctx->E > 0)
    HIPCHK(ctx, hipMemcpyAsync(base + o_ed, ctx->d_edges.p, 8 * (size_t)ctx->E, hipMemcpyDeviceToHost, st));
  HIPCHK(ctx, hipStreamSynchronize(st));
  // merge the nodes of the job's images into the persistent per-node results; the edge lists of the
  // whole run are appended to the pool in one piece (nodes outside the job have none)
  const long long pool_base = (long long)ctx->valid_edges.pool.size();
  ctx->valid_edges.pool.insert(ctx->valid_edges.pool.end(), edges, edges + 2 * (size_t)ctx->E);
  long long pairs = 0;
  const long long n_job = (long long)ctx->job_imgs.size();
#pragma omp parallel for num_threads(lt::host_threads()) schedule(dynamic, 4) reduction(+ : pairs)
  for (long long j = 0; j < n_job; ++j) {
    const int idx = ctx->job_imgs[(size_t)j];
    ctx->best_c_set[(size_t)idx] = 1;
    for (long long g = ctx->seg_off[idx]; g < ctx->seg_off[idx + 1]; ++g) {
      ctx->n_tris[g] = nt[g];
      pairs += (long long)nt[g] * nt[g];
      ctx->has_best[g] = nt[g] > 0 ? 1 : 0;
      ctx->best_c[g] = bc[g];
      ctx->best_score[g] = bs[g];
      // src image index -> id
      ctx->best_src2[2 * g] = nt[g] > 0 ? ctx->img_ids[bsrc[2 * g]] : 0;
      ctx->best_src2[2 * g + 1] = nt[g] > 0 ? bsrc[2 * g + 1] : 0;
      ctx->valid_edges.off[(size_t)g] = pool_base + 2 * edge_off[g];
      ctx->valid_edges.cnt[(size_t)g] = (int)(2 * (edge_off[g + 1] - edge_off[g]));
    }
  }
  define_best_of_other_images(ctx);
  ctx->stat_pairs = pairs;
  if (ctx->cfg.debug_mode && ctx->C > 0) {  // keep this batch's tris_ on the host (later batches reuse the device arrays)
    const long long C = ctx->C;
    {
      int rcm = materialize_compact(ctx);
      if (rcm) return rcm;
    }
    std::vector<Cand> c((size_t)C);
    std::vector<CandLite> l((size_t)C);
    std::vector<double> sc((size_t)C);
    HIPCHK(ctx, hipMemcpy(c.data(), ctx->d_hcand.p, sizeof(Cand) * (size_t)C, hipMemcpyDeviceToHost));
    HIPCHK(ctx, hipMemcpy(l.data(), ctx->d_hlite.p, sizeof(CandLite) * (size_t)C, hipMemcpyDeviceToHost));
    HIPCHK(ctx, hipMemcpy(sc.data(), ctx->d_score.p, 8 * (size_t)C, hipMemcpyDeviceToHost));
    for (long long j = 0; j < n_job; ++j) {
      const int idx = ctx->job_imgs[(size_t)j];
      for (long long g = ctx->seg_off[idx]; g < ctx->seg_off[idx + 1]; ++g) {
        ctx->dbg_off[(size_t)g] = (long long)ctx->dbg_pool.size();
        ctx->dbg_cnt[(size_t)g] = (int)(tri_off[g + 1] - tri_off[g]);
        for (long long t = tri_off[g]; t < tri_off[g + 1]; ++t) {
          lt_ctx::DebugTri r;
          for (int k = 0; k < 3; ++k) { r.line10[k] = c[t].s[k]; r.line10[3 + k] = c[t].e[k]; }
          r.line10[6] = c[t].depth[0]; r.line10[7] = c[t].depth[1]; r.line10[8] = c[t].unc; r.line10[9] = c[t].score3;
          r.score = sc[t];
          r.src2[0] = ctx->img_ids[lite_img(l[t])];
          r.src2[1] = l[t].ng_line;
          ctx->dbg_pool.push_back(r);
        }
      }
    }
  }
  ctx->downloaded = true;
  ctx->timers[9] = now_ms() - t0;
  return LT_OK;
}

int lt_flush(lt_ctx *ctx) {
  int rc;
  if (ctx->job_mode == 0 && !ctx->uploaded) {
    if (!ctx->inited) return fail(ctx, LT_ERR_STATE, "flush before Init");
    ctx->downloaded = true;
    return LT_OK;
  }
  if (!ctx->uploaded && (rc = lt_upload(ctx))) return rc;
  if (!ctx->ran && (rc = lt_run_device(ctx))) return rc;
  if (!ctx->downloaded && (rc = lt_download(ctx))) return rc;
  return LT_OK;
}

}  // extern "C"
