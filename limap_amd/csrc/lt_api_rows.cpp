// lt_api_rows.cpp -- C ABI, part 2: TriangulateImage / TriangulateImageExhaustiveMatch / TriangulateAll -- validation
// and buffering of the match rows (base_line_triangulator.cc:79-98), streamed to HBM while the caller goes on.
#include "lt_host.h"

using namespace lt;
using namespace lt_impl;

extern "C" {

// ---- the host pass over the match rows (lt_rows.h), shared out over the persistent team (lt_pool.h) ----
// The staged form of a block is the COMPRESSED one of lt_rows.h (17 bits per row) at word offset `dst` of the stream
// (cb_words(n) words, known before the pass); a block that cannot take it (a line step other than 0 / +1) goes in the
// plain one-word-per-row form to the context's overflow array instead, and its place in the stream stays unused.
struct RowBlk {  // one (image, neighbour) block of rows
  const int32_t *src; long long n, dst; long long M1, M2; int img_id, nb_id;
};
struct RowJob {
  const RowBlk *blks; int nb_total; const int *chunk_of; std::atomic<int> *chunk_done; int *bad; unsigned *out;
  long long *ovf_off = nullptr;  // per block: -1 = compressed in the stream, else first word in the overflow array
  int *line0 = nullptr;          // per block: line id of its first row
  std::vector<unsigned> *ovf = nullptr;
  std::mutex *ovf_mu = nullptr;
  std::atomic<int> next_blk{0}, bad_any{0}, uns_any{0};
  static void run(void *arg, int, int) {
    RowJob &J = *static_cast<RowJob *>(arg);
    int uns_t = 0;
    for (;;) {
      const int b = J.next_blk.fetch_add(1, std::memory_order_relaxed);
      if (b >= J.nb_total) break;
      const RowBlk &B = J.blks[b];
      const lt::RowStats rs = lt::pack_rows_cb(B.src, B.n, J.out + B.dst);
      int err = 0;
      if (B.n > 0 && (unsigned long long)rs.mx_line >= (unsigned long long)B.M1) err |= 1;
      if (B.n > 0 && (unsigned long long)rs.mx_ng >= (unsigned long long)B.M2) err |= 2;
      J.bad[b] = err;
      J.line0[b] = B.n > 0 ? (int)B.src[0] : 0;
      J.ovf_off[b] = -1;
      if (rs.irregular && !err) {  // not the matchers' shape: the plain form, line | neighbour line << 16
        // (the only allocations of the pass: out of memory becomes the block's error code 4 -- on a pool worker an
        // exception would be swallowed and the chunk never counted as done; on the caller it would cross extern "C")
        try {
          static thread_local std::vector<unsigned> tmp;
          tmp.resize((size_t)B.n);
          (void)lt::pack_rows(B.src, B.n, tmp.data());
          std::lock_guard<std::mutex> lk(*J.ovf_mu);
          J.ovf_off[b] = (long long)J.ovf->size();
          J.ovf->insert(J.ovf->end(), tmp.begin(), tmp.end());
        } catch (const std::exception &) {
          err = 4;
          J.bad[b] = err;
          J.ovf_off[b] = -1;
        }
      }
      uns_t |= rs.unsorted;
      if (err) J.bad_any.store(1, std::memory_order_relaxed);
      if (J.chunk_done) J.chunk_done[J.chunk_of[b]].fetch_add(1, std::memory_order_release);
    }
    if (uns_t) J.uns_any.store(1, std::memory_order_relaxed);
  }
};

static int begin_image(lt_ctx *ctx, int img_id, int mode, int *idx_out) {
  if (!ctx->inited) return fail(ctx, LT_ERR_STATE, "TriangulateImage called before Init");
  auto it = ctx->id2idx.find(img_id);
  if (it == ctx->id2idx.end()) return fail(ctx, LT_ERR_ARGUMENT, "unknown image id " + std::to_string(img_id));
  *idx_out = it->second;
  // already_scored_ guard (global_line_triangulator.cc:73): the call changes nothing -- in particular it does not
  // invalidate the results or tracks of the batch this image belongs to (the callers return right behind this)
  if (ctx->triangulated[(size_t)it->second]) return LT_OK;
  // ComputeLineTracks ended the batch: with the tail on the device the per-node results are still there -- fetch them
  // now, so that this call starts a new batch (like the host tail, which downloads before it runs) instead of
  // appending to the finished one
  if (ctx->tracks_done && ctx->ran && !ctx->downloaded) {
    int rc = lt_download(ctx);
    if (rc) return rc;
  }
  if (ctx->job_mode != 0 && ctx->job_mode != mode && !ctx->downloaded) {
    int rc = lt_flush(ctx);  // switching between matched and exhaustive calls: run what is buffered
    if (rc) return rc;
  }
  if (ctx->downloaded) {  // a new batch after results were read: start a fresh job
    ctx->job_imgs.clear(); ctx->job_nbs.clear(); ctx->job_order.clear();
    ctx->h_m_off.assign(1, 0); ctx->h_m_pairs.clear(); ctx->streamed_ints = 0;
    ctx->h_c_off.clear(); ctx->h_ovf_off.clear(); ctx->h_line0.clear(); ctx->h_ovf.clear();
    ctx->rows_sorted = true;
    ctx->uploaded = ctx->ran = ctx->downloaded = false;
  }
  ctx->job_mode = mode;
  ctx->uploaded = ctx->ran = false;
  ctx->tracks_done = false;
  return LT_OK;
}

// TriangulateImage with the rows of every neighbour given by its own pointer (no concatenation on the
// caller's side).  Validation (base_line_triangulator.cc:79,87-94), the sortedness probe for the
// sort-free placement and the single copy into the staging buffer run in one parallel pass.
int lt_triangulate_image_rows(lt_ctx *ctx, int img_id, int n_nb, const int32_t *nb_ids,
                              const int32_t *const *rows, const int64_t *n_rows) {
  LT_FINISH(ctx);
  struct Acc {  // [12] host ms spent buffering match rows (all calls of the batch)
    lt_ctx *c; double t0;
    ~Acc() { c->timers[12] += now_ms() - t0; }
  } acc{ctx, now_ms()};
  int idx;
  int rc = begin_image(ctx, img_id, 1, &idx);
  if (rc) return rc;
  if (ctx->triangulated[idx]) return LT_OK;  // already_scored_ guard (global_line_triangulator.cc:73)
  if (n_nb > 255) return fail(ctx, LT_ERR_ARGUMENT, "more than 255 neighbours (uint8 neighbour index, base_line_triangulator.h:15)");
  // the reference iterates std::map<int, MatrixXi>: ascending neighbour id
  std::vector<int> order(n_nb);
  for (int k = 0; k < n_nb; ++k) order[k] = k;
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return nb_ids[a] < nb_ids[b]; });
  std::vector<int> nbs(n_nb), ord(n_nb);
  std::vector<long long> M2(n_nb), dst(n_nb + 1, 0);  // dst: word offsets of the blocks' compressed forms
  const long long M1 = ctx->seg_off[idx + 1] - ctx->seg_off[idx];
  for (int k = 0; k < n_nb; ++k) {
    int o = order[k];
    if (k > 0 && nb_ids[o] == nb_ids[order[k - 1]]) return fail(ctx, LT_ERR_ARGUMENT, "duplicate neighbour id in matches");
    auto it = ctx->id2idx.find(nb_ids[o]);
    if (it == ctx->id2idx.end()) return fail(ctx, LT_ERR_ARGUMENT, "unknown neighbour image id " + std::to_string(nb_ids[o]));
    if (n_rows[o] < 0) return fail(ctx, LT_ERR_ARGUMENT, "negative row count");
    nbs[k] = it->second;
    ord[k] = k;  // already ascending id
    M2[k] = ctx->seg_off[it->second + 1] - ctx->seg_off[it->second];
    dst[k + 1] = dst[k] + (long long)lt::cb_words(n_rows[o]);
  }
  const size_t base = ctx->h_m_pairs.size();
  const size_t ovf_base = ctx->h_ovf.size();
  {
    // the staging block may move when it grows: no asynchronous copy may still be reading it
    size_t want = base + (size_t)dst[n_nb];
    if (ctx->job_imgs.empty() && dst[n_nb] > 0)  // first image of a batch: one allocation for the usual case
      // (every image of the scene in one batch; capped at 1 GB -- a large scene arrives in batches, and a
      // page-locked allocation costs ~0.1 s per GB)
      want = std::max(want, std::min<size_t>((size_t)dst[n_nb] * (size_t)std::max(1, ctx->n_img) + 1024, (size_t)1 << 28));
    if (want > ctx->h_m_pairs.capacity()) {
      if (ctx->streamed_ints > 0) HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
      if (!ctx->h_m_pairs.reserve(want)) return fail(ctx, LT_ERR_RUNTIME, "out of host memory for the match rows");
    }
  }
  if (!ctx->h_m_pairs.grow_to(base + (size_t)dst[n_nb])) return fail(ctx, LT_ERR_RUNTIME, "out of host memory for the match rows");
  int *out = ctx->h_m_pairs.data() + base;
  // ONE pass over the rows (lt_rows.h): validation as reductions + the staged copy, packed to one word per row; the
  // blocks are shared out between this thread and the workers of the persistent team that are awake (lt_pool.h)
  std::vector<int> bad(n_nb, 0);
  std::vector<RowBlk> blks((size_t)n_nb);
  for (int k = 0; k < n_nb; ++k)
    blks[(size_t)k] = RowBlk{rows[order[k]], n_rows[order[k]], dst[k], M1, M2[k], img_id, nb_ids[order[k]]};
  std::vector<long long> ovf_off((size_t)std::max(n_nb, 1), -1);
  std::vector<int> line0((size_t)std::max(n_nb, 1), 0);
  RowJob job;
  job.blks = blks.data(); job.nb_total = n_nb; job.chunk_of = nullptr; job.chunk_done = nullptr;
  job.bad = bad.data(); job.out = reinterpret_cast<unsigned *>(out);
  job.ovf_off = ovf_off.data(); job.line0 = line0.data(); job.ovf = &ctx->h_ovf; job.ovf_mu = &ctx->ovf_mu;
  if (dst[n_nb] >= (1 << 13)) {
    lt_host::SpinPool &pool = lt_host::SpinPool::get(lt_host::row_workers());
    const bool shared = pool.begin(&RowJob::run, &job);  // false: the team is busy with another caller's job
    RowJob::run(&job, 0, 0);
    if (shared) pool.end();
  } else {
    RowJob::run(&job, 0, 0);  // a few rows: not worth a notify
  }
  if (job.uns_any.load()) ctx->rows_sorted = false;
  for (int k = 0; k < n_nb; ++k) {
    if (!bad[k]) continue;
    ctx->h_m_pairs.grow_to(base);
    ctx->h_ovf.resize(ovf_base);
    if (bad[k] & 4) return fail(ctx, LT_ERR_RUNTIME, "out of host memory while buffering the match rows");
    if (bad[k] & 1)  // base_line_triangulator.cc:87-94
      return fail(ctx, LT_ERR_RUNTIME,
                  "IndexError! Out-of-index matches exist between image (img_id = " + std::to_string(img_id) +
                      ") and neighbor image (img_id = " + std::to_string(nb_ids[order[k]]) +
                      "). Please make sure you are reusing the correct descriptors and matches when using the "
                      "--skip_exists option.");
    return fail(ctx, LT_ERR_RUNTIME, "IndexError! neighbour line id out of range in matches of image " + std::to_string(img_id));
  }
  // stream the rows to the device while the caller prepares the next image (they are final: staging
  // is in call order, which is the device order whenever the images arrive in ascending id order)
  // (one copy per ~4 MB of rows: an enqueue costs the host ~5 us, an image brings ~0.8 MB; lt_upload sends the rest)
  if (ctx->h_m_pairs.blk.pinned && ctx->streamed_ints <= base && dst[n_nb] > 0 &&
      base + (size_t)dst[n_nb] - ctx->streamed_ints >= (1u << 20)) {
    const size_t from = ctx->streamed_ints, end = base + (size_t)dst[n_nb];
    if (hipSetDevice(ctx->device) == hipSuccess) {
      bool ok = true;
      if (sizeof(int) * end > ctx->d_c_stream.cap) {
        // grow the device buffer (first copy: sized for the whole batch), keeping the streamed prefix
        DevBuf nb;
        size_t want = sizeof(int) * std::max(end, ctx->h_m_pairs.capacity());
        ok = nb.ensure(want);
        if (ok && from > 0)
          ok = hipMemcpyAsync(nb.p, ctx->d_c_stream.p, sizeof(int) * from, hipMemcpyDeviceToDevice, ctx->stream) == hipSuccess &&
               hipStreamSynchronize(ctx->stream) == hipSuccess;
        if (ok) {
          ctx->d_c_stream.take(nb);
        } else {
          nb.release();
          (void)hipGetLastError();
        }
      }
      if (ok && hipMemcpyAsync(ctx->d_c_stream.as<int>() + from, ctx->h_m_pairs.data() + from, sizeof(int) * (end - from),
                               hipMemcpyHostToDevice, ctx->stream) == hipSuccess)
        ctx->streamed_ints = end;
      else
        (void)hipGetLastError();  // not fatal: lt_upload sends whatever was not streamed
    }
  }
  for (int k = 0; k < n_nb; ++k) {
    ctx->h_m_off.push_back(ctx->h_m_off.back() + n_rows[order[k]]);
    ctx->h_c_off.push_back((long long)base + dst[k]);
    ctx->h_ovf_off.push_back(ovf_off[(size_t)k]);
    ctx->h_line0.push_back(line0[(size_t)k]);
  }
  ctx->job_imgs.push_back(idx);
  ctx->job_nbs.push_back(nbs);
  ctx->job_order.push_back(ord);
  ctx->neighbors[idx] = nbs;
  ctx->triangulated[idx] = 1;
  return LT_OK;
}

int lt_triangulate_all_rows(lt_ctx *ctx, int n_images, const int32_t *img_ids, const int64_t *nb_off, const int32_t *nb_ids,
                            const int32_t *const *rows, const int64_t *n_rows) {
  LT_FINISH(ctx);
  struct Acc {  // [12] host ms spent buffering match rows
    lt_ctx *c; double t0;
    ~Acc() { c->timers[12] += now_ms() - t0; }
  } acc{ctx, now_ms()};
  if (n_images < 0 || (n_images > 0 && (!img_ids || !nb_off))) return fail(ctx, LT_ERR_ARGUMENT, "null argument");
  struct Img {
    int idx; std::vector<int> nbs, ord; std::vector<long long> cnt;
  };
  static const bool all_trace = getenv("LT_TAIL_TRACE") != nullptr;
  double tl = acc.t0;
  auto lap = [&](const char *what) {
    if (!all_trace) return;
    double t = now_ms();
    std::fprintf(stderr, "[all] %-18s %.3f ms\n", what, t - tl);
    tl = t;
  };
  std::vector<RowBlk> blks;
  std::vector<Img> imgs;
  long long total_rows = 0;  // in WORDS of the stream: the blocks' compressed forms one behind the other
  std::vector<char> seen_here((size_t)std::max(ctx->n_img, 1), 0);
  // ---- pass 1 (serial, cheap): the per-image bookkeeping of lt_triangulate_image_rows, block descriptors ----
  for (int k = 0; k < n_images; ++k) {
    int idx;
    int rc = begin_image(ctx, img_ids[k], 1, &idx);
    if (rc) return rc;
    if (ctx->triangulated[idx] || seen_here[(size_t)idx]) continue;  // already_scored_ guard (global_line_triangulator.cc:73)
    seen_here[(size_t)idx] = 1;
    const int n_nb = (int)(nb_off[k + 1] - nb_off[k]);
    const int32_t *nb = nb_ids + nb_off[k];
    if (n_nb > 255) return fail(ctx, LT_ERR_ARGUMENT, "more than 255 neighbours (uint8 neighbour index, base_line_triangulator.h:15)");
    std::vector<int> order(n_nb);
    for (int e = 0; e < n_nb; ++e) order[e] = e;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return nb[a] < nb[b]; });  // std::map order
    Img im;
    im.idx = idx;
    const long long M1 = ctx->seg_off[idx + 1] - ctx->seg_off[idx];
    for (int e = 0; e < n_nb; ++e) {
      const int o = order[e];
      if (e > 0 && nb[o] == nb[order[e - 1]]) return fail(ctx, LT_ERR_ARGUMENT, "duplicate neighbour id in matches");
      auto it = ctx->id2idx.find(nb[o]);
      if (it == ctx->id2idx.end()) return fail(ctx, LT_ERR_ARGUMENT, "unknown neighbour image id " + std::to_string(nb[o]));
      const long long n = n_rows[nb_off[k] + o];
      if (n < 0) return fail(ctx, LT_ERR_ARGUMENT, "negative row count");
      im.nbs.push_back(it->second);
      im.ord.push_back(e);
      im.cnt.push_back(n);
      blks.push_back(RowBlk{rows[nb_off[k] + o], n, total_rows, M1, ctx->seg_off[it->second + 1] - ctx->seg_off[it->second],
                         img_ids[k], nb[o]});
      total_rows += (long long)lt::cb_words(n);
    }
    imgs.push_back(std::move(im));
  }
  if (imgs.empty()) return LT_OK;
  // read only now: pass 1 writes no rows, but its first begin_image() starts a new batch when the previous one has been
  // read back (results downloaded / tracks computed) and then clears the staging block, its offsets and streamed_ints
  const size_t base = ctx->h_m_pairs.size();
  const size_t ovf_base = ctx->h_ovf.size();
  lap("bookkeeping");
  // ---- staging: one allocation for the whole call ----
  {
    const size_t want = base + (size_t)total_rows;
    if (want > ctx->h_m_pairs.capacity()) {
      if (ctx->streamed_ints > 0) HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
      if (!ctx->h_m_pairs.reserve(want)) return fail(ctx, LT_ERR_RUNTIME, "out of host memory for the match rows");
    }
    if (!ctx->h_m_pairs.grow_to(want)) return fail(ctx, LT_ERR_RUNTIME, "out of host memory for the match rows");
  }
  int *out = ctx->h_m_pairs.data() + base;
  lap("staging");
  // ---- pass 2: validation (reductions over the rows) + the single copy, in CHUNKS of >= 8 MB of rows: one parallel
  // region per chunk over its (image, neighbour) blocks, and the chunk's host -> device copy enqueued right behind it, so
  // that the DMA of chunk c runs under the host pass of chunk c + 1 (one copy at the end left 1.5 ms of DMA exposed) ----
  const int nb_total = (int)blks.size();
  std::vector<int> bad((size_t)nb_total, 0);
  int unsorted = 0;
  bool stream_ok = ctx->h_m_pairs.blk.pinned && ctx->streamed_ints <= base && total_rows > 0 &&
                   hipSetDevice(ctx->device) == hipSuccess;
  if (stream_ok && ctx->streamed_ints < base) {
    // rows of earlier calls that were not streamed yet go first (the device buffer is filled in order)
    stream_ok = false;
  }
  if (stream_ok) {
    const size_t end = base + (size_t)total_rows;
    if (sizeof(int) * end > ctx->d_c_stream.cap) {
      DevBuf nbuf;
      bool ok = nbuf.ensure(sizeof(int) * std::max(end, ctx->h_m_pairs.capacity()));
      if (ok && base > 0)
        ok = hipMemcpyAsync(nbuf.p, ctx->d_c_stream.p, sizeof(int) * base, hipMemcpyDeviceToDevice, ctx->stream) == hipSuccess &&
             hipStreamSynchronize(ctx->stream) == hipSuccess;
      if (ok) {
        ctx->d_c_stream.take(nbuf);
      } else {
        nbuf.release();
        (void)hipGetLastError();
        stream_ok = false;
      }
    }
  }
  // The workers (lt_pool.h: a persistent team, already awake when Init preceded this call) take blocks in order from
  // a shared counter; this thread does no row work -- it waits for each chunk (>= 2 MB of packed rows) to be complete
  // and enqueues its host -> device copy, so the DMA of chunk c runs under the workers' pass over chunk c + 1 (a copy
  // at the very end left 1.5 ms of DMA exposed; 8 MB chunks delayed the first copy by a fifth of the pass)
  constexpr long long kChunkRows = 512 << 10;
  std::vector<int> chunk_end;  // block index behind every chunk
  {
    long long acc_rows = 0;
    for (int b = 0; b < nb_total; ++b) {
      acc_rows += (long long)lt::cb_words(blks[(size_t)b].n);
      if (acc_rows >= kChunkRows || b == nb_total - 1) {
        chunk_end.push_back(b + 1);
        acc_rows = 0;
      }
    }
  }
  const int n_chunks = (int)chunk_end.size();
  std::vector<int> chunk_of((size_t)nb_total);
  for (int c = 0, b = 0; c < n_chunks; ++c)
    for (; b < chunk_end[(size_t)c]; ++b) chunk_of[(size_t)b] = c;
  std::vector<std::atomic<int>> chunk_done((size_t)n_chunks);
  for (auto &x : chunk_done) x.store(0, std::memory_order_relaxed);
  std::vector<long long> ovf_off((size_t)std::max(nb_total, 1), -1);
  std::vector<int> line0((size_t)std::max(nb_total, 1), 0);
  RowJob job;
  job.blks = blks.data(); job.nb_total = nb_total; job.chunk_of = chunk_of.data(); job.chunk_done = chunk_done.data();
  job.bad = bad.data(); job.out = reinterpret_cast<unsigned *>(out);
  job.ovf_off = ovf_off.data(); job.line0 = line0.data(); job.ovf = &ctx->h_ovf; job.ovf_mu = &ctx->ovf_mu;
  int *const d_rows = stream_ok ? ctx->d_c_stream.as<int>() : nullptr;
  int *const h_rows = ctx->h_m_pairs.data();
  size_t streamed_to = ctx->streamed_ints;
  lap("device buffer");
  lt_host::SpinPool &pool = lt_host::SpinPool::get(lt_host::row_workers());
  const bool shared = pool.begin(&RowJob::run, &job);
  if (!shared) RowJob::run(&job, 0, 0);  // the team is busy with another caller's job (or has no threads): this thread
                                         // packs every block itself, the chunk loop below then only enqueues the copies
  {
    bool ok = d_rows != nullptr;
    for (int c = 0; c < n_chunks; ++c) {
      const int first = c == 0 ? 0 : chunk_end[(size_t)c - 1], want = chunk_end[(size_t)c] - first;
      while (chunk_done[(size_t)c].load(std::memory_order_acquire) < want) {
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
      }
      if (!ok || job.bad_any.load(std::memory_order_relaxed)) continue;
      const size_t from = base + (size_t)blks[(size_t)first].dst;
      const size_t to = base + (size_t)(blks[(size_t)chunk_end[(size_t)c] - 1].dst +
                                        (long long)lt::cb_words(blks[(size_t)chunk_end[(size_t)c] - 1].n));
      if (to > from) {
        if (hipMemcpyAsync(d_rows + from, h_rows + from, sizeof(int) * (to - from), hipMemcpyHostToDevice, ctx->stream) == hipSuccess)
          streamed_to = to;
        else {
          (void)hipGetLastError();  // not fatal: lt_upload sends whatever was not streamed
          ok = false;
        }
      }
    }
  }
  if (shared) pool.end();
  lap("row pass");
  unsorted = job.uns_any.load();
  const bool any_bad = job.bad_any.load() != 0;
  if (streamed_to > ctx->streamed_ints) ctx->streamed_ints = streamed_to;
  for (int b = 0; b < nb_total && any_bad; ++b) {  // the first offending block in call order raises, like the per-image calls
    if (!bad[(size_t)b]) continue;
    if (ctx->streamed_ints > base) {  // chunks of this call are already on their way: they are void
      (void)hipStreamSynchronize(ctx->stream);
      ctx->streamed_ints = base;
    }
    ctx->h_m_pairs.grow_to(base);
    ctx->h_ovf.resize(ovf_base);
    const RowBlk &B = blks[(size_t)b];
    if (bad[(size_t)b] & 4) return fail(ctx, LT_ERR_RUNTIME, "out of host memory while buffering the match rows");
    if (bad[(size_t)b] & 1)  // base_line_triangulator.cc:87-94
      return fail(ctx, LT_ERR_RUNTIME,
                  "IndexError! Out-of-index matches exist between image (img_id = " + std::to_string(B.img_id) +
                      ") and neighbor image (img_id = " + std::to_string(B.nb_id) +
                      "). Please make sure you are reusing the correct descriptors and matches when using the "
                      "--skip_exists option.");
    return fail(ctx, LT_ERR_RUNTIME, "IndexError! neighbour line id out of range in matches of image " + std::to_string(B.img_id));
  }
  if (unsorted) ctx->rows_sorted = false;
  {
    size_t b = 0;
    for (Img &im : imgs)
      for (size_t e = 0; e < im.cnt.size(); ++e, ++b) {
        ctx->h_c_off.push_back((long long)base + blks[b].dst);
        ctx->h_ovf_off.push_back(ovf_off[b]);
        ctx->h_line0.push_back(line0[b]);
      }
  }
  for (Img &im : imgs) {
    for (long long n : im.cnt) ctx->h_m_off.push_back(ctx->h_m_off.back() + n);
    ctx->job_imgs.push_back(im.idx);
    ctx->neighbors[im.idx] = im.nbs;
    ctx->job_nbs.push_back(std::move(im.nbs));
    ctx->job_order.push_back(std::move(im.ord));
    ctx->triangulated[im.idx] = 1;
  }
  return LT_OK;
}

int lt_triangulate_image(lt_ctx *ctx, int img_id, int n_nb, const int32_t *nb_ids, const int64_t *m_off,
                         const int32_t *m_pairs) {
  std::vector<const int32_t *> rows(std::max(n_nb, 1));
  std::vector<int64_t> n_rows(std::max(n_nb, 1));
  for (int k = 0; k < n_nb; ++k) {
    rows[k] = m_pairs + 2 * m_off[k];
    n_rows[k] = m_off[k + 1] - m_off[k];
  }
  return lt_triangulate_image_rows(ctx, img_id, n_nb, nb_ids, rows.data(), n_rows.data());
}

int lt_triangulate_image_exhaustive(lt_ctx *ctx, int img_id, int n_nb, const int32_t *nb_ids) {
  LT_FINISH(ctx);
  int idx;
  int rc = begin_image(ctx, img_id, 2, &idx);
  if (rc) return rc;
  if (ctx->triangulated[idx]) return LT_OK;
  if (n_nb > 255) return fail(ctx, LT_ERR_ARGUMENT, "more than 255 neighbours (uint8 neighbour index, base_line_triangulator.h:15)");
  std::vector<int> nbs;
  for (int k = 0; k < n_nb; ++k) {
    auto it = ctx->id2idx.find(nb_ids[k]);
    if (it == ctx->id2idx.end()) return fail(ctx, LT_ERR_ARGUMENT, "unknown neighbour image id " + std::to_string(nb_ids[k]));
    for (int p = 0; p < k; ++p)
      if (nb_ids[p] == nb_ids[k]) return fail(ctx, LT_ERR_ARGUMENT, "duplicate neighbour id in neighbors list");
    nbs.push_back(it->second);
  }
  // exhaustive mode keeps the caller's neighbour order (:113-114); the per-image support sum
  // still runs over ascending image ids (std::map score_table, global_line_triangulator.cc:83,110)
  std::vector<int> ord(n_nb);
  for (int k = 0; k < n_nb; ++k) ord[k] = k;
  std::stable_sort(ord.begin(), ord.end(), [&](int a, int b) { return nb_ids[a] < nb_ids[b]; });
  ctx->job_imgs.push_back(idx);
  ctx->job_nbs.push_back(nbs);
  ctx->job_order.push_back(ord);
  ctx->neighbors[idx] = nbs;
  ctx->triangulated[idx] = 1;
  return LT_OK;
}

}  // extern "C"
