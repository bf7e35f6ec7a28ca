// lt_dist_rccl.cpp -- liblimap_amd_rccl.so: the multi-GPU exchange of include/limap_amd_rccl.h for a C / C++ host,
// on RCCL directly (limap_amd/dist.py is the same protocol over torch.distributed).  Uses only the public C ABI of
// liblimap_amd.so (lt_init_device, lt_set_scene_chunks, lt_refresh_scene_chunks, lt_shard_*), the HIP runtime and rccl.h.
// images sharded in id order, ONE all-gather of kvec | qvec | tvec | segs before the run, the shards'
// per-node results and valid-edge keys to rank 0 afterwards (global_line_triangulator.cc:138-151 writes only the own
// nodes of an image; :234-351 is the serial tail).
#include "../../include/limap_amd_rccl.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

struct lt_dist {
  lt_ctx *ctx = nullptr;
  ncclComm_t comm = nullptr;
  hipStream_t st = nullptr;
  int rank = 0, world = 1, n_img = 0;
  std::vector<int32_t> img_ids;
  std::vector<int64_t> seg_off, bounds, sizes;  // sizes: doubles of every rank's packed slice
  int64_t max_size = 0;
  double *d_local = nullptr, *d_recv = nullptr;            // packed slice of this rank | world x max_size
  double *d_k = nullptr, *d_q = nullptr, *d_t = nullptr, *d_s = nullptr;  // the unpacked scene (first gather: lt_init_device)
  bool initialised = false;
  void *d_blob = nullptr, *d_got = nullptr;  // merge: this rank's blob | rank 0: world blobs
  size_t blob_cap = 0, got_cap = 0;
  long long *d_cnt = nullptr;  // 3 x world int64 (size exchange of the two-collective merge)
  long long h_hdr[8] = {0}, h_mine[3] = {0};  // sources of asynchronous copies: must outlive an early error return
  std::string err;
};

namespace {

int fail(lt_dist *d, int code, const std::string &msg) {
  d->err = msg;
  return code;
}
#define DHIP(d, call)                                                                         \
  do {                                                                                        \
    hipError_t e_ = (call);                                                                   \
    if (e_ != hipSuccess) return fail(d, LT_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
  } while (0)
#define DNCCL(d, call)                                                                         \
  do {                                                                                         \
    ncclResult_t r_ = (call);                                                                  \
    if (r_ != ncclSuccess) return fail(d, LT_ERR_HIP, std::string(#call) + ": " + ncclGetErrorString(r_)); \
  } while (0)
#define DLT(d, call)                                                                   \
  do {                                                                                 \
    int rc_ = (call);                                                                  \
    if (rc_ != LT_OK) return fail(d, rc_, std::string(#call) + ": " + lt_last_error(d->ctx)); \
  } while (0)

int ensure(lt_dist *d, void **p, size_t *cap, size_t bytes) {
  if (*cap >= bytes && *p) return LT_OK;
  if (*p) DHIP(d, hipFree(*p));
  *p = nullptr;
  *cap = 0;
  DHIP(d, hipMalloc(p, std::max<size_t>(bytes, 64)));
  *cap = std::max<size_t>(bytes, 64);
  return LT_OK;
}

}  // namespace

extern "C" {

int lt_dist_shard_bounds(int n_img, int world, const double *weights, int64_t *bounds) {
  if (n_img < 0 || world < 1 || !bounds) return LT_ERR_ARGUMENT;
  // limap_amd.dist.shard_bounds: cut the cumulative weight at total * r / world (first index whose prefix reaches it)
  std::vector<double> cum((size_t)n_img + 1, 0.0);
  for (int i = 0; i < n_img; ++i) cum[(size_t)i + 1] = cum[(size_t)i] + (weights ? weights[i] : 1.0);
  const double total = cum[(size_t)n_img];
  bounds[0] = 0;
  for (int r = 1; r < world; ++r) {
    const double target = total * (double)r / (double)world;
    int64_t b = (int64_t)(std::lower_bound(cum.begin(), cum.end(), target) - cum.begin());
    b = std::min<int64_t>(std::max<int64_t>(b, bounds[r - 1]), n_img);
    bounds[r] = b;
  }
  bounds[world] = n_img;
  return LT_OK;
}

lt_dist *lt_dist_create(lt_ctx *ctx, void *rccl_comm, void *hip_stream, int rank, int world, int n_img,
                        const int32_t *img_ids, const int64_t *seg_off, const double *weights) {
  if (!ctx || !rccl_comm || world < 1 || rank < 0 || rank >= world || n_img <= 0 || !img_ids || !seg_off) return nullptr;
  for (int i = 1; i < n_img; ++i)
    if (img_ids[i] <= img_ids[i - 1]) return nullptr;  // ascending ids: the gathered arrays are a plain concatenation
  lt_dist *d = new lt_dist();
  d->ctx = ctx;
  d->comm = static_cast<ncclComm_t>(rccl_comm);
  d->st = static_cast<hipStream_t>(hip_stream);
  d->rank = rank; d->world = world; d->n_img = n_img;
  d->img_ids.assign(img_ids, img_ids + n_img);
  d->seg_off.assign(seg_off, seg_off + n_img + 1);
  d->bounds.resize((size_t)world + 1);
  lt_dist_shard_bounds(n_img, world, weights, d->bounds.data());
  d->sizes.resize((size_t)world);
  for (int r = 0; r < world; ++r) {
    const int64_t a = d->bounds[r], b = d->bounds[r + 1];
    d->sizes[r] = 11 * (b - a) + 4 * (d->seg_off[b] - d->seg_off[a]);
    d->max_size = std::max(d->max_size, d->sizes[r]);
  }
  d->max_size = std::max<int64_t>(d->max_size, 1);
  const int64_t G = std::max<int64_t>(d->seg_off[n_img], 1);
  bool ok = lt_set_stream(ctx, hip_stream) == LT_OK;
  ok = ok && hipMalloc((void **)&d->d_local, 8 * (size_t)d->max_size) == hipSuccess;
  ok = ok && hipMalloc((void **)&d->d_recv, 8 * (size_t)d->max_size * (size_t)world) == hipSuccess;
  ok = ok && hipMalloc((void **)&d->d_k, 8 * 4 * (size_t)n_img) == hipSuccess;
  ok = ok && hipMalloc((void **)&d->d_q, 8 * 4 * (size_t)n_img) == hipSuccess;
  ok = ok && hipMalloc((void **)&d->d_t, 8 * 3 * (size_t)n_img) == hipSuccess;
  ok = ok && hipMalloc((void **)&d->d_s, 8 * 4 * (size_t)G) == hipSuccess;
  ok = ok && hipMalloc((void **)&d->d_cnt, 8 * 3 * (size_t)world) == hipSuccess;
  if (!ok) {
    lt_dist_destroy(d);
    return nullptr;
  }
  return d;
}

void lt_dist_destroy(lt_dist *d) {
  if (!d) return;
  for (void *p : {(void *)d->d_local, (void *)d->d_recv, (void *)d->d_k, (void *)d->d_q, (void *)d->d_t, (void *)d->d_s,
                  d->d_blob, d->d_got, (void *)d->d_cnt})
    if (p) (void)hipFree(p);
  delete d;
}

const char *lt_dist_last_error(const lt_dist *d) { return d ? d->err.c_str() : "null lt_dist"; }

int lt_dist_my_images(const lt_dist *d, int *first, int *last) {
  if (!d || !first || !last) return LT_ERR_ARGUMENT;
  *first = (int)d->bounds[d->rank];
  *last = (int)d->bounds[d->rank + 1];
  return LT_OK;
}

int lt_dist_load_local(lt_dist *d, const double *kvec, const double *qvec, const double *tvec, const double *segs) {
  if (!d || !kvec || !qvec || !tvec || !segs) return LT_ERR_ARGUMENT;
  const int64_t a = d->bounds[d->rank], b = d->bounds[d->rank + 1], n = b - a;
  const int64_t s0 = d->seg_off[a], s1 = d->seg_off[b];
  // packed slice: kvec | qvec | tvec | segs of the own images (what SceneGather.load_local builds)
  std::vector<double> buf((size_t)d->sizes[d->rank]);
  double *o = buf.data();
  std::memcpy(o, kvec + 4 * a, 8 * 4 * (size_t)n); o += 4 * n;
  std::memcpy(o, qvec + 4 * a, 8 * 4 * (size_t)n); o += 4 * n;
  std::memcpy(o, tvec + 3 * a, 8 * 3 * (size_t)n); o += 3 * n;
  std::memcpy(o, segs + 4 * s0, 8 * 4 * (size_t)(s1 - s0));
  // (ordered behind whatever still reads the previous contents on the stream; the pageable source makes the copy
  // synchronous with respect to the host)
  if (!buf.empty()) DHIP(d, hipMemcpyAsync(d->d_local, buf.data(), 8 * buf.size(), hipMemcpyHostToDevice, d->st));
  DHIP(d, hipStreamSynchronize(d->st));
  return LT_OK;
}

int lt_dist_all_gather_scene(lt_dist *d) {
  if (!d) return LT_ERR_ARGUMENT;
  // ONE collective: every rank contributes max_size doubles (its packed slice, padded)
  DNCCL(d, ncclAllGather(d->d_local, d->d_recv, (size_t)d->max_size, ncclDouble, d->comm, d->st));
  if (d->initialised) {
    DLT(d, lt_refresh_scene_chunks(d->ctx));  // invariants rebuilt straight from the receive buffer, same stream
    return LT_OK;
  }
  // first gather: the context is initialised from contiguous arrays (device-to-device unpack, once), and the receive
  // buffer's chunks are registered for the per-step path
  std::vector<int32_t> img_begin;
  std::vector<const void *> pk, pq, pt, ps;
  for (int r = 0; r < d->world; ++r) {
    const int64_t a = d->bounds[r], b = d->bounds[r + 1], n = b - a;
    const double *base = d->d_recv + (size_t)r * (size_t)d->max_size;
    if (n > 0) {
      const int64_t s0 = d->seg_off[a], s1 = d->seg_off[b];
      DHIP(d, hipMemcpyAsync(d->d_k + 4 * a, base, 8 * 4 * (size_t)n, hipMemcpyDeviceToDevice, d->st));
      DHIP(d, hipMemcpyAsync(d->d_q + 4 * a, base + 4 * n, 8 * 4 * (size_t)n, hipMemcpyDeviceToDevice, d->st));
      DHIP(d, hipMemcpyAsync(d->d_t + 3 * a, base + 8 * n, 8 * 3 * (size_t)n, hipMemcpyDeviceToDevice, d->st));
      if (s1 > s0)
        DHIP(d, hipMemcpyAsync(d->d_s + 4 * s0, base + 11 * n, 8 * 4 * (size_t)(s1 - s0), hipMemcpyDeviceToDevice, d->st));
    }
    if (n == 0 && r > 0) continue;  // empty shards are skipped (as SceneGather.chunk_pointers does)
    img_begin.push_back((int32_t)a);
    pk.push_back(base); pq.push_back(base + 4 * n); pt.push_back(base + 8 * n); ps.push_back(base + 11 * n);
  }
  DLT(d, lt_init_device(d->ctx, d->n_img, d->img_ids.data(), d->d_k, d->d_q, d->d_t, d->seg_off.data(), d->d_s));
  DLT(d, lt_set_scene_chunks(d->ctx, (int)img_begin.size(), img_begin.data(), pk.data(), pq.data(), pt.data(), ps.data()));
  d->initialised = true;
  return LT_OK;
}

int lt_dist_merge_shards(lt_dist *d, int64_t key_cap, int64_t *n_keys_merged) {
  if (!d) return LT_ERR_ARGUMENT;
  if (n_keys_merged) *n_keys_merged = 0;
  if (d->world == 1) return LT_OK;
  const int world = d->world, rank = d->rank;
  std::vector<int64_t> lo((size_t)world), hi((size_t)world);
  int64_t max_nodes = 1;
  for (int r = 0; r < world; ++r) {
    lo[r] = d->seg_off[d->bounds[r]];
    hi[r] = d->seg_off[d->bounds[r + 1]];
    max_nodes = std::max(max_nodes, hi[r] - lo[r]);
  }
  int64_t n_keys = 0;
  DLT(d, lt_shard_count(d->ctx, &n_keys));
  std::vector<long long> counts;  // known up front only in the two-collective form
  int64_t max_keys = key_cap;
  if (key_cap <= 0) {
    long long *mine = d->h_mine;
    mine[0] = (long long)n_keys; mine[1] = (long long)lo[rank]; mine[2] = (long long)hi[rank];
    DHIP(d, hipMemcpyAsync(d->d_cnt + 3 * rank, mine, sizeof(d->h_mine), hipMemcpyHostToDevice, d->st));
    DNCCL(d, ncclAllGather(d->d_cnt + 3 * rank, d->d_cnt, 3, ncclInt64, d->comm, d->st));
    std::vector<long long> all((size_t)3 * world);
    DHIP(d, hipMemcpyAsync(all.data(), d->d_cnt, 8 * all.size(), hipMemcpyDeviceToHost, d->st));
    DHIP(d, hipStreamSynchronize(d->st));
    counts.resize((size_t)world);
    max_keys = 1;
    for (int r = 0; r < world; ++r) {
      counts[r] = all[(size_t)3 * r];
      // a peer that sharded differently (other weights, another seg_off) would be imported into the wrong node range
      if (counts[r] < 0 || all[(size_t)3 * r + 1] != lo[r] || all[(size_t)3 * r + 2] != hi[r]) {
        char msg[200];
        std::snprintf(msg, sizeof(msg), "lt_dist_merge_shards: rank %d reports node range [%lld, %lld), expected [%lld, %lld)", r,
                      all[(size_t)3 * r + 1], all[(size_t)3 * r + 2], (long long)lo[r], (long long)hi[r]);
        return fail(d, LT_ERR_ARGUMENT, msg);
      }
      max_keys = std::max<int64_t>(max_keys, counts[r]);
    }
  }
  max_keys = std::max<int64_t>(max_keys, 1);
  const bool truncated = rank != 0 && n_keys > max_keys;  // (rank 0 sends no keys: its own are already in place)
  const size_t nb = (size_t)lt_shard_node_bytes();
  const size_t o_nodes = 64, o_keys = 64 + (((size_t)max_nodes * nb + 63) / 64) * 64;
  const size_t blob_bytes = o_keys + (size_t)max_keys * 8;
  int rc = ensure(d, &d->d_blob, &d->blob_cap, blob_bytes);
  if (rc) return rc;
  long long *hdr = d->h_hdr;
  hdr[0] = (long long)n_keys; hdr[1] = (long long)lo[rank]; hdr[2] = (long long)hi[rank]; hdr[3] = truncated ? 1 : 0;
  hdr[4] = hdr[5] = hdr[6] = hdr[7] = 0;
  DHIP(d, hipMemcpyAsync(d->d_blob, hdr, sizeof(d->h_hdr), hipMemcpyHostToDevice, d->st));
  if (!counts.empty()) {
    long long total = 0;
    for (long long c : counts) total += c;
    DLT(d, lt_shard_build(d->ctx, rank == 0 ? total : n_keys));
  } else if (rank != 0) {
    DLT(d, lt_shard_build(d->ctx, n_keys));
  }
  if (rank != 0 && !truncated)
    DLT(d, lt_shard_export(d->ctx, lo[rank], hi[rank], (char *)d->d_blob + o_nodes, (char *)d->d_blob + o_keys));
  if (rank == 0) {
    rc = ensure(d, &d->d_got, &d->got_cap, blob_bytes * (size_t)world);
    if (rc) return rc;
  }
  // the one collective of the merge: every other rank's blob to rank 0
  DNCCL(d, ncclGroupStart());
  if (rank == 0) {
    for (int r = 1; r < world; ++r)
      DNCCL(d, ncclRecv((char *)d->d_got + blob_bytes * (size_t)r, blob_bytes, ncclUint8, r, d->comm, d->st));
  } else {
    DNCCL(d, ncclSend(d->d_blob, blob_bytes, ncclUint8, 0, d->comm, d->st));
  }
  DNCCL(d, ncclGroupEnd());
  DHIP(d, hipStreamSynchronize(d->st));
  if (truncated) {
    char msg[256];
    std::snprintf(msg, sizeof(msg), "lt_dist_merge_shards: rank %d has %lld valid-edge keys, the blob has room for %lld (key_cap)",
                  rank, (long long)n_keys, (long long)max_keys);
    return fail(d, LT_ERR_ARGUMENT, msg);
  }
  if (rank != 0) return LT_OK;
  if (counts.empty()) {  // one collective: the headers carry the counts
    counts.assign((size_t)world, 0);
    counts[0] = n_keys;
    for (int r = 1; r < world; ++r) {
      long long h[8];
      DHIP(d, hipMemcpy(h, (char *)d->d_got + blob_bytes * (size_t)r, sizeof(h), hipMemcpyDeviceToHost));
      if (h[3]) {
        char msg[200];
        std::snprintf(msg, sizeof(msg), "lt_dist_merge_shards: rank %d could not send its keys (key_cap = %lld too small)", r,
                      (long long)max_keys);
        return fail(d, LT_ERR_ARGUMENT, msg);
      }
      // the header is a peer's word: a count beyond the blob's key room would make lt_shard_import read past it, a
      // different node range would import the slices into the wrong place
      if (h[0] < 0 || h[0] > (long long)max_keys || h[1] != (long long)lo[r] || h[2] != (long long)hi[r]) {
        char msg[256];
        std::snprintf(msg, sizeof(msg), "lt_dist_merge_shards: rank %d's header says %lld keys for nodes [%lld, %lld); expected "
                      "at most %lld keys for [%lld, %lld)", r, h[0], h[1], h[2], (long long)max_keys, (long long)lo[r], (long long)hi[r]);
        return fail(d, LT_ERR_ARGUMENT, msg);
      }
      counts[r] = h[0];
    }
    long long total = 0;
    for (long long c : counts) total += c;
    DLT(d, lt_shard_build(d->ctx, total));
  }
  long long total = 0;
  for (long long c : counts) total += c;
  for (int r = 1; r < world; ++r) {
    const char *b = (const char *)d->d_got + blob_bytes * (size_t)r;
    DLT(d, lt_shard_import(d->ctx, lo[r], hi[r], b + o_nodes, counts[r], b + o_keys));
  }
  if (n_keys_merged) *n_keys_merged = total;
  return LT_OK;
}

}  // extern "C"
