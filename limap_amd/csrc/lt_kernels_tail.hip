// lt_kernels_tail.hip -- device half of ComputeLineTracks (run_clustering, global_line_triangulator.cc:234-291).
//
// The reference walks valid_edges_ into a std::set of undirected node pairs and scores every pair with
// LineLinker3d::compute_score in spatial-merging mode between the two nodes' best candidates (:243-290; the two 2D
// scores it also computes are overwritten, :283).  The best candidates and the valid edges are still resident in
// HBM after the run, so the edge set (sorted unique 64-bit keys: the global node index is monotone in
// (img_id, line_id), i.e. the std::set order) and its similarities are built here, and only
//   sorted keys + similarities (16 B per directed valid edge) and the best candidates of the GRAPH NODES (128 B each)
// cross PCIe -- instead of 144 B for every node of the scene.  Union-find, labels and the aggregator stay on the
// host (serial, lt_api.cpp).
#include "lt_devfn.h"

#include <rocprim/device/device_radix_sort.hpp>

namespace lt {

// directed valid edges of node g (candidate order, like k_edge_fill) -> undirected keys (min << kb | max)
__global__ void __launch_bounds__(256)
k_tail_keys(long long G, const long long *__restrict__ tri_off, const unsigned *__restrict__ edge_flag,
            const long long *__restrict__ edge_off, const CRec *__restrict__ cand,
            const long long *__restrict__ seg_off, int kb, unsigned long long *__restrict__ keys,
            const unsigned *__restrict__ perm, int directed) {
  // directed: (source node << kb | target node) -- what a shard ships when the node filter is on (k_outer_*_keys below
  // count a node's edges into the kept set from them; k_keys_undirect then brings them to the form above)
  const long long g = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (g >= G) return;
  const int lane = lane_id();
  const long long off = tri_off[g];
  const int n = (int)(tri_off[g + 1] - off);
  long long base = edge_off[g];
  if (edge_off[g + 1] == base) return;
  for (int i0 = 0; i0 < n; i0 += 64) {
    const int i = i0 + lane;
    const bool f = (i < n) && edge_flag[off + i];
    const unsigned long long m = __ballot(f);
    if (f) {
      const int2 l = *reinterpret_cast<const int2 *>(&cand[perm ? (long long)perm[off + i] : off + i].nb_slot);
      const unsigned long long h = (unsigned long long)(seg_off[(int)((unsigned)l.x >> 8)] + (long long)l.y);
      const unsigned long long a = (unsigned long long)g < h ? (unsigned long long)g : h;
      const unsigned long long b = (unsigned long long)g < h ? h : (unsigned long long)g;
      keys[base + __popcll(m & lanemask_lt())] =
          directed ? (((unsigned long long)g << kb) | h) : ((a << kb) | b);  // kb = bits of a node index: fewer sort passes
    }
    base += __popcll(m);
  }
}

// filterNodeByNumOuterEdges (global_line_triangulator.cc:168-232): a node stays valid while at least min_outer of its valid
// edges lead to valid nodes.  The reference removes nodes through a queue; the result is the greatest fixed point of
// "every kept node keeps min_outer outgoing edges into the kept set", which any monotone removal order reaches: here every
// pass clears the nodes that fall short against the current flags (in place), until a pass changes nothing.
__global__ void __launch_bounds__(256)
k_outer_filter(long long G, const long long *__restrict__ tri_off, const unsigned *__restrict__ edge_flag,
               const CRec *__restrict__ cand, const long long *__restrict__ seg_off, const unsigned *__restrict__ perm,
               int min_outer, unsigned char *flags, int *__restrict__ changed) {
  const long long g = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (g >= G) return;
  if (!flags[g]) return;
  const int lane = lane_id();
  const long long off = tri_off[g];
  const int n = (int)(tri_off[g + 1] - off);
  int cnt = 0;
  for (int i0 = 0; i0 < n && cnt < min_outer; i0 += 64) {
    const int i = i0 + lane;
    bool f = (i < n) && edge_flag[off + i];
    if (f) {
      const int2 l = *reinterpret_cast<const int2 *>(&cand[perm ? (long long)perm[off + i] : off + i].nb_slot);
      const long long h = seg_off[(int)((unsigned)l.x >> 8)] + (long long)l.y;
      f = flags[h] != 0;
    }
    cnt += (int)__popcll(__ballot(f));
  }
  if (cnt < min_outer && lane == 0) {
    flags[g] = 0;
    *changed = 1;
  }
}

// The same filter over a list of DIRECTED keys (source << kb | target) -- the merged tail of a multi-GPU job, where the valid
// edges of the other ranks' nodes exist on this device only as the keys their shards brought (round 6).  One pass = count,
// for every kept source, its edges into the kept set; then clear the nodes that fall short (and the counters, for the next
// pass).  Same greatest fixed point as k_outer_filter; a node without any key has no valid edge and falls at once.
__global__ void __launch_bounds__(256)
k_outer_count_keys(long long E, const unsigned long long *__restrict__ keys, int kb, const unsigned char *__restrict__ flags,
                   unsigned *__restrict__ counts) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= E) return;
  const unsigned long long k = keys[i];
  const unsigned long long a = k >> kb, b = k & ((1ull << kb) - 1ull);
  if (flags[a] && flags[b]) atomicAdd(&counts[a], 1u);
}
__global__ void __launch_bounds__(256)
k_outer_apply_keys(long long G, unsigned *__restrict__ counts, int min_outer, unsigned char *__restrict__ flags,
                   int *__restrict__ changed) {
  const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= G) return;
  const unsigned c = counts[g];
  counts[g] = 0u;
  if (flags[g] && c < (unsigned)min_outer) {
    flags[g] = 0;
    *changed = 1;
  }
}
__global__ void __launch_bounds__(256)
k_keys_undirect(long long E, unsigned long long *__restrict__ keys, int kb) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= E) return;
  const unsigned long long k = keys[i];
  const unsigned long long a = k >> kb, b = k & ((1ull << kb) - 1ull);
  keys[i] = a < b ? k : ((b << kb) | a);
}

// similarity of every distinct key (the first of a run of equal keys; the others get -1), and the nodes that enter
// the graph (score != 0, global_line_triangulator.cc:284-285)
__global__ void __launch_bounds__(256)
k_tail_sims(long long E, const unsigned long long *__restrict__ skeys, const int *__restrict__ n_tris,
            const Cand *__restrict__ best_c, LinkCfg3 cfg, int kb, double *__restrict__ sims,
            unsigned *__restrict__ mark, unsigned *__restrict__ keep, const unsigned char *__restrict__ flags) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i == E) keep[E] = 0u;  // the scan of the flags runs over E + 1 entries
  if (i >= E) return;
  const unsigned long long key = skeys[i];
  if (i > 0 && skeys[i - 1] == key) {
    sims[i] = -1.0;
    keep[i] = 0u;
    return;
  }
  const long long a = (long long)(key >> kb), b = (long long)(key & ((1ull << kb) - 1ull));
  double s = 0.0;
  // a node without any candidate holds a value-initialised TriTuple in the reference: its zero line scores 0; an edge
  // with a filtered endpoint (flags: k_outer_filter) is not in the reference's edge set at all (:243-261)
  if (n_tris[a] > 0 && n_tris[b] > 0 && (!flags || (flags[a] && flags[b]))) {
    const Cand ca = best_c[a];
    const Cand cb = best_c[b];
    L3 la{mk3(ca.s[0], ca.s[1], ca.s[2]), mk3(ca.e[0], ca.e[1], ca.e[2])};
    L3 lb{mk3(cb.s[0], cb.s[1], cb.s[2]), mk3(cb.e[0], cb.e[1], cb.e[2])};
    s = score3d(cfg, la, lb, ca.unc, cb.unc, ca.depth);
  }
  sims[i] = s;
  keep[i] = s != 0.0 ? 1u : 0u;
  if (s != 0.0) {
    mark[a] = 1u;
    mark[b] = 1u;
  }
}

// the graph's edges (distinct keys with a non-zero similarity), in key order, packed into (pair, sim) entries -- the
// destination may be page-locked host memory.  pair = (rank of node a among the graph's nodes) << 32 | (rank of node b):
// the positions k_tail_gather writes the two nodes' records to (npos: the scan of `mark`).  The host then builds the graph
// over tables of the graph's size (round 6; with global node ids its node map had one entry per node of the SCENE and every
// lookup missed the cache: 0.43 ms of config 3's tail).
__global__ void __launch_bounds__(256)
k_tail_compact(long long E, const unsigned long long *__restrict__ skeys, const double *__restrict__ sims,
               const unsigned *__restrict__ keep, const long long *__restrict__ kpos, double2 *__restrict__ out,
               long long *__restrict__ n_out, const long long *__restrict__ npos, int kb) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0) *n_out = kpos[E];
  if (i >= E || !keep[i]) return;
  const unsigned long long key = skeys[i];
  const unsigned long long a = (unsigned long long)npos[key >> kb], b = (unsigned long long)npos[key & ((1ull << kb) - 1ull)];
  out[kpos[i]] = double2{__longlong_as_double((long long)((a << 32) | b)), sims[i]};
}

// the graph nodes' results, packed in ascending node order: 128-byte records (best candidate 112 B, support score,
// source (image index, line)) + the node index of every record
struct TailRec {
  Cand c;
  double score;
  int src[2];
};
static_assert(sizeof(TailRec) == 128, "TailRec layout");

__global__ void __launch_bounds__(256)
k_tail_gather(long long G, const unsigned *__restrict__ mark, const long long *__restrict__ pos,
              const Cand *__restrict__ best_c, const double *__restrict__ best_score,
              const int *__restrict__ best_src2, TailRec *__restrict__ recs, int *__restrict__ nodes,
              long long *__restrict__ n_out) {
  // recs / nodes / n_out may be page-locked HOST memory (the records then cross PCIe as the kernel writes them and
  // the host needs no size before the copy)
  if (n_out && blockIdx.x == 0 && threadIdx.x == 0) *n_out = pos[G];
  // 8 lanes per node: 16-byte units of the 128-byte record
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long g = t >> 3;
  const int u = (int)(t & 7);
  if (g >= G || !mark[g]) return;
  const long long p = pos[g];
  double2 v;
  if (u < 7) v = reinterpret_cast<const double2 *>(best_c + g)[u];
  else {
    v.x = best_score[g];
    int2 s = *reinterpret_cast<const int2 *>(best_src2 + 2 * g);
    v.y = __longlong_as_double(((long long)(unsigned)s.y << 32) | (long long)(unsigned)s.x);
  }
  reinterpret_cast<double2 *>(recs + p)[u] = v;
  if (u == 0) nodes[p] = (int)g;
}

size_t tail_rec_bytes() { return sizeof(TailRec); }

size_t tail_sort_temp_bytes(long long E, int end_bit) {
  size_t bytes = 0;
  (void)rocprim::radix_sort_keys(nullptr, bytes, (unsigned long long *)nullptr, (unsigned long long *)nullptr, (size_t)E, 0,
                                 end_bit, (hipStream_t)0);
  return bytes;
}

void launch_tail_keys(hipStream_t st, long long G, const long long *tri_off, const unsigned *edge_flag,
                      const long long *edge_off, const CRec *cand, const long long *seg_off, int kb,
                      unsigned long long *keys, const unsigned *perm, int directed) {
  if (G > 0)
    hipLaunchKernelGGL(k_tail_keys, dim3((unsigned)((G * 64 + 255) / 256)), dim3(256), 0, st, G, tri_off, edge_flag,
                       edge_off, cand, seg_off, kb, keys, perm, directed);
}
void launch_outer_pass_keys(hipStream_t st, long long E, const unsigned long long *keys, int kb, long long G, unsigned *counts,
                            int min_outer, unsigned char *flags, int *changed) {
  if (E > 0)
    hipLaunchKernelGGL(k_outer_count_keys, dim3((unsigned)((E + 255) / 256)), dim3(256), 0, st, E, keys, kb, flags, counts);
  if (G > 0)
    hipLaunchKernelGGL(k_outer_apply_keys, dim3((unsigned)((G + 255) / 256)), dim3(256), 0, st, G, counts, min_outer, flags, changed);
}
void launch_keys_undirect(hipStream_t st, long long E, unsigned long long *keys, int kb) {
  if (E > 0) hipLaunchKernelGGL(k_keys_undirect, dim3((unsigned)((E + 255) / 256)), dim3(256), 0, st, E, keys, kb);
}

int launch_tail_sort(hipStream_t st, void *temp, size_t temp_bytes, long long E, const unsigned long long *keys_in,
                     unsigned long long *keys_out, int end_bit) {
  if (E <= 0) return 0;
  return (int)rocprim::radix_sort_keys(temp, temp_bytes, keys_in, keys_out, (size_t)E, 0, end_bit, st);
}

void launch_tail_sims(hipStream_t st, long long E, const unsigned long long *skeys, const int *n_tris, const Cand *best_c,
                      const LinkCfg3 &cfg, int kb, double *sims, unsigned *mark, unsigned *keep, const unsigned char *flags) {
  if (E > 0)
    hipLaunchKernelGGL(k_tail_sims, dim3((unsigned)((E + 1 + 255) / 256)), dim3(256), 0, st, E, skeys, n_tris, best_c,
                       cfg, kb, sims, mark, keep, flags);
}
// keys that arrive from another rank (lt_shard_import): both node ids must be nodes of this scene and min < max as
// lt_shard_export writes them (directed keys -- node filter on -- are source << kb | target: any order, not equal) -- the
// similarity kernel indexes the per-node arrays with them
__global__ void k_check_keys(long long n, const unsigned long long *__restrict__ keys, int kb, long long G,
                             int *__restrict__ bad, int directed) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const unsigned long long k = keys[i];
  const unsigned long long a = k >> kb, b = k & ((1ull << kb) - 1ull);
  if (a >= (unsigned long long)G || b >= (unsigned long long)G || (directed ? a == b : a >= b)) *bad = 1;
}
void launch_check_keys(hipStream_t st, long long n, const unsigned long long *keys, int kb, long long G, int *bad, int directed) {
  if (n > 0) hipLaunchKernelGGL(k_check_keys, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n, keys, kb, G, bad, directed);
}
void launch_outer_filter(hipStream_t st, long long G, const long long *tri_off, const unsigned *edge_flag, const CRec *cand,
                         const long long *seg_off, const unsigned *perm, int min_outer, unsigned char *flags, int *changed) {
  if (G > 0)
    hipLaunchKernelGGL(k_outer_filter, dim3((unsigned)((G * 64 + 255) / 256)), dim3(256), 0, st, G, tri_off, edge_flag, cand,
                       seg_off, perm, min_outer, flags, changed);
}
void launch_tail_compact(hipStream_t st, long long E, const unsigned long long *skeys, const double *sims,
                         const unsigned *keep, const long long *kpos, void *out_pairs, long long *n_out, const long long *npos,
                         int kb) {
  if (E > 0)
    hipLaunchKernelGGL(k_tail_compact, dim3((unsigned)((E + 255) / 256)), dim3(256), 0, st, E, skeys, sims, keep, kpos,
                       reinterpret_cast<double2 *>(out_pairs), n_out, npos, kb);
}

void launch_tail_gather(hipStream_t st, long long G, const unsigned *mark, const long long *pos, const Cand *best_c,
                        const double *best_score, const int *best_src2, void *recs, int *nodes, long long *n_out) {
  if (G > 0)
    hipLaunchKernelGGL(k_tail_gather, dim3((unsigned)((G * 8 + 255) / 256)), dim3(256), 0, st, G, mark, pos, best_c,
                       best_score, best_src2, reinterpret_cast<TailRec *>(recs), nodes, n_out);
}

}  // namespace lt
