// lt_kernels.hip -- hand-written HIP kernels (gfx950 / CDNA4) of the line-triangulation path.
//
//   k_build_cams / k_build_segs / k_build_pairs : hoisted invariants (per image, per 2D segment,
//       per (image, neighbour) pair) -- the reference recomputes them per connection.
//   radix sort + k_node_offsets                 : matched mode, generic (stable) grouping of the
//       candidates by node when the rows of a block are not sorted by line id.
//   k_gates_ex / k_tri_ex / k_place_ex / k_fill_ex : HOT LOOP 1, triangulateOneNode (exhaustive mode)
//       (triangulation/base_line_triangulator.cc:161-337) for TriangulateImageExhaustiveMatch:
//       degeneracy gates, weak epipolar IoU, ray/plane triangulation, sensitivity gate, uncertainty, ranges.
//       One-pass form: cheap gates with the neighbour lines in registers -> cheirality / range pre-test -> entry
//       blocks -> dense evaluation into staging slots -> permutation; k_gates_ex<false> + k_fill_ex are the exact
//       two-pass fallback.  k_gen_ex_block / k_gen_exhaustive_pts: the same with VP-guided / point-guided proposals
//       (wave per (node, neighbour image)).  (Matched mode: k_gates + k_tri_rows in lt_kernels_v2.hip; scoring: k_score3.)
//   k_select                                    : per-node strict arg-max (lowest index wins ties,
//       global_line_triangulator.cc:145-153) and valid-edge flags (:118-142).
//
// Nothing here is a dense contraction, so there is no MFMA; the work is FP64 VALU + gathers that
// hit L2/MALL, and the rules that matter are coalescing, wave64 ballots/shuffles, LDS atomics.
// Compiled with -ffp-contract=off (see lt_geom.h).

#include "lt_devfn.h"

#include <algorithm>

#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/iterator/transform_iterator.hpp>

namespace lt {

// ---------------------------------------------------------------------------------------------
// invariants
// ---------------------------------------------------------------------------------------------
__global__ void k_build_cams(int n, const double *__restrict__ kvec, const double *__restrict__ qvec,
                             const double *__restrict__ tvec, Cam *__restrict__ cams) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Cam c;
  cam_build(kvec + 4 * i, qvec + 4 * i, tvec + 3 * i, &c);
  cams[i] = c;
}

__global__ void k_build_segs(long long n_segs, int n_img, const long long *__restrict__ seg_off,
                             const double *__restrict__ segs, double halfpix,
                             const Cam *__restrict__ cams, Seg *__restrict__ out, SegGate *__restrict__ gates) {
  long long s = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n_segs) return;
  // image of this segment: upper_bound(seg_off, s) - 1
  int lo = 0, hi = n_img;
  while (hi - lo > 1) {
    int mid = (lo + hi) >> 1;
    if (seg_off[mid] <= s) lo = mid; else hi = mid;
  }
  const double *p = segs + 4 * s;
  Seg r;
  // offsetHalfPixel (base_line_triangulator.cc:33-43): start + (0.5, 0.5) when add_halfpix
  double x1 = p[0], y1 = p[1], x2 = p[2], y2 = p[3];
  if (halfpix != 0.0) {
    x1 = x1 + halfpix; y1 = y1 + halfpix; x2 = x2 + halfpix; y2 = y2 + halfpix;
  }
  seg_build(cams[lo], x1, y1, x2, y2, &r);
  out[s] = r;
  if (gates) seg_gate_build(r, &gates[s]);  // stage A's per-segment record (k_gates), see lt_devfn.h
}

// Scene given as chunks (one per rank of the all-gather): chunk c holds images [img_begin[c],
// img_begin[c+1]) as kvec | qvec | tvec | segs in its own buffers.  Reads the gathered receive buffer
// in place -- no unpack copies on the per-step path of the multi-GPU job.
__global__ void k_build_cams_chunked(int n, int n_chunks, const SceneChunk *__restrict__ ch, Cam *__restrict__ cams) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int c = 0;
  while (c + 1 < n_chunks && i >= ch[c + 1].img_begin) ++c;
  int li = i - ch[c].img_begin;
  Cam cm;
  cam_build(ch[c].k + 4 * li, ch[c].q + 4 * li, ch[c].t + 3 * li, &cm);
  cams[i] = cm;
}

__global__ void k_build_segs_chunked(long long n_segs, int n_img, int n_chunks, const SceneChunk *__restrict__ ch,
                                     const long long *__restrict__ seg_off, double halfpix,
                                     const Cam *__restrict__ cams, Seg *__restrict__ out,
                                     SegGate *__restrict__ gates) {
  long long s = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n_segs) return;
  int lo = 0, hi = n_img;
  while (hi - lo > 1) {
    int mid = (lo + hi) >> 1;
    if (seg_off[mid] <= s) lo = mid; else hi = mid;
  }
  int c = 0;
  while (c + 1 < n_chunks && lo >= ch[c + 1].img_begin) ++c;
  const double *p = ch[c].s + 4 * (s - ch[c].seg_begin);
  double x1 = p[0], y1 = p[1], x2 = p[2], y2 = p[3];
  if (halfpix != 0.0) {
    x1 = x1 + halfpix; y1 = y1 + halfpix; x2 = x2 + halfpix; y2 = y2 + halfpix;
  }
  Seg r;
  seg_build(cams[lo], x1, y1, x2, y2, &r);
  out[s] = r;
  if (gates) seg_gate_build(r, &gates[s]);  // stage A's per-segment record (k_gates), see lt_devfn.h
}

// Same per image of a list (grid.y = list entry): no search for the segment's image, and images the
// buffered job does not reference (neither triangulated here nor a neighbour) can be left out -- with
// image shards spread over several GPUs a rank needs ~1/N of the gathered scene.
__global__ void k_build_segs_listed(const int *__restrict__ img_list, int n_chunks, const SceneChunk *__restrict__ ch,
                                    const long long *__restrict__ seg_off, double halfpix,
                                    Cam *__restrict__ cams, Seg *__restrict__ out,
                                    SegGate *__restrict__ gates) {
  const int img = img_list[blockIdx.y];
  const long long s0 = seg_off[img], M = seg_off[img + 1] - s0;
  const long long l = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  int c = 0;
  while (c + 1 < n_chunks && img >= ch[c + 1].img_begin) ++c;
  // the image's camera record is built here as well (every thread for itself, the first one stores it):
  // one launch per refresh instead of a camera kernel followed by a segment kernel
  Cam cm;
  {
    const int li = img - ch[c].img_begin;
    cam_build(ch[c].k + 4 * li, ch[c].q + 4 * li, ch[c].t + 3 * li, &cm);
    if (l == 0) cams[img] = cm;
  }
  if (l >= M) return;
  const long long s = s0 + l;
  const double *p = ch[c].s + 4 * (s - ch[c].seg_begin);
  double x1 = p[0], y1 = p[1], x2 = p[2], y2 = p[3];
  if (halfpix != 0.0) {
    x1 = x1 + halfpix; y1 = y1 + halfpix; x2 = x2 + halfpix; y2 = y2 + halfpix;
  }
  Seg r;
  seg_build(cm, x1, y1, x2, y2, &r);
  out[s] = r;
  if (gates) seg_gate_build(r, &gates[s]);
}

// also clears the run's two device scalars (error flag, pair statistic): one launch instead of three
__global__ void k_build_pairs(int n_blk, const int *__restrict__ blk_img,
                              const int *__restrict__ blk_nb, const Cam *__restrict__ cams,
                              PairRec *__restrict__ out, int *__restrict__ err_flag,
                              unsigned long long *__restrict__ pair_counter,
                              unsigned long long *__restrict__ scan_status, int n_status,
                              unsigned *__restrict__ blk_surv) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  for (int k = b; k < n_status; k += gridDim.x * blockDim.x) scan_status[k] = 0ull;  // k_node_prefix's look-back
  if (blk_surv && b < n_blk) blk_surv[b] = 0u;  // k_gates_ln's survivor cursors
  if (b == 0) {
    if (err_flag) *err_flag = 0;
    if (pair_counter) *pair_counter = 0ull;
  }
  if (b >= n_blk) return;
  PairRec p;
  pair_build(cams[blk_img[b]], cams[blk_nb[b]], &p);
  out[b] = p;
}

// conn_off[g] = first sorted position whose key >= g  (keys sorted ascending)
__global__ void k_node_offsets(long long P, long long G, const unsigned *__restrict__ skeys,
                               long long *__restrict__ conn_off) {
  long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t > P) return;
  long long a = (t == 0) ? -1 : (long long)skeys[t - 1];
  long long b = (t == P) ? G : (long long)skeys[t];
  if (a >= G) return;
  if (b > G) b = G;
  for (long long g = a + 1; g <= b; ++g) conn_off[g] = t;
}

// ---------------------------------------------------------------------------------------------
// HOT LOOP 1: one connection -> at most one candidate
// ---------------------------------------------------------------------------------------------
// Exhaustive mode (TriangulateImageExhaustiveMatch, base_line_triangulator.cc:111-136): the
// connection list is implicit.  Work item = (node, neighbour block, chunk of 64 ng lines);
// the wave writes its ballot of survivors as one 64-bit word (pass 1), and after a scan over the
// popcounts pass 2 re-derives the survivors and writes them at their final, ordered position.
// kVP: with VP-guided proposals (base_line_triangulator.cc:250-281) a connection yields up to three
// candidates in the order vp(l1), vp(l2), algebraic; the item then owns three ballots
// masks[3 * item + {0: algebraic, 1: vp(l1), 2: vp(l2)}].
// Exhaustive mode, one wave per (node, neighbour image), eight 64-line chunks of the neighbour at a time.
// Only ~4 % of the connections survive the gates, so with one chunk per wave nearly every chunk ran the
// triangulation (~900 instructions) for 2-3 active lanes.  Here phase A runs the cheap three-way gates over
// the eight chunks and compacts the connections that need more work (ballot + popcount) into an LDS list,
// phase B evaluates the list densely (exact gates where stage A could not decide, triangulation, VP
// proposals), phase C writes the eight survivor ballots.  kFill: the list is rebuilt from the ballots of
// pass 1 together with each candidate's output offset (items of a (node, neighbour) block are consecutive, so
// is their output), and phase B writes the candidates in place.
template <bool kFill, bool kVP>
__global__ void __launch_bounds__(256)
k_gen_ex_block(long long n_items, GenCfg cfg, const long long *__restrict__ item_off, long long G,
               const int *__restrict__ node_img, const long long *__restrict__ nb_off,
               const int *__restrict__ blk_nb, const long long *__restrict__ seg_off,
               const Cam *__restrict__ cams, const Seg *__restrict__ segs, const PairRec *__restrict__ pairs,
               unsigned long long *__restrict__ masks, const long long *__restrict__ mask_pos,
               CRec *__restrict__ out_r, double *__restrict__ out_unc, const double *__restrict__ seg_vp,
               const unsigned char *__restrict__ seg_has_vp, const int *__restrict__ blk_chunk_off, int max_nb,
               const SegGate *__restrict__ gates) {
  constexpr int kMasks = kVP ? 3 : 1;
  constexpr int kGroup = 8;
  constexpr int kCap = kGroup * 64 * kMasks;
  __shared__ unsigned s_list[4][kCap];
  __shared__ unsigned long long s_mask[4][kGroup * kMasks];
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lane = lane_id();
  unsigned *list = s_list[wv];
  unsigned long long *lmask = s_mask[wv];
  const long long w = (long long)blockIdx.x * 4 + wv;
  const long long g = w / max_nb;
  if (g >= G) return;
  const int k = (int)(w - g * max_nb);
  const int i1 = node_img[g];
  const long long b = nb_off[i1] + k;
  if (b >= nb_off[i1 + 1]) return;
  const int i2 = blk_nb[b];
  const long long g2base = seg_off[i2];
  const int M2 = (int)(seg_off[i2 + 1] - g2base);
  const int n_chunks = (M2 + 63) >> 6;
  const long long item0 = item_off[g] + blk_chunk_off[b];
  if (item0 >= n_items) return;
  const Seg &s1 = segs[g];
  const PairRec &pr = pairs[b];
  const int nbs = lite_pack(k, i2);
  bool len1_ok = true, has_vp1 = false;
  if (kVP) {
    L2 l1{mk2(s1.x1, s1.y1), mk2(s1.x2, s1.y2)};
    len1_ok = !(len(l1) <= cfg.min_length_2d);  // :166
    has_vp1 = seg_has_vp[g] != 0;
  }
  const unsigned long long lt_mask = lanemask_lt();
  for (int c0 = 0; c0 < n_chunks; c0 += kGroup) {
    const int nc = min(kGroup, n_chunks - c0);
    int n_ent = 0;
    if (!kFill && lane < kGroup * kMasks) lmask[lane] = 0ull;
    unsigned base_off = 0;
    // ---- phase A: which connections need phase B ----
    for (int cc = 0; cc < nc; ++cc) {
      const int ng = ((c0 + cc) << 6) + lane;
      const bool in_range = ng < M2;
      bool t0 = false, t1 = false, t2 = false, und = false;
      unsigned o0 = 0, o1 = 0, o2 = 0;
      if (!kFill) {
        if (in_range) {
          const SegGate gg = gates[g2base + ng];
          const int res = gate3(cfg, s1.x1, s1.y1, s1.x2, s1.y2, s1.rs[0], s1.rs[1], s1.rs[2], s1.re[0], s1.re[1], s1.re[2],
                                gg.n[0], gg.n[1], gg.n[2], gg.lcx, gg.lcy, gg.P, gg.Q, gg.w1, gg.sv, gg.q2, pr.F);
          t0 = res != 0;
          und = res == 2;
          if (kVP && len1_ok) {
            const Seg &s2 = segs[g2base + ng];
            L2 l2{mk2(s2.x1, s2.y1), mk2(s2.x2, s2.y2)};
            const bool len_ok = !(len(l2) <= cfg.min_length_2d);  // :177
            t1 = len_ok && has_vp1;
            t2 = len_ok && seg_has_vp[g2base + ng] != 0;
          }
        }
      } else {
        const long long item = item0 + c0 + cc;
        const unsigned long long m = masks[kMasks * item];
        unsigned long long m1 = 0, m2 = 0;
        if (kVP) {
          m1 = masks[kMasks * item + 1];
          m2 = masks[kMasks * item + 2];
        }
        t0 = (m >> lane) & 1ull;
        t1 = (m1 >> lane) & 1ull;
        t2 = (m2 >> lane) & 1ull;
        // output order inside an item: lanes ascending, per lane vp(l1), vp(l2), algebraic
        const unsigned lane_off = base_off + (unsigned)(__popcll(m & lt_mask) + __popcll(m1 & lt_mask) + __popcll(m2 & lt_mask));
        o1 = lane_off;
        o2 = lane_off + (t1 ? 1u : 0u);
        o0 = o2 + (t2 ? 1u : 0u);
        base_off += (unsigned)(__popcll(m) + __popcll(m1) + __popcll(m2));
      }
      const unsigned idx = (unsigned)((cc << 6) | lane);
      {
        const unsigned long long bm = __ballot(t0);
        if (t0) list[n_ent + __popcll(bm & lt_mask)] = idx | (0u << 9) | (und ? 1u << 11 : 0u) | (o0 << 12);
        n_ent += __popcll(bm);
      }
      if (kVP) {
        const unsigned long long b1 = __ballot(t1);
        if (t1) list[n_ent + __popcll(b1 & lt_mask)] = idx | (1u << 9) | (o1 << 12);
        n_ent += __popcll(b1);
        const unsigned long long b2 = __ballot(t2);
        if (t2) list[n_ent + __popcll(b2 & lt_mask)] = idx | (2u << 9) | (o2 << 12);
        n_ent += __popcll(b2);
      }
    }
    wave_lds_sync();
    // ---- phase B: dense evaluation ----
    const long long pos0 = kFill ? mask_pos[item0 + c0] : 0;
    for (int e0 = 0; e0 < n_ent; e0 += 64) {
      const int e = e0 + lane;
      if (e < n_ent) {
        const unsigned ent = list[e];
        const int cc = (int)((ent >> 6) & 7u), ln = (int)(ent & 63u), kind = (int)((ent >> 9) & 3u);
        const int ng = ((c0 + cc) << 6) + ln;
        const Seg &s2 = segs[g2base + ng];
        GenOut o;
        bool ok;
        if (kind == 0) {
          // ONE inlined gen_finish for both kinds of entry (pass 1 proved the gates / the cheap gates could not decide):
          // as two call sites a wave with mixed lanes ran the function twice
          bool pass = true;
          if (!kFill && ((ent >> 11) & 1u)) pass = gen_gates(cfg, s1, s2, pr.F);
          ok = pass && gen_finish(cfg, cams[i1], cams[i2], s1, s2, pr.B, &o);
        } else {
          ok = vp_candidate(cfg, cams[i1], cams[i2], s1, s2, pr.B, seg_vp + 3 * (kind == 1 ? g : g2base + ng), &o);
        }
        if (!kFill) {
          if (ok) atomicOr(&lmask[cc * kMasks + kind], 1ull << ln);
        } else if (ok) {
          const long long pos = pos0 + (long long)(ent >> 12);
          o.r.nb_slot = nbs;
          o.r.ng_line = ng;
          out_r[pos] = o.r;
          out_unc[pos] = o.unc;
        }
      }
    }
    wave_lds_sync();
    // ---- phase C: the survivor ballots of the group ----
    if (!kFill && lane < nc * kMasks) {
      const int cc = lane / kMasks, kind = lane - cc * kMasks;
      masks[kMasks * (item0 + c0 + cc) + kind] = lmask[cc * kMasks + kind];
    }
    wave_lds_sync();
  }
}

// Pass 1 of the exhaustive mode without VP / point proposals, with the loops swapped: one wave per (image pair,
// 64-line chunk of the NEIGHBOUR image, quarter of the image's nodes).  Every lane keeps the gate operands of its
// neighbour line in registers for the whole wave; the node side is wave-uniform -- scalar loads of the segment
// record, and the epipolar lines of its two endpoints (gate3_epi) are computed once per node and image pair (64
// endpoints per pass, all lanes busy) and broadcast through LDS.  Only ~4 % of the connections survive the cheap
// gates: the connections that need the dense evaluation (exact gates where the cheap ones could not decide,
// triangulation: ~1000 instructions) are appended to an LDS list ACROSS nodes and evaluated 64 at a time, so a dense
// round runs with every lane active (k_gen_ex_block evaluates the ~20 entries of its eight chunks per round: 31 %).
// The survivor ballots of up to kExSeg nodes are collected in LDS and written once -- the same words, at the same
// item indices, as k_gen_ex_block<false, false> writes.
constexpr int kExParts = 4;
constexpr int kExRegions = 16;  // staging regions of the one-pass form (one bump counter each)
constexpr int kExSeg = 128;  // nodes per segment (ballots held in LDS)
constexpr int kExSub = 32;   // nodes per epipolar-line pass
// kList (one-pass form): the kernel stops after the cheap gates -- the listed connections go, 64 at a time, into
// blocks of a global entry list (kExRegions regions with one bump counter each, 128 bytes apart, so that the ~4e5
// draws of a scene do not serialise; the last block of a wave is padded with ~0) and k_tri_ex evaluates them, one
// block per wave: this kernel then needs no candidate registers and keeps 4 waves per SIMD.  A full region raises
// error flag 5: the host repeats the run in the two-pass form.
// entry: ng line | node index within its image << 16 | undecided << 32 | (image, neighbour) block << 33
template <bool kList>
__global__ void __launch_bounds__(256)
k_gates_ex(int n_blk, int max_chunks, long long n_items, GenCfg cfg, const long long *__restrict__ item_off,
           const int *__restrict__ blk_img, const int *__restrict__ blk_nb, const long long *__restrict__ seg_off,
           const Cam *__restrict__ cams, const Seg *__restrict__ segs, const PairRec *__restrict__ pairs,
           unsigned long long *__restrict__ masks, const int *__restrict__ blk_chunk_off,
           const SegGate *__restrict__ gates, unsigned long long *__restrict__ ent_out,
           unsigned long long *__restrict__ ctr, unsigned region_cap, int *__restrict__ err_flag) {
  __shared__ unsigned s_list[4][128];
  __shared__ unsigned s_list2[4][kList ? 128 : 1];
  __shared__ unsigned long long s_mask[4][kList ? 1 : kExSeg];
  __shared__ double s_epi[4][kExSub * 12];  // per node: (ax, ay, az, n2a, na, q1) of the start point, same of the end point
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lane = lane_id();
  unsigned *list = s_list[wv];
  unsigned *list2 = s_list2[wv];
  unsigned long long *lmask = s_mask[wv];
  double *epi = s_epi[wv];
  const long long w = (long long)blockIdx.x * 4 + wv;
  const int part = (int)(w % kExParts);
  const long long r = w / kExParts;
  const int c = (int)(r % max_chunks);
  const long long b = r / max_chunks;
  if (b >= n_blk) return;
  const int i1 = blk_img[b], i2 = blk_nb[b];
  const long long g2base = seg_off[i2];
  const int M2 = (int)(seg_off[i2 + 1] - g2base);
  if ((c << 6) >= M2) return;
  const long long g1base = seg_off[i1];
  const int M1 = (int)(seg_off[i1 + 1] - g1base);
  const int n_lo = (int)((long long)M1 * part / kExParts), n_hi = (int)((long long)M1 * (part + 1) / kExParts);
  const int ng = (c << 6) + lane;
  const bool in_range = ng < M2;
  const SegGate gg = gates[g2base + (in_range ? ng : M2 - 1)];
  const PairRec &pr = pairs[b];
  const int chunk_off = blk_chunk_off[b] + c;
  const unsigned long long lt_mask = lanemask_lt();
  const int region = (int)((b + c) & (kExRegions - 1));  // every region sees every chunk index and a spread of image pairs
  int n_ent = 0, n_ent2 = 0;
  auto flush2 = [&]() {  // kList: (up to) 64 entries of the second list -> one block of the global entry list
    wave_lds_sync();
    const int base2 = max(n_ent2 - 64, 0);
    unsigned long long first = 0ull;
    if (lane == 0) first = atomicAdd(&ctr[region * 16], 64ull);
    first = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(first >> 32)) << 32) |
            (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(first & 0xFFFFFFFFull));
    unsigned long long e = ~0ull;
    if (base2 + lane < n_ent2) {
      const unsigned e2 = list2[base2 + lane];
      e = (unsigned long long)(unsigned)((c << 6) | (int)(e2 & 63u)) | ((unsigned long long)(e2 >> 7) << 16) |
          ((unsigned long long)((e2 >> 6) & 1u) << 32) | ((unsigned long long)b << 33);
    }
    if (first + 64ull <= (unsigned long long)region_cap) ent_out[(size_t)region * region_cap + (size_t)first + lane] = e;
    else *err_flag = 5;
    n_ent2 = base2;
    wave_lds_sync();
  };
  for (int seg0 = n_lo; seg0 < n_hi; seg0 += kExSeg) {
    const int ns = min(kExSeg, n_hi - seg0);
    const bool last_seg = seg0 + kExSeg >= n_hi;
    if (!kList)
      for (int l = lane; l < ns; l += 64) lmask[l] = 0ull;
    for (int sub0 = 0; sub0 < ns; sub0 += kExSub) {
      const int nsub = min(kExSub, ns - sub0);
      const bool last_sub = sub0 + kExSub >= ns;
      wave_lds_sync();  // the previous pass' epipolar lines are no longer read
      if (lane < 2 * nsub) {
        const Seg &s = segs[g1base + seg0 + sub0 + (lane >> 1)];
        const bool second = (lane & 1) != 0;
        const GateEpi e = gate3_epi(pr.F, second ? s.x2 : s.x1, second ? s.y2 : s.y1);
        const double d1x = s.x1 - s.x2, d1y = s.y1 - s.y2;
        double *o = epi + 6 * lane;
        o[0] = e.ax; o[1] = e.ay; o[2] = e.az; o[3] = e.n2a; o[4] = e.na;
        o[5] = __builtin_fma(d1x, d1x, d1y * d1y);
      }
      wave_lds_sync();
      for (int cc = 0;; ++cc) {
        const bool more = cc < nsub;
        if (more) {
          // ---- phase A: does this connection need the dense evaluation ----
          // (two nodes per iteration, for two independent dependency chains per lane: measured, no change)
          const Seg &s1 = segs[g1base + seg0 + sub0 + cc];
          const double *o = epi + 12 * cc;
          GateEpi ea, eb;
          ea.ax = o[0]; ea.ay = o[1]; ea.az = o[2]; ea.n2a = o[3]; ea.na = o[4];
          eb.ax = o[6]; eb.ay = o[7]; eb.az = o[8]; eb.n2a = o[9]; eb.na = o[10];
          const int res = gate3_core_fma(cfg, o[5], s1.rs[0], s1.rs[1], s1.rs[2], s1.re[0], s1.re[1], s1.re[2], gg.n[0],
                                     gg.n[1], gg.n[2], gg.lcx, gg.lcy, gg.P, gg.Q, gg.w1, gg.sv, gg.q2, ea, eb);
          const bool t0 = in_range && res != 0;
          const unsigned long long bm = __ballot(t0);
          if (t0)
            list[n_ent + __popcll(bm & lt_mask)] = (unsigned)(((seg0 + sub0 + cc) << 7) | lane) | (res == 2 ? 64u : 0u);
          n_ent += __popcll(bm);
        }
        // the list is emptied 64 entries at a time; the rest goes at the end of the segment (of the wave: kList)
        if (n_ent >= 64 || (!more && last_sub && (last_seg || !kList) && n_ent > 0)) {
          wave_lds_sync();
          const int base = max(n_ent - 64, 0);
          if (kList) {
            // ---- pre-test of (up to) 64 listed connections: the intersections, cheirality and the ranges reject about
            // half of what the cheap gates pass, for ~5 % of the cost of the full evaluation; the rest goes to the
            // second list, and from there in blocks of 64 to the global entry list ----
            bool keep = false;
            unsigned ent = 0u;
            if (base + lane < n_ent) {
              ent = list[base + lane];
              keep = gen_pretest(cfg, cams[i1], cams[i2], segs[g1base + (int)(ent >> 7)],
                                 segs[g2base + (c << 6) + (int)(ent & 63u)], pr.B);
            }
            const unsigned long long km = __ballot(keep);
            if (keep) list2[n_ent2 + __popcll(km & lt_mask)] = ent;
            n_ent2 += __popcll(km);
            n_ent = base;
            if (n_ent2 >= 64) flush2();
          } else if (base + lane < n_ent) {
            // ---- phase B: dense evaluation of (up to) 64 listed connections ----
            const unsigned ent = list[base + lane];
            const int nd = (int)(ent >> 7), ln = (int)(ent & 63u);
            const Seg &s1 = segs[g1base + nd];
            const Seg &s2 = segs[g2base + (c << 6) + ln];
            GenOut o;
            bool pass = true;  // one inlined gen_finish, see k_gen_list
            if (ent & 64u) pass = gen_gates(cfg, s1, s2, pr.F);
            const bool ok = pass && gen_finish(cfg, cams[i1], cams[i2], s1, s2, pr.B, &o);
            if (ok) atomicOr(&lmask[nd - seg0], 1ull << ln);
          }
          n_ent = base;
          wave_lds_sync();
        }
        if (!more) break;
      }
    }
    if (!kList) {
      wave_lds_sync();
      // ---- phase C: the survivor ballots of the segment ----
      for (int l = lane; l < ns; l += 64) {
        const long long item = item_off[g1base + seg0 + l] + chunk_off;
        if (item < n_items) masks[item] = lmask[l];
      }
      wave_lds_sync();
    }
  }
  if (kList && n_ent2 > 0) flush2();
}

// One-pass exhaustive mode, dense evaluation: one wave per block of 64 listed connections (all of one image pair).
// Exact gates where the cheap ones could not decide, triangulation.  The survivors of a block (78 % of its entries on
// the bench scene: 2.2e7 of 2.85e7) are COMPACTED to the front of the block's 64 staging slots, in lane order: record,
// uncertainty, depth key, node and -- in place -- the 8-byte entry k_place_ex decodes; the slots behind them are holes
// (node = ~0).  Round 5: the kernel used to store the whole 8 KB record row of every block that had a survivor, holes
// included: 4.75 -> 4.0 GB of writes per launch, 1.82 -> 1.58 ms.  The kernel is bound by those stores (2.2e7 records
// of 128 bytes + 20 bytes of keys each; 3.2 TB/s of mixed traffic, VALU issue 0.28; without its stores: 0.8 ms).
// Measured and not kept: the staging arrays dense across blocks -- one cursor per region serialises its 3e4 atomics at
// ~100 ns each (+1.3 ms), 64 cursors per region: +0.1 ms against this form; the kernel persistent (grid = resident
// set): no change; persistent with the next block's entries and image ids loaded ahead: 187 registers (+0.1 ms),
// capped at 128: 45 spills (+0.26 ms); entry and uncertainty compacted by shuffles instead of LDS (32 KB: five
// workgroups to a CU instead of four): +0.15 ms.  A survivor also sets its bit in the ballot word of its work item
// (masks zeroed beforehand) -- from there on the counts, offsets and the permutation are those of the two-pass form.
__global__ void __launch_bounds__(256)
k_tri_ex(unsigned long long *__restrict__ ent, const unsigned long long *__restrict__ ctr, unsigned region_cap,
         GenCfg cfg, long long n_items, const long long *__restrict__ item_off, const int *__restrict__ blk_img,
         const int *__restrict__ blk_nb, const long long *__restrict__ nb_off, const long long *__restrict__ seg_off,
         const Cam *__restrict__ cams, const Seg *__restrict__ segs, const PairRec *__restrict__ pairs,
         const int *__restrict__ blk_chunk_off, unsigned long long *__restrict__ masks, CRec *__restrict__ st_r,
         double *__restrict__ st_unc, unsigned *__restrict__ st_node, float *__restrict__ st_z) {
  // The records go through LDS and leave as full 128-byte lines, 64 lanes on consecutive 16-byte pieces (a lane
  // storing its own 128-byte record writes 16-byte pieces 128 bytes apart -- every store instruction then touches 64
  // cache lines).
  __shared__ double2 s_out[4][64 * 8];
  __shared__ unsigned long long s_ent[4][64];
  __shared__ double s_unc[4][64];
  __shared__ unsigned s_node[4][64];
  __shared__ float s_z[4][64];
  const int region = blockIdx.y;
  const unsigned long long n = ctr[region * 16];
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const unsigned long long t0 = ((unsigned long long)blockIdx.x * 4 + (unsigned)wv) * 64ull;
  if (t0 >= n || t0 + 64ull > (unsigned long long)region_cap) return;
  const int lane = lane_id();
  const size_t slot0 = (size_t)region * region_cap + (size_t)t0;
  const size_t slot = slot0 + lane;
  const unsigned long long e = ent[slot];
  // lane 0 of a block always holds an entry, and every entry of a block comes from the same wave of k_gates_ex
  const unsigned e_hi0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(e >> 32));
  const long long b = (long long)(e_hi0 >> 1);
  const int i1 = blk_img[b], i2 = blk_nb[b];
  bool ok = false;
  GenOut o;
  long long g = 0;
  int ng = 0;
  if (e != ~0ull) {
    ng = (int)(e & 0xFFFFu);
    const int nd = (int)((e >> 16) & 0xFFFFu);
    g = seg_off[i1] + nd;
    const Seg &s1 = segs[g];
    const Seg &s2 = segs[seg_off[i2] + ng];
    const PairRec &pr = pairs[b];
    // the exact gates only where the cheap ones could not decide, then ONE inlined gen_finish for every lane (as
    // two call sites a wave with mixed lanes ran gen_finish twice)
    bool pass = true;
    if ((e >> 32) & 1ull) pass = gen_gates(cfg, s1, s2, pr.F);
    ok = pass && gen_finish(cfg, cams[i1], cams[i2], s1, s2, pr.B, &o);
  }
  const unsigned long long okm = __ballot(ok);
  const int cnt = __popcll(okm);
  if (ok) {
    const int rank = __popcll(okm & lanemask_lt());
    o.r.nb_slot = lite_pack((int)(b - nb_off[i1]), i2);
    o.r.ng_line = ng;
    *reinterpret_cast<CRec *>(&s_out[wv][8 * rank]) = o.r;
    s_ent[wv][rank] = e;
    s_unc[wv][rank] = o.unc;
    s_node[wv][rank] = (unsigned)g;
    // the depth-order keys of the scoring stage (single-precision start depth), 4 bytes per slot: k_depth_order then
    // gathers from a 0.1 GB array that stays in the last-level cache instead of one 112-byte record per key
    s_z[wv][rank] = (float)o.r.depth[0];
    const long long item = item_off[g] + blk_chunk_off[b] + (ng >> 6);
    if (item < n_items) atomicOr(&masks[item], 1ull << (ng & 63));
  }
  if (cnt == 0) {
    st_node[slot] = 0xFFFFFFFFu;
    return;
  }
  wave_lds_sync();
  st_node[slot] = lane < cnt ? s_node[wv][lane] : 0xFFFFFFFFu;
  if (lane < cnt) {
    ent[slot] = s_ent[wv][lane];
    st_unc[slot] = s_unc[wv][lane];
    st_z[slot] = s_z[wv][lane];
  }
  double2 *dc = reinterpret_cast<double2 *>(st_r + slot0);
#pragma unroll
  for (int k = 0; k < 8; ++k)
    if (k * 64 + lane < cnt * 8) dc[k * 64 + lane] = s_out[wv][k * 64 + lane];
}

// Pass 2 of the plain exhaustive mode: one wave per (image pair, eighth of the image's nodes).  The survivors of
// pass 1 (ballots masks[item], output offsets mask_pos[item]) are expanded into an LDS list across chunks and nodes
// and triangulated 64 at a time with every lane active (a (node, neighbour) block has ~20 survivors: the
// wave-per-block form ran its rounds at 31 %); both cameras and the pair record stay wave-uniform.  Every candidate
// is written at mask_pos[item] + (survivors of the lower lanes): the order of base_line_triangulator.cc:111-136.
constexpr int kFillParts = 8;
__global__ void __launch_bounds__(256)
k_fill_ex(int n_blk, long long n_items, GenCfg cfg, const long long *__restrict__ item_off,
          const int *__restrict__ blk_img, const int *__restrict__ blk_nb, const long long *__restrict__ nb_off,
          const long long *__restrict__ seg_off, const Cam *__restrict__ cams, const Seg *__restrict__ segs,
          const PairRec *__restrict__ pairs, const unsigned long long *__restrict__ masks,
          const long long *__restrict__ mask_pos, CRec *__restrict__ out_r, double *__restrict__ out_unc,
          const int *__restrict__ blk_chunk_off) {
  __shared__ unsigned s_list[4][128];
  __shared__ long long s_pos[4][128];
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lane = lane_id();
  unsigned *list = s_list[wv];
  long long *lpos = s_pos[wv];
  const long long w = (long long)blockIdx.x * 4 + wv;
  const int part = (int)(w % kFillParts);
  const long long b = w / kFillParts;
  if (b >= n_blk) return;
  const int i1 = blk_img[b], i2 = blk_nb[b];
  const long long g2base = seg_off[i2];
  const int M2 = (int)(seg_off[i2 + 1] - g2base);
  const int n_chunks = (M2 + 63) >> 6;
  const long long g1base = seg_off[i1];
  const int M1 = (int)(seg_off[i1 + 1] - g1base);
  const int n_lo = (int)((long long)M1 * part / kFillParts), n_hi = (int)((long long)M1 * (part + 1) / kFillParts);
  const PairRec &pr = pairs[b];
  const int nbs = lite_pack((int)(b - nb_off[i1]), i2);
  const int chunk_off = blk_chunk_off[b];
  const unsigned long long lt_mask = lanemask_lt();
  int n_ent = 0;
  auto dense = [&]() {
    wave_lds_sync();
    const int base = max(n_ent - 64, 0);
    if (base + lane < n_ent) {
      const unsigned ent = list[base + lane];
      const long long pos = lpos[base + lane];
      const int nd = (int)(ent >> 16), ng = (int)(ent & 0xFFFFu);
      const Seg &s1 = segs[g1base + nd];
      const Seg &s2 = segs[g2base + ng];
      GenOut o;
      if (gen_finish(cfg, cams[i1], cams[i2], s1, s2, pr.B, &o)) {  // pass 1 proved the gates
        o.r.nb_slot = nbs;
        o.r.ng_line = ng;
        out_r[pos] = o.r;
        out_unc[pos] = o.unc;
      }
    }
    n_ent = base;
    wave_lds_sync();
  };
  for (int nd = n_lo; nd < n_hi; ++nd) {
    const long long item0 = item_off[g1base + nd] + chunk_off;
    for (int c0 = 0; c0 < n_chunks; c0 += 64) {
      const int ncb = min(64, n_chunks - c0);
      unsigned long long m_l = 0ull;
      long long p_l = 0;
      if (lane < ncb && item0 + c0 + lane < n_items) {
        m_l = masks[item0 + c0 + lane];
        p_l = mask_pos[item0 + c0 + lane];
      }
      unsigned long long nonempty = __ballot(m_l != 0ull);
      while (nonempty) {
        const int cc = __builtin_ctzll(nonempty);
        nonempty &= nonempty - 1ull;
        const unsigned long long m =
            ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)(m_l >> 32), cc) << 32) |
            (unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)(m_l & 0xFFFFFFFFull), cc);
        const long long p0 =
            (long long)(((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)((unsigned long long)p_l >> 32), cc) << 32) |
                        (unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)((unsigned long long)p_l & 0xFFFFFFFFull), cc));
        if ((m >> lane) & 1ull) {
          const int rank = __popcll(m & lt_mask);
          list[n_ent + rank] = ((unsigned)nd << 16) | (unsigned)(((c0 + cc) << 6) | lane);
          lpos[n_ent + rank] = p0 + rank;
        }
        n_ent += __popcll(m);
        if (n_ent >= 64) dense();
      }
    }
  }
  if (n_ent > 0) dense();
}

// Exhaustive mode with the point-guided proposals (and, if set, the VP ones): per connection a variable
// number of candidates in the reference's order many-points, one-point (one per shared point, ascending
// point3D_id), vp(l1), vp(l2), algebraic (base_line_triangulator.cc:183-325).  Pass 1 (kFill == false)
// counts them per connection (16 bits each: no practical limit on the shared points) and per work item; pass 2 recomputes them and writes
// every valid one at  mask_pos[item] + (candidates of the lower lanes) + (rank within the connection).
// This is the configuration of the reference's third CI run (exhaustive matcher + use_pointsfm).
template <bool kFill>
__global__ void __launch_bounds__(256)
k_gen_exhaustive_pts(long long n_items, GenCfg cfg, const long long *__restrict__ item_off, long long G,
                     const int *__restrict__ node_img, const long long *__restrict__ nb_off,
                     const int *__restrict__ blk_nb, const long long *__restrict__ seg_off,
                     const Cam *__restrict__ cams, const Seg *__restrict__ segs,
                     const PairRec *__restrict__ pairs, unsigned short *__restrict__ cnt8,
                     unsigned *__restrict__ item_cnt, const long long *__restrict__ mask_pos,
                     CRec *__restrict__ out_r, double *__restrict__ out_unc, const double *__restrict__ seg_vp,
                     const unsigned char *__restrict__ seg_has_vp, const long long *__restrict__ seg_pt_off,
                     const SegPoint *__restrict__ seg_pts, const double *__restrict__ sfm_xyz,
                     int *__restrict__ err_flag, int many_on, int one_on, const int *__restrict__ blk_chunk_off,
                     int max_nb, int max_chunks, const SegGate *__restrict__ gates) {
  const long long w = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const long long per_node = (long long)max_nb * max_chunks;
  const long long g = w / per_node;
  if (g >= G) return;
  const int rest = (int)(w - g * per_node);
  const int k = rest / max_chunks;
  const long long rem = rest - k * max_chunks;
  const int i1 = node_img[g];
  const long long b = nb_off[i1] + k;
  if (b >= nb_off[i1 + 1]) return;
  const int i2 = blk_nb[b];
  const long long M2 = seg_off[i2 + 1] - seg_off[i2];
  if (rem >= ((M2 + 63) >> 6)) return;
  const long long item = item_off[g] + blk_chunk_off[b] + rem;
  if (item >= n_items) return;
  const int lane = lane_id();
  const int ng_line = (int)(rem << 6) + lane;
  const long long g2 = seg_off[i2] + ng_line;
  const bool in_range = ng_line < M2;
  unsigned cnt = 0;
  long long pos = 0;
  bool work = in_range;
  if (kFill) {
    const unsigned mine = cnt8[item * 64 + lane];
    unsigned incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const unsigned t = (unsigned)__shfl_up((int)incl, d);
      if (lane >= d) incl += t;
    }
    pos = mask_pos[item] + (long long)(incl - mine);
    work = work && mine != 0;
  }
  if (work) {
    const Seg &s1 = segs[g];
    const Seg &s2 = segs[g2];
    const int nbs = lite_pack((int)(b - nb_off[i1]), i2);
    auto emit = [&](GenOut &o) {
      if (kFill) {
        o.r.nb_slot = nbs;
        o.r.ng_line = ng_line;
        out_r[pos] = o.r;
        out_unc[pos] = o.unc;
        ++pos;
      }
      ++cnt;
    };
    L2 l1{mk2(s1.x1, s1.y1), mk2(s1.x2, s1.y2)};
    L2 l2{mk2(s2.x1, s2.y1), mk2(s2.x2, s2.y2)};
    const bool len_ok = !(len(l1) <= cfg.min_length_2d) && !(len(l2) <= cfg.min_length_2d);  // :166,177
    GenOut o;
    if (len_ok && seg_pts) {
      const long long pa0 = seg_pt_off[g], pb0 = seg_pt_off[g2];
      const SegPoint *pa = seg_pts + pa0, *pb = seg_pts + pb0;
      const int na = (int)(seg_pt_off[g + 1] - pa0), nb = (int)(seg_pt_off[g2 + 1] - pb0);
      if (na > 0 && nb > 0) {
        if (many_on) {
          bool missing = false;
          if (points_candidate(cfg, cams[i1], cams[i2], s1, s2, pa, na, pb, nb, sfm_xyz, &o, &missing)) emit(o);
          if (missing) *err_flag = 2;
        }
        if (one_on) {
          int i = 0, j = 0, idx = 0;
          while (i < na && j < nb) {
            const int ia = pa[i].p3d_id, ib = pb[j].p3d_id;
            if (ia < ib) { ++i; continue; }
            if (ib < ia) { ++j; continue; }
            d3 P = mk3(0, 0, 0);
            bool okp = true;
            if (sfm_xyz) {
              const int sidx = pa[i].sfm;
              if (sidx < 0) { *err_flag = 2; okp = false; }
              else P = mk3(sfm_xyz[3 * sidx], sfm_xyz[3 * sidx + 1], sfm_xyz[3 * sidx + 2]);
            } else {
              okp = tri_point(cams[i1], cam_ray(cams[i1], d2{pa[i].x, pa[i].y}), cams[i2],
                              cam_ray(cams[i2], d2{pb[j].x, pb[j].y}), &P);
            }
            if (okp) {
              if (idx >= 65000) { *err_flag = 3; break; }  // the per-connection count is 16 bits wide
              if (one_point_candidate(cfg, cams[i1], cams[i2], s1, s2, P, &o)) emit(o);
              ++idx;
            }
            ++i; ++j;
          }
        }
      }
    }
    if (len_ok && seg_vp) {
      if (seg_has_vp[g] && vp_candidate(cfg, cams[i1], cams[i2], s1, s2, pairs[b].B, seg_vp + 3 * g, &o)) emit(o);
      if (seg_has_vp[g2] && vp_candidate(cfg, cams[i1], cams[i2], s1, s2, pairs[b].B, seg_vp + 3 * g2, &o)) emit(o);
    }
    {
      const SegGate gg = gates[g2];
      const int res = gate3(cfg, s1.x1, s1.y1, s1.x2, s1.y2, s1.rs[0], s1.rs[1], s1.rs[2], s1.re[0], s1.re[1], s1.re[2],
                            gg.n[0], gg.n[1], gg.n[2], gg.lcx, gg.lcy, gg.P, gg.Q, gg.w1, gg.sv, gg.q2, pairs[b].F);
      bool ok = false;
      bool pass = res != 0;  // one inlined gen_finish for decided (1) and undecided (2) pairs
      if (res == 2) pass = gen_gates(cfg, s1, s2, pairs[b].F);
      if (pass) ok = gen_finish(cfg, cams[i1], cams[i2], s1, s2, pairs[b].B, &o);
      if (ok) emit(o);
    }
  }
  if (!kFill) {
    cnt8[item * 64 + lane] = (unsigned short)cnt;
    unsigned tot = cnt;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) tot += (unsigned)__shfl_xor((int)tot, d);
    if (lane == 0) item_cnt[item] = tot;
  }
}

__global__ void k_popc(long long n, const unsigned long long *__restrict__ masks,
                       unsigned *__restrict__ cnt, int n_masks) {
  long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  unsigned c = 0;
  for (int k = 0; k < n_masks; ++k) c += (unsigned)__popcll(masks[(long long)n_masks * t + k]);
  cnt[t] = c;
}

// tri_off[g] = mask_pos[item_off[g]].  total < 0: the candidate count stays on the device (mask_pos[n_items]);
// cap >= 0: the staging capacity of the one-pass form -- a count beyond it leaves every node empty and raises
// error flag 5 (the host repeats the run in the two-pass form).
__global__ void k_tri_offsets_ex(long long G, const long long *__restrict__ item_off,
                                 const long long *__restrict__ mask_pos, long long n_items,
                                 long long total, long long cap, long long *__restrict__ tri_off,
                                 int *__restrict__ err_flag) {
  long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (g > G) return;
  if (total < 0) total = mask_pos[n_items];
  if (cap >= 0 && (total > cap || *err_flag == 5)) {  // 5: a staging region of k_gates_ex was full
    tri_off[g] = 0;
    if (g == 0) *err_flag = 5;
    return;
  }
  long long it = item_off[g];
  tri_off[g] = (it >= n_items) ? total : mask_pos[it];
}

// One-pass exhaustive mode: the final position of the candidate in staging slot s -- its work item is
// (node, neighbour block, chunk of its neighbour line), its rank the survivors of the lower lanes of that item.
// Block and neighbour line come from the slot's 8-byte entry (a coalesced stream), not from the 128-byte record:
// two ints out of every record cost one cache line per slot, 3.6 GB of the kernel's 4.5 GB.
__global__ void __launch_bounds__(256)
k_place_ex(const unsigned long long *__restrict__ ctr, unsigned region_cap, const unsigned long long *__restrict__ ent,
           const unsigned *__restrict__ st_node, const long long *__restrict__ item_off,
           const int *__restrict__ blk_chunk_off, const unsigned long long *__restrict__ masks,
           const long long *__restrict__ mask_pos, long long n_items, const long long *__restrict__ tri_off,
           long long G, unsigned *__restrict__ perm, long long *__restrict__ fill_out) {
  const int region = blockIdx.y;
  const unsigned long long n = ctr[region * 16];
  const unsigned long long t = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t == 0 && region == 0) {  // the fullest region: what the host sizes the next run's staging by
    unsigned long long mx = 0ull;
    for (int r = 0; r < kExRegions; ++r) mx = ctr[r * 16] > mx ? ctr[r * 16] : mx;
    *fill_out = (long long)mx;
  }
  if (t >= n || t >= (unsigned long long)region_cap) return;
  if (tri_off[G] == 0) return;  // overflow: nothing is placed
  const size_t slot = (size_t)region * region_cap + (size_t)t;
  const unsigned gu = st_node[slot];
  if (gu == 0xFFFFFFFFu) return;  // a connection that failed the dense evaluation
  const unsigned long long e = ent[slot];
  const int ng = (int)(e & 0xFFFFu);
  const long long b = (long long)(e >> 33);
  const long long item = item_off[(long long)gu] + blk_chunk_off[b] + (ng >> 6);
  if (item >= n_items) return;
  const unsigned long long m = masks[item];
  const long long pos = mask_pos[item] + __popcll(m & ((1ull << (ng & 63)) - 1ull));
  perm[pos] = (unsigned)slot;
}

// ---------------------------------------------------------------------------------------------
// per-node selection: best candidate = first strict maximum; valid-edge flags
// ---------------------------------------------------------------------------------------------
template <int kLanes>  // lanes per node: 16 (matched mode, ~12 candidates per node) or 64 (exhaustive, hundreds)
__global__ void __launch_bounds__(256)
k_select(long long G, const long long *__restrict__ tri_off, const double *__restrict__ score,
         double fullscore_th, int max_valid_conns, long long *__restrict__ best_idx,
         unsigned *__restrict__ edge_flag, unsigned *__restrict__ n_valid, const CRec *__restrict__ cand,
         const double *__restrict__ cand_unc, Cand *__restrict__ best_c, double *__restrict__ best_score,
         int *__restrict__ best_src2, int *__restrict__ n_tris, const int *__restrict__ err_flag,
         const unsigned long long *__restrict__ pair_counter, long long *__restrict__ result3,
         const unsigned *__restrict__ perm) {
  // the run's three result scalars (error flag, candidate count, pair statistic) are gathered into one record
  // here, in the last kernel of the run, so that one 24-byte copy brings them to the host instead of three
  if (result3 && blockIdx.x == 0 && threadIdx.x == 0) {
    result3[0] = tri_off[G];
    result3[1] = (long long)*err_flag;
    result3[2] = (long long)*pair_counter;
  }
  // kLanes = 16: a quarter wave per node -- a whole wave per node left 4/5 of the lanes idle in matched mode;
  // xor-shuffles below kLanes stay inside the group
  long long g = ((long long)blockIdx.x * blockDim.x + threadIdx.x) / kLanes;
  if (g >= G) return;
  const int lane = lane_id() & (kLanes - 1);
  const long long off = tri_off[g];
  const int n = (int)(tri_off[g + 1] - off);
  double bs = -1.0;
  int bi = -1;
  int n_full = 0;
  for (int i = lane; i < n; i += kLanes) {
    double s = score[off + i];
    if (s > bs) {  // ascending i inside a lane: first strict max
      bs = s;
      bi = i;
    }
    if (s >= fullscore_th) ++n_full;
  }
  for (int d = kLanes / 2; d >= 1; d >>= 1) {
    double os = __shfl_xor(bs, d);
    int oi = __shfl_xor(bi, d);
    n_full += __shfl_xor(n_full, d);
    bool take = (oi >= 0) && (bi < 0 || os > bs || (os == bs && oi < bi));
    if (take) {
      bs = os;
      bi = oi;
    }
  }
  if (lane == 0) best_idx[g] = (bi < 0) ? -1 : off + bi;
  // the best candidate's record goes to the dense per-node arrays right here (lanes 0-6: the 7 16-byte
  // units of the Cand, lane 7: score, lane 8: source (image, line), lane 9: candidate count)
  {
    static_assert(sizeof(Cand) == 7 * 16 && sizeof(CRec) == 8 * 16, "records in 16-byte units");
    long long b = (bi < 0) ? -1 : off + bi;
    if (perm && b >= 0) b = (long long)perm[b];  // records still in the staging lists (k_place wrote the permutation)
    if (lane < 7) {
      // Cand units: 0-3 = s, e, depth (CRec units 0-3), 4 = (unc, score3 = 1), 5-6 = seg (CRec units 4-5)
      double2 v = double2{0.0, 0.0};
      if (b >= 0) {
        if (lane == 4) v = double2{cand_unc[b], 1.0};
        else v = reinterpret_cast<const double2 *>(cand + b)[lane < 4 ? lane : lane - 1];
      }
      reinterpret_cast<double2 *>(best_c + g)[lane] = v;
    } else if (lane == 7) {
      best_score[g] = (b >= 0) ? bs : 0.0;
    } else if (lane == 8) {
      int src_img = -1, src_line = -1;
      if (b >= 0) {
        const int2 l = *reinterpret_cast<const int2 *>(&cand[b].nb_slot);
        src_img = (int)((unsigned)l.x >> 8);
        src_line = l.y;
      }
      best_src2[2 * g] = src_img;
      best_src2[2 * g + 1] = src_line;
    } else if (lane == 9) {
      n_tris[g] = n;
    }
  }
  // valid edges: the max_valid_conns best by (score, tri_id) descending, kept if score >= th
  const bool need_rank = n_full > max_valid_conns;
  int kept = 0;
  for (int i = lane; i < n; i += kLanes) {
    double s = score[off + i];
    unsigned f = s >= fullscore_th ? 1u : 0u;
    if (f && need_rank) {
      int rank = 0;
      for (int j = 0; j < n; ++j) {
        double sj = score[off + j];
        if (sj > s || (sj == s && j > i)) ++rank;
      }
      if (rank >= max_valid_conns) f = 0;
    }
    edge_flag[off + i] = f;
    kept += (int)f;
  }
  for (int d = kLanes / 2; d >= 1; d >>= 1) kept += __shfl_xor(kept, d);
  if (lane == 0) n_valid[g] = (unsigned)kept;
}

// valid edges of a node, in candidate order, at edge_off[g] (one wave per node)
__global__ void __launch_bounds__(256)
k_edge_fill(long long G, const long long *__restrict__ tri_off, const unsigned *__restrict__ edge_flag,
            const long long *__restrict__ edge_off, const CRec *__restrict__ cand,
            int *__restrict__ edges2, const unsigned *__restrict__ perm) {
  long long g = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (g >= G) return;
  const int lane = lane_id();
  const long long off = tri_off[g];
  const int n = (int)(tri_off[g + 1] - off);
  long long base = edge_off[g];
  if (edge_off[g + 1] == base) return;
  for (int i0 = 0; i0 < n; i0 += 64) {
    int i = i0 + lane;
    bool f = (i < n) && edge_flag[off + i];
    unsigned long long m = __ballot(f);
    if (f) {
      long long p = base + __popcll(m & lanemask_lt());
      const int2 l = *reinterpret_cast<const int2 *>(&cand[perm ? (long long)perm[off + i] : off + i].nb_slot);
      edges2[2 * p] = l.x & 0xFF;
      edges2[2 * p + 1] = l.y;
    }
    base += __popcll(m);
  }
}

// ---------------------------------------------------------------------------------------------
// launch wrappers (called from lt_api.cpp)
// ---------------------------------------------------------------------------------------------
static inline unsigned nblk(long long n, int bs) { return (unsigned)((n + bs - 1) / bs); }

void launch_build_cams(hipStream_t st, int n, const double *k, const double *q, const double *t, Cam *cams) {
  if (n > 0) hipLaunchKernelGGL(k_build_cams, dim3(nblk(n, 128)), dim3(128), 0, st, n, k, q, t, cams);
}
void launch_build_segs(hipStream_t st, long long n_segs, int n_img, const long long *seg_off,
                       const double *segs, double halfpix, const Cam *cams, Seg *out, void *gates) {
  if (n_segs > 0)
    hipLaunchKernelGGL(k_build_segs, dim3(nblk(n_segs, 256)), dim3(256), 0, st, n_segs, n_img, seg_off, segs,
                       halfpix, cams, out, reinterpret_cast<SegGate *>(gates));
}
// img_list (n_list entries, at most 65535 per launch) restricts the segment records to those images;
// nullptr = all images
void launch_build_scene_chunked(hipStream_t st, int n_img, long long n_segs, int n_chunks, const SceneChunk *ch,
                                const long long *seg_off, double halfpix, Cam *cams, Seg *segs,
                                const int *img_list, int n_list, long long max_segs_per_img, void *gates_v) {
  SegGate *gates = reinterpret_cast<SegGate *>(gates_v);
  const bool listed = img_list && n_list > 0 && max_segs_per_img > 0 && n_segs > 0;
  if (n_img > 0 && !listed)  // the listed form builds the cameras of the listed images itself
    hipLaunchKernelGGL(k_build_cams_chunked, dim3(nblk(n_img, 128)), dim3(128), 0, st, n_img, n_chunks, ch, cams);
  if (n_segs <= 0) return;
  if (listed) {
    for (int y0 = 0; y0 < n_list; y0 += 65535) {
      const int ny = std::min(65535, n_list - y0);
      hipLaunchKernelGGL(k_build_segs_listed, dim3(nblk(max_segs_per_img, 256), ny), dim3(256), 0, st, img_list + y0,
                         n_chunks, ch, seg_off, halfpix, cams, segs, gates);
    }
  } else {
    hipLaunchKernelGGL(k_build_segs_chunked, dim3(nblk(n_segs, 256)), dim3(256), 0, st, n_segs, n_img, n_chunks, ch,
                       seg_off, halfpix, cams, segs, gates);
  }
}
void launch_build_pairs(hipStream_t st, int n_blk, const int *blk_img, const int *blk_nb, const Cam *cams,
                        PairRec *out, int *err_flag, unsigned long long *pair_counter,
                        unsigned long long *scan_status, int n_status, unsigned *blk_surv) {
  hipLaunchKernelGGL(k_build_pairs, dim3(nblk(std::max(n_blk, 1), 128)), dim3(128), 0, st, n_blk, blk_img, blk_nb, cams,
                     out, err_flag, pair_counter, scan_status, n_status, blk_surv);
}
size_t sort_temp_bytes(long long P, int end_bit) {
  size_t bytes = 0;
  (void)rocprim::radix_sort_pairs(nullptr, bytes, (unsigned *)nullptr, (unsigned *)nullptr, (unsigned *)nullptr,
                            (unsigned *)nullptr, (size_t)P, 0, end_bit, (hipStream_t)0);
  return bytes;
}
int launch_sort(hipStream_t st, void *temp, size_t temp_bytes, long long P, const unsigned *keys_in,
                unsigned *keys_out, const unsigned *vals_in, unsigned *vals_out, int end_bit) {
  if (P <= 0) return 0;
  return (int)rocprim::radix_sort_pairs(temp, temp_bytes, keys_in, keys_out, vals_in, vals_out, (size_t)P, 0,
                                        end_bit, st);
}
void launch_node_offsets(hipStream_t st, long long P, long long G, const unsigned *skeys, long long *conn_off) {
  hipLaunchKernelGGL(k_node_offsets, dim3(nblk(P + 1, 256)), dim3(256), 0, st, P, G, skeys, conn_off);
}
size_t scan_temp_bytes_u32_to_i64(long long n) {
  size_t bytes = 0;
  (void)rocprim::exclusive_scan(nullptr, bytes, (unsigned *)nullptr, (long long *)nullptr, 0ll, (size_t)n,
                          rocprim::plus<long long>(), (hipStream_t)0);
  return bytes;
}
// exclusive scan of the popcounts of n 64-bit ballots (the plain exhaustive mode: no separate count pass or array)
struct PopcFn {
  __host__ __device__ long long operator()(unsigned long long m) const {
#if defined(__HIP_DEVICE_COMPILE__)
    return (long long)__popcll(m);
#else
    return (long long)__builtin_popcountll(m);
#endif
  }
};
size_t scan_temp_bytes_popc(long long n) {
  size_t bytes = 0;
  auto it = rocprim::make_transform_iterator((const unsigned long long *)nullptr, PopcFn());
  (void)rocprim::exclusive_scan(nullptr, bytes, it, (long long *)nullptr, 0ll, (size_t)n, rocprim::plus<long long>(),
                                (hipStream_t)0);
  return bytes;
}
int launch_scan_popc(hipStream_t st, void *temp, size_t temp_bytes, long long n, const unsigned long long *masks,
                     long long *out) {
  if (n <= 0) return 0;
  auto it = rocprim::make_transform_iterator(masks, PopcFn());
  return (int)rocprim::exclusive_scan(temp, temp_bytes, it, out, 0ll, (size_t)n, rocprim::plus<long long>(), st);
}
int launch_scan_u32_to_i64(hipStream_t st, void *temp, size_t temp_bytes, long long n, const unsigned *in,
                           long long *out) {
  if (n <= 0) return 0;
  return (int)rocprim::exclusive_scan(temp, temp_bytes, in, out, 0ll, (size_t)n, rocprim::plus<long long>(), st);
}
void launch_gen_exhaustive(hipStream_t st, bool fill, long long n_items, const GenCfg &cfg,
                           const long long *item_off, long long G, const int *node_img, const long long *nb_off,
                           const int *blk_nb, const long long *seg_off, const Cam *cams, const Seg *segs,
                           const PairRec *pairs, unsigned long long *masks, const long long *mask_pos,
                           CRec *out_r, double *out_unc, const double *seg_vp, const unsigned char *seg_has_vp,
                           const int *blk_chunk_off, int max_nb, int max_chunks, const void *gates_v) {
  if (n_items <= 0) return;
  const SegGate *gates = reinterpret_cast<const SegGate *>(gates_v);
  dim3 grid(nblk(G * (long long)max_nb * 64, 256)), block(256);  // one wave per (node, neighbour image)
#define LT_LAUNCH_EX(FILL, VP)                                                                                       \
  hipLaunchKernelGGL((k_gen_ex_block<FILL, VP>), grid, block, 0, st, n_items, cfg, item_off, G, node_img, nb_off, blk_nb, \
                     seg_off, cams, segs, pairs, masks, mask_pos, out_r, out_unc, seg_vp, seg_has_vp, blk_chunk_off, max_nb,  \
                     gates)
  if (seg_vp) {
    if (!fill) LT_LAUNCH_EX(false, true); else LT_LAUNCH_EX(true, true);
  } else {
    if (!fill) LT_LAUNCH_EX(false, false); else LT_LAUNCH_EX(true, false);
  }
#undef LT_LAUNCH_EX
}
// pass 1 of the plain exhaustive mode (no VP / point proposals): see k_gates_ex.  ent_out != nullptr: the one-pass form
// (gates only; entry blocks into ent_out; ctr = ex_regions() zeroed counters 128 bytes apart; region_cap slots per region)
int ex_regions() { return kExRegions; }
void launch_gates_exhaustive(hipStream_t st, int n_blk, int max_chunks, long long n_items, const GenCfg &cfg,
                             const long long *item_off, const int *blk_img, const int *blk_nb, const long long *seg_off,
                             const Cam *cams, const Seg *segs, const PairRec *pairs, unsigned long long *masks,
                             const int *blk_chunk_off, const void *gates_v, unsigned long long *ent_out,
                             unsigned long long *ctr, unsigned region_cap, int *err_flag) {
  if (n_items <= 0 || n_blk <= 0) return;
  const long long waves = (long long)n_blk * max_chunks * kExParts;
  const dim3 grid(nblk(waves * 64, 256)), block(256);
  const SegGate *gates = reinterpret_cast<const SegGate *>(gates_v);
  if (ent_out)
    hipLaunchKernelGGL(k_gates_ex<true>, grid, block, 0, st, n_blk, max_chunks, n_items, cfg, item_off, blk_img, blk_nb,
                       seg_off, cams, segs, pairs, masks, blk_chunk_off, gates, ent_out, ctr, region_cap, err_flag);
  else
    hipLaunchKernelGGL(k_gates_ex<false>, grid, block, 0, st, n_blk, max_chunks, n_items, cfg, item_off, blk_img, blk_nb,
                       seg_off, cams, segs, pairs, masks, blk_chunk_off, gates, ent_out, ctr, region_cap, err_flag);
}
// one-pass form: dense evaluation of the entry blocks (masks zeroed beforehand)
void launch_tri_exhaustive(hipStream_t st, unsigned long long *ent, const unsigned long long *ctr,
                           unsigned region_cap, const GenCfg &cfg, long long n_items, const long long *item_off,
                           const int *blk_img, const int *blk_nb, const long long *nb_off, const long long *seg_off,
                           const Cam *cams, const Seg *segs, const PairRec *pairs, const int *blk_chunk_off,
                           unsigned long long *masks, CRec *st_r, double *st_unc, unsigned *st_node, float *st_z) {
  if (region_cap == 0) return;
  hipLaunchKernelGGL(k_tri_ex, dim3(nblk((long long)region_cap, 256), kExRegions), dim3(256), 0, st, ent, ctr, region_cap,
                     cfg, n_items, item_off, blk_img, blk_nb, nb_off, seg_off, cams, segs, pairs, blk_chunk_off, masks, st_r,
                     st_unc, st_node, st_z);
}
// one-pass form: perm[final position] = staging slot, for every slot the regions handed out
void launch_place_exhaustive(hipStream_t st, const unsigned long long *ctr, unsigned region_cap,
                             const unsigned long long *ent, const unsigned *st_node, const long long *item_off,
                             const int *blk_chunk_off, const unsigned long long *masks, const long long *mask_pos,
                             long long n_items, const long long *tri_off, long long G, unsigned *perm,
                             long long *fill_out) {
  if (region_cap == 0) return;
  hipLaunchKernelGGL(k_place_ex, dim3(nblk((long long)region_cap, 256), kExRegions), dim3(256), 0, st, ctr, region_cap, ent,
                     st_node, item_off, blk_chunk_off, masks, mask_pos, n_items, tri_off, G, perm, fill_out);
}
// pass 2 of the plain exhaustive mode: see k_fill_ex
void launch_fill_exhaustive(hipStream_t st, int n_blk, long long n_items, const GenCfg &cfg, const long long *item_off,
                            const int *blk_img, const int *blk_nb, const long long *nb_off, const long long *seg_off,
                            const Cam *cams, const Seg *segs, const PairRec *pairs, const unsigned long long *masks,
                            const long long *mask_pos, CRec *out_r, double *out_unc, const int *blk_chunk_off) {
  if (n_items <= 0 || n_blk <= 0) return;
  const long long waves = (long long)n_blk * kFillParts;
  hipLaunchKernelGGL(k_fill_ex, dim3(nblk(waves * 64, 256)), dim3(256), 0, st, n_blk, n_items, cfg, item_off, blk_img,
                     blk_nb, nb_off, seg_off, cams, segs, pairs, masks, mask_pos, out_r, out_unc, blk_chunk_off);
}
void launch_gen_exhaustive_pts(hipStream_t st, bool fill, long long n_items, const GenCfg &cfg,
                               const long long *item_off, long long G, const int *node_img, const long long *nb_off,
                               const int *blk_nb, const long long *seg_off, const Cam *cams, const Seg *segs,
                               const PairRec *pairs, unsigned short *cnt8, unsigned *item_cnt,
                               const long long *mask_pos, CRec *out_r, double *out_unc, const double *seg_vp,
                               const unsigned char *seg_has_vp, const long long *seg_pt_off, const void *seg_pts,
                               const double *sfm_xyz, int *err_flag, int many_on, int one_on,
                               const int *blk_chunk_off, int max_nb, int max_chunks, const void *gates_v) {
  if (n_items <= 0) return;
  const SegGate *gates = reinterpret_cast<const SegGate *>(gates_v);
  dim3 grid(nblk(G * (long long)max_nb * max_chunks * 64, 256)), block(256);
  const SegPoint *sp = reinterpret_cast<const SegPoint *>(seg_pts);
  if (!fill)
    hipLaunchKernelGGL((k_gen_exhaustive_pts<false>), grid, block, 0, st, n_items, cfg, item_off, G, node_img, nb_off,
                       blk_nb, seg_off, cams, segs, pairs, cnt8, item_cnt, mask_pos, out_r, out_unc, seg_vp, seg_has_vp,
                       seg_pt_off, sp, sfm_xyz, err_flag, many_on, one_on, blk_chunk_off, max_nb, max_chunks, gates);
  else
    hipLaunchKernelGGL((k_gen_exhaustive_pts<true>), grid, block, 0, st, n_items, cfg, item_off, G, node_img, nb_off,
                       blk_nb, seg_off, cams, segs, pairs, cnt8, item_cnt, mask_pos, out_r, out_unc, seg_vp, seg_has_vp,
                       seg_pt_off, sp, sfm_xyz, err_flag, many_on, one_on, blk_chunk_off, max_nb, max_chunks, gates);
}
// n_masks ballots per item (3 with VP proposals)
void launch_popc(hipStream_t st, long long n, const unsigned long long *masks, unsigned *cnt, int n_masks) {
  if (n > 0) hipLaunchKernelGGL(k_popc, dim3(nblk(n, 256)), dim3(256), 0, st, n, masks, cnt, n_masks);
}
void launch_tri_offsets_ex(hipStream_t st, long long G, const long long *item_off, const long long *mask_pos,
                           long long n_items, long long total, long long cap, long long *tri_off, int *err_flag) {
  hipLaunchKernelGGL(k_tri_offsets_ex, dim3(nblk(G + 1, 256)), dim3(256), 0, st, G, item_off, mask_pos, n_items,
                     total, cap, tri_off, err_flag);
}
void launch_select(hipStream_t st, long long G, const long long *tri_off, const double *score, double th,
                   int max_valid, long long *best_idx, unsigned *edge_flag, unsigned *n_valid, const CRec *cand,
                   const double *cand_unc, Cand *best_c, double *best_score, int *best_src2, int *n_tris,
                   bool wide, const int *err_flag, const unsigned long long *pair_counter, long long *result3,
                   const unsigned *perm) {
  if (G > 0)
  {
    if (wide)
      hipLaunchKernelGGL(k_select<64>, dim3(nblk(G * 64, 256)), dim3(256), 0, st, G, tri_off, score, th, max_valid,
                         best_idx, edge_flag, n_valid, cand, cand_unc, best_c, best_score, best_src2, n_tris, err_flag,
                         pair_counter, result3, perm);
    else
      hipLaunchKernelGGL(k_select<16>, dim3(nblk(G * 16, 256)), dim3(256), 0, st, G, tri_off, score, th, max_valid,
                         best_idx, edge_flag, n_valid, cand, cand_unc, best_c, best_score, best_src2, n_tris, err_flag,
                         pair_counter, result3, perm);
  }
}
void launch_edge_fill(hipStream_t st, long long G, const long long *tri_off, const unsigned *edge_flag,
                      const long long *edge_off, const CRec *cand, int *edges2, const unsigned *perm) {
  if (G > 0)
    hipLaunchKernelGGL(k_edge_fill, dim3(nblk(G * 64, 256)), dim3(256), 0, st, G, tri_off, edge_flag, edge_off,
                       cand, edges2, perm);
}


// ---------------------------------------------------------------------------------------------
// RemergeLineTracks (merging/merging.cc:523-556): all-pairs LineLinker3d::check_connection between
// the track lines.  lane = track i (active only), wave-uniform sweep over a chunk of tracks j;
// conservative cosine early exit (angle <= th_angle can only hold if |cos| >= cos(th (1+1e-6))),
// exact check for the rest; edges (min << 32 | max) appended through a device counter.
// When every track is active the reference tests each unordered pair once, from the side given by
// the parity of i + j (:535-540); that orientation is kept because l1/l2 are not interchangeable
// bit for bit.
// ---------------------------------------------------------------------------------------------
constexpr int kTrackChunk = 64;  // tracks j per workgroup column: small, so that a few thousand tracks already fill the GPU
// Two phases per wave (round 4; the one-phase form ran check3d -- ~700 instructions -- for the whole wave whenever ONE lane's
// pair passed the cosine test: 100-165 us for 1 340 tracks): a sweep of the 64 x 64 pairs of the wave with the cosine test
// only (unit directions of the j tracks once per workgroup in LDS) that queues the survivors in LDS, then check3d over the
// queue with every lane busy.  Same tests on the same operands, same edges.
__global__ void __launch_bounds__(256)
k_track_connect(int T, const double *__restrict__ line7, const unsigned char *__restrict__ active, int all_active,
                LinkCfg3 cfg, double cos_guard, unsigned long long *__restrict__ edges,
                unsigned long long capacity, unsigned long long *__restrict__ n_edges) {
  __shared__ double s_lj[kTrackChunk][7];     // the j tracks of this column
  __shared__ double s_dj[kTrackChunk][3];     // their unit directions
  __shared__ double s_li[4][64][7];           // per wave: its i tracks
  __shared__ unsigned short s_q[4][64 * kTrackChunk];  // per wave: surviving pairs (lane of i << 8 | j - j0)
  const int wave = threadIdx.x >> 6, lane = lane_id();
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int j0 = blockIdx.y * kTrackChunk;
  const int j1 = min(T, j0 + kTrackChunk);
  const bool live = (i < T) && active[i];
  if (__syncthreads_or(live ? 1 : 0) == 0) return;  // no active track among the workgroup's 256
  if (threadIdx.x < (unsigned)(j1 - j0)) {
    const double *q = line7 + 7 * (long long)(j0 + (int)threadIdx.x);
    for (int k = 0; k < 7; ++k) s_lj[threadIdx.x][k] = q[k];
    const d3 dj = dir(L3{mk3(q[0], q[1], q[2]), mk3(q[3], q[4], q[5])});
    s_dj[threadIdx.x][0] = dj.x; s_dj[threadIdx.x][1] = dj.y; s_dj[threadIdx.x][2] = dj.z;
  }
  d3 di = mk3(0, 0, 0);
  if (live) {
    const double *p = line7 + 7 * (long long)i;
    for (int k = 0; k < 7; ++k) s_li[wave][lane][k] = p[k];
    di = dir(L3{mk3(p[0], p[1], p[2]), mk3(p[3], p[4], p[5])});
  }
  __syncthreads();
  // phase 1: which pairs go on (j is wave-uniform)
  int qn = 0;
  for (int j = j0; j < j1; ++j) {
    bool test = live && (j != i);
    if (test && all_active) {
      if (i < j && ((i + j) & 1) == 0) test = false;
      if (i > j && ((i + j) & 1) == 1) test = false;
    }
    if (test && cfg.use_angle) {
      const d3 dj = mk3(s_dj[j - j0][0], s_dj[j - j0][1], s_dj[j - j0][2]);
      test = !(fabs(dot(di, dj)) < cos_guard);
    }
    const unsigned long long m = __ballot(test);
    if (m) {
      if (test) s_q[wave][qn + __popcll(m & lanemask_lt())] = (unsigned short)((lane << 8) | (j - j0));
      qn += __popcll(m);
    }
  }
  wave_lds_sync();
  // phase 2: the exact check, one pair per lane
  const double dep[2] = {0.0, 0.0};
  for (int q0 = 0; q0 < qn; q0 += 64) {
    const int p = q0 + lane;
    bool hit = false;
    int ii = 0, jj = 0;
    if (p < qn) {
      const unsigned e = s_q[wave][p];
      const int il = (int)(e >> 8), jl = (int)(e & 0xFFu);
      const double *a = s_li[wave][il], *b = s_lj[jl];
      ii = (int)(blockIdx.x * blockDim.x) + wave * 64 + il;
      jj = j0 + jl;
      hit = check3d(cfg, L3{mk3(a[0], a[1], a[2]), mk3(a[3], a[4], a[5])}, L3{mk3(b[0], b[1], b[2]), mk3(b[3], b[4], b[5])},
                    a[6], b[6], dep);
    }
    // one counter update per wave and round (a device-scope atomic per edge would serialise)
    const unsigned long long m = __ballot(hit);
    if (m) {
      unsigned long long base = 0;
      if (lane == 0) base = atomicAdd(n_edges, (unsigned long long)__popcll(m));
      base = ((unsigned long long)(unsigned)__shfl((int)(base >> 32), 0) << 32) | (unsigned)__shfl((int)(base & 0xFFFFFFFFull), 0);
      if (hit) {
        const unsigned long long slot = base + (unsigned long long)__popcll(m & lanemask_lt());
        const unsigned long long a = (unsigned long long)min(ii, jj), b = (unsigned long long)max(ii, jj);
        if (slot < capacity) edges[slot] = (a << 32) | b;
      }
    }
  }
}

void launch_track_connect(hipStream_t st, int T, const double *line7, const unsigned char *active, int all_active,
                          const LinkCfg3 &cfg, double cos_guard, unsigned long long *edges,
                          unsigned long long capacity, unsigned long long *n_edges) {
  if (T <= 0) return;
  hipLaunchKernelGGL(k_track_connect, dim3(nblk(T, 256), nblk(T, kTrackChunk)), dim3(256), 0, st, T, line7, active,
                     all_active, cfg, cos_guard, edges, capacity, n_edges);
}

// ---------------------------------------------------------------------------------------------
// free-function queries (limap.triangulation.get_normal_direction / get_direction_from_VP /
// compute_fundamental_matrix / compute_epipolar_IoU / triangulate_point / triangulate_line[_by_endpoints] /
// triangulate_line_with_direction, bindings.cc:22-31): one thread.
// in37 = seg1[4] cam1[11] seg2[4] cam2[11] v[3] p1[2] p2[2]
// out40 = n(seg1)[3] F[9] IoU line10[10] | dir_from_vp(v, cam1)[3] | point[3] ok | line10 with direction v [10]
// ---------------------------------------------------------------------------------------------
__global__ void k_fn_query(const double *__restrict__ in, int by_endpoints, double *__restrict__ out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  Cam c1, c2;
  cam_build(in + 4, in + 8, in + 12, &c1);
  cam_build(in + 19, in + 23, in + 27, &c2);
  Seg s1, s2;
  seg_build(c1, in[0], in[1], in[2], in[3], &s1);
  seg_build(c2, in[15], in[16], in[17], in[18], &s2);
  PairRec pr;
  pair_build(c1, c2, &pr);
  out[0] = s1.n[0]; out[1] = s1.n[1]; out[2] = s1.n[2];
  for (int k = 0; k < 9; ++k) out[3 + k] = pr.F[k];
  out[12] = epipolar_iou(s1, s2, pr.F);
  d3 ps, pe;
  double zs, ze, d21, d22;
  bool ok;
  if (!by_endpoints) {
    ok = tri_line(c1, c2, s1, s2, pr.B, &ps, &pe, &zs, &ze, &d21, &d22);
  } else {
    d3 r1s = mk3(s1.rs[0], s1.rs[1], s1.rs[2]), r1e = mk3(s1.re[0], s1.re[1], s1.re[2]);
    d3 c2s = mk3(s2.rs[0], s2.rs[1], s2.rs[2]), c2e = mk3(s2.re[0], s2.re[1], s2.re[2]);
    ok = tri_point(c1, r1s, c2, c2s, &ps) && tri_point(c1, r1e, c2, c2e, &pe);
    if (ok) {
      zs = cam_depth(c1, ps);
      ze = cam_depth(c1, pe);
    }
  }
  double *l = out + 13;
  if (ok) {
    l[0] = ps.x; l[1] = ps.y; l[2] = ps.z; l[3] = pe.x; l[4] = pe.y; l[5] = pe.z;
    l[6] = zs; l[7] = ze; l[8] = -1.0; l[9] = 1.0;
  } else {  // failure sentinel Line3d((0,0,0),(1,1,1),-1)  (functions.cc:300)
    l[0] = l[1] = l[2] = 0.0; l[3] = l[4] = l[5] = 1.0;
    l[6] = l[7] = -1.0; l[8] = -1.0; l[9] = -1.0;
  }
  // get_direction_from_VP(v, view1)
  const d3 dvp = unit(mv(c1.Minv, mk3(in[30], in[31], in[32])));
  out[23] = dvp.x; out[24] = dvp.y; out[25] = dvp.z;
  // triangulate_point(p1, view1, p2, view2): the product's tri_point without the cheirality part of the
  // line functions is not exposed separately; the reference's free function returns (point, ok) with the
  // same cheirality test (functions.cc:100-117)
  {
    d3 r1 = cam_ray(c1, d2{in[33], in[34]}), r2 = cam_ray(c2, d2{in[35], in[36]});
    d3 pt = mk3(0, 0, 0);
    bool okp = tri_point(c1, r1, c2, r2, &pt);
    out[26] = pt.x; out[27] = pt.y; out[28] = pt.z; out[29] = okp ? 1.0 : 0.0;
  }
  // triangulate_line_with_direction(l1, view1, l2, view2, v): the proposal without uncertainty / ranges
  {
    GenCfg cfg0;
    cfg0.var2d = 0.0; cfg0.use_ranges = 0;
    GenOut o;
    // vp_candidate maps a VP through view 1 first; here v already IS the world direction: feed K R v,
    // which view 1 maps back to unit(v) up to rounding -- not bit-faithful, so evaluate directly instead
    const d3 direction = mk3(in[30], in[31], in[32]);
    bool okd = dir_candidate(cfg0, c1, c2, s1, s2, pr.B, direction, &o);
    double *m = out + 30;
    if (okd) {
      m[0] = o.r.s[0]; m[1] = o.r.s[1]; m[2] = o.r.s[2]; m[3] = o.r.e[0]; m[4] = o.r.e[1]; m[5] = o.r.e[2];
      m[6] = o.r.depth[0]; m[7] = o.r.depth[1]; m[8] = -1.0; m[9] = 1.0;
    } else {
      m[0] = m[1] = m[2] = 0.0; m[3] = m[4] = m[5] = 1.0;
      m[6] = m[7] = -1.0; m[8] = -1.0; m[9] = -1.0;
    }
    // triangulate_line_with_one_point(l1, view1, l2, view2, point): same input slot as the direction
    GenOut o1;
    const bool ok1 = one_point_candidate(cfg0, c1, c2, s1, s2, direction, &o1);
    m = out + 40;
    if (ok1) {
      m[0] = o1.r.s[0]; m[1] = o1.r.s[1]; m[2] = o1.r.s[2]; m[3] = o1.r.e[0]; m[4] = o1.r.e[1]; m[5] = o1.r.e[2];
      m[6] = o1.r.depth[0]; m[7] = o1.r.depth[1]; m[8] = -1.0; m[9] = 1.0;
    } else {
      m[0] = m[1] = m[2] = 0.0; m[3] = m[4] = m[5] = 1.0;
      m[6] = m[7] = -1.0; m[8] = -1.0; m[9] = -1.0;
    }
  }
}

void launch_fn_query(hipStream_t st, const double *in30, int by_endpoints, double *out32) {
  hipLaunchKernelGGL(k_fn_query, dim3(1), dim3(64), 0, st, in30, by_endpoints, out32);
}

}  // namespace lt
