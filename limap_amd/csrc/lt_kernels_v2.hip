// lt_kernels_v2.hip -- second-generation kernels of the matched-mode pipeline and of the scoring
// stage (see DESIGN.md section 3 for the map; lt_kernels.hip keeps the invariant builders, the
// generic radix-sort grouping, the exhaustive generation and the selection kernels).
//
//   k_line_off_init / k_line_off / k_node_conn_count / k_build_rowlist
//       Fast grouping of the match rows by node when every (image, neighbour) block lists its rows
//       in non-decreasing line id (what limap's matchers write): per-(block, line) row offsets are
//       read off the row stream, so no sort is needed to obtain the reference's candidate order
//       (neighbour-ascending, then match-row order -- base_line_triangulator.cc:71-103).
//   k_gen_rows
//       HOT LOOP 1 in row (block) order: coalesced match rows, the neighbour's segment table is
//       shared by all rows of a block (L1/L2 locality), stage A = cheap gates on every row, stage B =
//       triangulation etc. only for the survivors, which are first gathered in an LDS queue so that
//       the expensive path runs on full wave64s instead of a few stray lanes.
//   k_node_fill
//       Ordered per-node compaction of the survivors (wave per node, ballot prefix).
//   k_score3
//       HOT LOOP 2, candidate-major: lane = candidate (nodes packed densely into waves), LDS-staged
//       sweep over the candidates of the lane's own node with a two-level conservative early exit
//       (cosine, then squared scale-invariant endpoint distance), survivors evaluated densely from
//       an LDS queue.
// Compiled with -ffp-contract=off (see lt_geom.h).

#include "lt_devfn.h"

namespace lt {

// ---------------------------------------------------------------------------------------------
// fast grouping
// ---------------------------------------------------------------------------------------------
// entry (b, l) of line_off lives at blk_line_base[b] + l, l in [0, M_img(b)]
__global__ void k_line_off_init(int n_blk, const long long *__restrict__ blk_line_base,
                                const long long *__restrict__ m_off, unsigned *__restrict__ line_off) {
  long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= blk_line_base[n_blk]) return;
  int lo = 0, hi = n_blk;
  while (hi - lo > 1) {
    int mid = (lo + hi) >> 1;
    if (blk_line_base[mid] <= e) lo = mid; else hi = mid;
  }
  line_off[e] = (unsigned)m_off[lo + 1];
}

// grid.y = neighbour block, grid.x = 256-row chunk of the block: no search for the block of a row
__global__ void __launch_bounds__(256)
k_line_off(const long long *__restrict__ m_off, const int *__restrict__ m_pairs,
           const long long *__restrict__ blk_line_base, unsigned *__restrict__ line_off,
           int *__restrict__ unsorted_flag) {
  const int b = blockIdx.y;
  const long long rb = m_off[b], re = m_off[b + 1];
  long long r = rb + (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= re) return;
  int line = m_pairs[2 * r];
  int prev = (r == rb) ? -1 : m_pairs[2 * (r - 1)];
  if (line < prev) {
    *unsorted_flag = 1;  // the host checked this already; never expected
    return;
  }
  long long base = blk_line_base[b];
  for (int l = prev + 1; l <= line; ++l) line_off[base + l] = (unsigned)r;
}

// connections of node g = sum over the image's neighbour blocks of the rows with this line id
__global__ void k_node_conn_count(long long G, const int *__restrict__ node_img,
                                  const long long *__restrict__ seg_off, const long long *__restrict__ nb_off,
                                  const long long *__restrict__ blk_line_base,
                                  const unsigned *__restrict__ line_off, unsigned *__restrict__ conn_cnt) {
  long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (g > G) return;
  unsigned c = 0;
  if (g < G) {
    int img = node_img[g];
    int line = (int)(g - seg_off[img]);
    for (long long b = nb_off[img]; b < nb_off[img + 1]; ++b) {
      long long e = blk_line_base[b] + line;
      c += line_off[e + 1] - line_off[e];
    }
  }
  conn_cnt[g] = c;
}

// srows[conn_off[g] ...] = the node's rows, neighbour-major (one wave per node, lane = block)
__global__ void __launch_bounds__(256)
k_build_rowlist(long long G, const int *__restrict__ node_img, const long long *__restrict__ seg_off,
                const long long *__restrict__ nb_off, const long long *__restrict__ blk_line_base,
                const unsigned *__restrict__ line_off, const long long *__restrict__ conn_off,
                unsigned *__restrict__ srows) {
  long long g = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (g >= G) return;
  const int lane = lane_id();
  long long out = conn_off[g];
  if (conn_off[g + 1] == out) return;
  int img = node_img[g];
  int line = (int)(g - seg_off[img]);
  long long b0 = nb_off[img], b1 = nb_off[img + 1];
  for (long long bb = b0; bb < b1; bb += 64) {
    long long b = bb + lane;
    unsigned lo = 0, cnt = 0;
    if (b < b1) {
      long long e = blk_line_base[b] + line;
      lo = line_off[e];
      cnt = line_off[e + 1] - lo;
    }
    unsigned pre = cnt;  // inclusive wave scan
    for (int d = 1; d < 64; d <<= 1) {
      unsigned v = __shfl_up(pre, d);
      if (lane >= d) pre += v;
    }
    unsigned total = __shfl(pre, 63);
    long long w = out + (pre - cnt);
    for (unsigned t = 0; t < cnt; ++t) srows[w + t] = lo + t;
    out += total;
  }
}

// ---------------------------------------------------------------------------------------------
// HOT LOOP 1, row order
// ---------------------------------------------------------------------------------------------
constexpr int kGenChunks = 8;   // 64-row chunks per wave
constexpr int kGenQCap = 192;   // LDS queue entries per wave (drained at >= 64 + ... see below)

__global__ void __launch_bounds__(256)
k_gen_rows(GenCfg cfg, const long long *__restrict__ m_off, const int *__restrict__ m_pairs,
           const int *__restrict__ blk_img, const int *__restrict__ blk_nb, const int *__restrict__ blk_slot,
           const long long *__restrict__ seg_off, const Cam *__restrict__ cams, const Seg *__restrict__ segs,
           const PairRec *__restrict__ pairs, Cand *__restrict__ st_c, CandLite *__restrict__ st_l,
           unsigned char *__restrict__ flag8, unsigned *__restrict__ n_tris) {
  __shared__ unsigned q_row[4][kGenQCap];
  const int wave = threadIdx.x >> 6;
  const int lane = lane_id();
  // grid.y = neighbour block (uniform: image pair, F, cameras live in scalar registers),
  // grid.x * 4 waves * kGenChunks * 64 rows cover the block's rows
  const int b = blockIdx.y;
  const long long rb = m_off[b], re = m_off[b + 1];
  const long long r0 = rb + ((long long)blockIdx.x * 4 + wave) * (64ll * kGenChunks);
  if (r0 >= re) return;
  const int i1 = blk_img[b], i2 = blk_nb[b], slot = blk_slot[b];
  const long long g1 = seg_off[i1], g2 = seg_off[i2];
  const PairRec *pr = pairs + b;
  unsigned *qr = q_row[wave];
  int qn = 0;

  auto stage_b = [&](int count) {  // dense: lanes 0..count-1 finish one surviving connection each
    if (lane < count) {
      unsigned r = qr[lane];
      int line = m_pairs[2 * (long long)r], ng = m_pairs[2 * (long long)r + 1];
      GenOut o;
      if (gen_finish(cfg, cams[i1], cams[i2], segs[g1 + line], segs[g2 + ng], pr->B, &o)) {
        o.l.nb_slot = lite_pack(slot, i2);
        o.l.ng_line = ng;
        st_c[r] = o.c;
        st_l[r] = o.l;
        flag8[r] = 1;
        atomicAdd(&n_tris[g1 + line], 1u);
      }
    }
  };

  for (int c = 0; c < kGenChunks; ++c) {
    long long r = r0 + 64ll * c + lane;
    bool pass = false;
    if (r < re) {
      int line = m_pairs[2 * r], ng = m_pairs[2 * r + 1];
      pass = gen_gates_fast(cfg, segs[g1 + line], segs[g2 + ng], pr->F);
    }
    unsigned long long m = __ballot(pass);
    if (m) {
      if (pass) qr[qn + __popcll(m & lanemask_lt())] = (unsigned)r;
      qn += __popcll(m);
      wave_lds_sync();
      while (qn >= 64) {  // process the oldest 64, shift the rest down
        stage_b(64);
        wave_lds_sync();
        int rest = qn - 64;
        unsigned tr = 0;
        if (lane < rest) tr = qr[64 + lane];
        wave_lds_sync();
        if (lane < rest) qr[lane] = tr;
        wave_lds_sync();
        qn = rest;
      }
    }
  }
  if (qn > 0) stage_b(qn);
}

// ordered per-node compaction of the staged survivors (one wave per node)
__global__ void __launch_bounds__(256)
k_node_fill(long long G, const long long *__restrict__ conn_off, const unsigned *__restrict__ srows,
            const unsigned char *__restrict__ flag8, const long long *__restrict__ tri_off,
            const Cand *__restrict__ st_c, const CandLite *__restrict__ st_l, Cand *__restrict__ cand,
            CandLite *__restrict__ lite, unsigned *__restrict__ cand_node) {
  long long g = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (g >= G) return;
  const int lane = lane_id();
  long long out = tri_off[g];
  if (tri_off[g + 1] == out) return;
  const long long c0 = conn_off[g], c1 = conn_off[g + 1];
  for (long long t0 = c0; t0 < c1; t0 += 256) {  // four independent 64-connection chunks in flight
    unsigned r[4];
    bool f[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      long long t = t0 + 64 * k + lane;
      r[k] = (t < c1) ? srows[t] : 0u;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      long long t = t0 + 64 * k + lane;
      f[k] = (t < c1) && flag8[r[k]] != 0;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      unsigned long long m = __ballot(f[k]);
      if (f[k]) {
        long long p = out + __popcll(m & lanemask_lt());
        cand[p] = st_c[r[k]];
        lite[p] = st_l[r[k]];
        cand_node[p] = (unsigned)g;
      }
      out += __popcll(m);
    }
  }
}

// cand_node for pipelines that produce the compact arrays directly (exhaustive mode)
__global__ void __launch_bounds__(256)
k_cand_node(long long G, const long long *__restrict__ tri_off, unsigned *__restrict__ cand_node) {
  long long g = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (g >= G) return;
  for (long long t = tri_off[g] + lane_id(); t < tri_off[g + 1]; t += 64) cand_node[t] = (unsigned)g;
}

// ---------------------------------------------------------------------------------------------
// HOT LOOP 2, candidate-major (scoreOneNode, global_line_triangulator.cc:71-116)
// ---------------------------------------------------------------------------------------------
// One wave64 per 64 consecutive candidates (lane = candidate i; small nodes are packed densely into
// the wave).  The candidates of all nodes the wave touches are staged through an LDS window (SoA:
// direction, endpoints, neighbour slot), so the O(n^2) sweep runs out of LDS: every lane walks the
// candidates j of ITS OWN node and applies a two-level conservative early exit (cosine of the 3D
// angle gate, then the squared one-way scale-invariant endpoint gate with l_i's depths).  Survivors
// are pushed (ballot + popcount) into an LDS queue and evaluated densely, one pair per lane; the
// per-neighbour-image maxima live in LDS (ds_max_u64 on the bit pattern of the non-negative scores)
// and are summed per lane in ascending image-id order (std::map order, :110-112).
constexpr int kSQCap = 192;
constexpr int kWin = 128;

struct Score3Args {
  long long G;
  const long long *tri_off;  // tri_off[G] = C
  const unsigned *cand_node;
  const Cand *cand;
  const CandLite *lite;
  const int *node_img;
  const long long *nb_off;
  const int *blk_order;
  const Cam *cams;
  double *score;
  unsigned long long *pair_counter;  // stats: pairs that reached the dense evaluation
  int max_nb;
};

__global__ void __launch_bounds__(64)
k_score3(Score3Args a, ScoreCfg cfg, double scaleinv_guard2) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  const int lane = threadIdx.x;
  // LDS: W[9][kWin] f64 | S[max_nb][64] u64 | woff[64] i64 | wslot[kWin] i32 | queue[kSQCap] u32
  double *W = reinterpret_cast<double *>(smem_raw);
  unsigned long long *S = reinterpret_cast<unsigned long long *>(smem_raw + 9 * kWin * 8);
  long long *woff = reinterpret_cast<long long *>(smem_raw + 9 * kWin * 8 + (size_t)a.max_nb * 64 * 8);
  int *wslot = reinterpret_cast<int *>(smem_raw + 9 * kWin * 8 + (size_t)a.max_nb * 64 * 8 + 64 * 8);
  unsigned *queue = reinterpret_cast<unsigned *>(smem_raw + 9 * kWin * 8 + (size_t)a.max_nb * 64 * 8 + 64 * 8 + kWin * 4);

  const long long C = a.tri_off[a.G];
  const long long i0 = (long long)blockIdx.x * 64;
  if (i0 >= C) return;
  const long long i = i0 + lane;
  const bool active = i < C;

  long long off = 0, nb0 = 0;
  int n = 0, n_nb = 0, sloti = -1;
  double dix = 0, diy = 0, diz = 0;
  double six = 0, siy = 0, siz = 0, eix = 0, eiy = 0, eiz = 0, gs2 = 0, ge2 = 0;
  if (active) {
    const unsigned g = a.cand_node[i];
    off = a.tri_off[g];
    n = (int)(a.tri_off[g + 1] - off);
    const int img = a.node_img[g];
    nb0 = a.nb_off[img];
    n_nb = (int)(a.nb_off[img + 1] - nb0);
    const CandLite li = a.lite[i];
    const Cand ci = a.cand[i];
    dix = li.dir[0]; diy = li.dir[1]; diz = li.dir[2];
    sloti = lite_slot(li);
    six = ci.s[0]; siy = ci.s[1]; siz = ci.s[2];
    eix = ci.e[0]; eiy = ci.e[1]; eiz = ci.e[2];
    // dist / (depth + eps) > th_scaleinv (1 + 1e-6) can never score >= score_th (line_dists.cc:55-60)
    double zs = ci.depth[0] + kEps, ze = ci.depth[1] + kEps;
    gs2 = (zs > 0.0) ? scaleinv_guard2 * zs * zs : 1e300;  // odd depths: leave it to the exact path
    ge2 = (ze > 0.0) ? scaleinv_guard2 * ze * ze : 1e300;
  }
  woff[lane] = off;
  for (int k = 0; k < a.max_nb; ++k) S[k * 64 + lane] = 0ull;
  // candidate range of all nodes this wave touches (lane 0 is always active)
  const long long lo = __shfl(off, 0);
  long long hi = active ? off + n : 0;
  for (int d = 32; d >= 1; d >>= 1) {
    long long o = __shfl_xor(hi, d);
    hi = o > hi ? o : hi;
  }
  int qn = 0;
  unsigned long long n_eval = 0;

  auto drain = [&]() {
    wave_lds_sync();
    for (int q0 = 0; q0 < qn; q0 += 64) {
      int p = q0 + lane;
      if (p < qn) {
        unsigned e = queue[p];
        int il = (int)(e >> 26);
        long long j = woff[il] + (long long)(e & 0x3FFFFFFu);
        long long ii = i0 + il;
        const Cand ci = a.cand[ii];
        const CandLite li = a.lite[ii];
        const CandLite lj = a.lite[j];
        const Cand cj = a.cand[j];
        double sc = pair_score(cfg, mk3(ci.s[0], ci.s[1], ci.s[2]), mk3(ci.e[0], ci.e[1], ci.e[2]),
                               mk3(li.dir[0], li.dir[1], li.dir[2]), ci.depth[0], ci.depth[1],
                               mk3(cj.s[0], cj.s[1], cj.s[2]), mk3(cj.e[0], cj.e[1], cj.e[2]),
                               mk3(lj.dir[0], lj.dir[1], lj.dir[2]), cj.seg, a.cams[lite_img(lj)]);
        if (sc > 0.0) atomicMax(&S[lite_slot(lj) * 64 + il], (unsigned long long)__double_as_longlong(sc));
      }
    }
    n_eval += (unsigned long long)qn;
    qn = 0;
    wave_lds_sync();
  };

  for (long long wb = lo; wb < hi; wb += kWin) {
    wave_lds_sync();
    const int wn = (int)((hi - wb) < kWin ? (hi - wb) : kWin);
    for (int e = lane; e < wn; e += 64) {
      const CandLite l = a.lite[wb + e];
      const Cand c = a.cand[wb + e];
      W[0 * kWin + e] = l.dir[0]; W[1 * kWin + e] = l.dir[1]; W[2 * kWin + e] = l.dir[2];
      W[3 * kWin + e] = c.s[0]; W[4 * kWin + e] = c.s[1]; W[5 * kWin + e] = c.s[2];
      W[6 * kWin + e] = c.e[0]; W[7 * kWin + e] = c.e[1]; W[8 * kWin + e] = c.e[2];
      wslot[e] = lite_slot(l);
    }
    wave_lds_sync();
    // this lane's sub-range of the window
    long long jlo = off > wb ? off : wb;
    long long jhi = (off + n) < (wb + wn) ? (off + n) : (wb + wn);
    int cnt = (active && jhi > jlo) ? (int)(jhi - jlo) : 0;
    int cmax = cnt;
    for (int d = 32; d >= 1; d >>= 1) cmax = max(cmax, __shfl_xor(cmax, d));
    const int w0 = (int)(jlo - wb);
    const int jj0 = (int)(jlo - off);
    for (int t = 0; t < cmax; ++t) {
      bool pass = t < cnt;
      if (pass) {
        const int w = w0 + t;
        pass = (jlo + t != i) && (wslot[w] != sloti);
        if (pass) {
          double c = fabs((dix * W[0 * kWin + w] + diy * W[1 * kWin + w]) + diz * W[2 * kWin + w]);
          pass = !(c < cfg.cos_guard);  // below the guard the 3D angle score is certainly gated to 0
          if (pass) {
            double ax = six - W[3 * kWin + w], ay = siy - W[4 * kWin + w], az = siz - W[5 * kWin + w];
            double bx = eix - W[6 * kWin + w], by = eiy - W[7 * kWin + w], bz = eiz - W[8 * kWin + w];
            double ds2 = ax * ax + ay * ay + az * az, de2 = bx * bx + by * by + bz * bz;
            pass = !(ds2 > gs2) && !(de2 > ge2);
          }
        }
      }
      unsigned long long m = __ballot(pass);
      if (m) {
        if (pass) queue[qn + __popcll(m & lanemask_lt())] = ((unsigned)lane << 26) | (unsigned)(jj0 + t);
        qn += __popcll(m);
        if (qn > kSQCap - 64) drain();
      }
    }
  }
  drain();

  if (active) {
    double sum = 0.0;
    for (int r = 0; r < n_nb; ++r) {
      int k = a.blk_order[nb0 + r];
      sum += __longlong_as_double((long long)S[k * 64 + lane]);
    }
    a.score[i] = sum;
  }
  if (lane == 0 && a.pair_counter) atomicAdd(a.pair_counter, n_eval);
}

// ---------------------------------------------------------------------------------------------
// launch wrappers
// ---------------------------------------------------------------------------------------------
static inline unsigned nblk2(long long n, int bs) { return (unsigned)((n + bs - 1) / bs); }

void launch_line_off(hipStream_t st, long long P, int n_blk, long long n_entries, long long max_rows,
                     const long long *m_off, const int *m_pairs, const long long *blk_line_base,
                     unsigned *line_off, int *unsorted_flag) {
  if (n_entries > 0)
    hipLaunchKernelGGL(k_line_off_init, dim3(nblk2(n_entries, 256)), dim3(256), 0, st, n_blk, blk_line_base, m_off,
                       line_off);
  if (P > 0 && n_blk > 0 && max_rows > 0)
    hipLaunchKernelGGL(k_line_off, dim3(nblk2(max_rows, 256), n_blk), dim3(256), 0, st, m_off, m_pairs,
                       blk_line_base, line_off, unsorted_flag);
}
void launch_node_conn_count(hipStream_t st, long long G, const int *node_img, const long long *seg_off,
                            const long long *nb_off, const long long *blk_line_base, const unsigned *line_off,
                            unsigned *conn_cnt) {
  hipLaunchKernelGGL(k_node_conn_count, dim3(nblk2(G + 1, 256)), dim3(256), 0, st, G, node_img, seg_off, nb_off,
                     blk_line_base, line_off, conn_cnt);
}
void launch_build_rowlist(hipStream_t st, long long G, const int *node_img, const long long *seg_off,
                          const long long *nb_off, const long long *blk_line_base, const unsigned *line_off,
                          const long long *conn_off, unsigned *srows) {
  if (G > 0)
    hipLaunchKernelGGL(k_build_rowlist, dim3(nblk2(G * 64, 256)), dim3(256), 0, st, G, node_img, seg_off, nb_off,
                       blk_line_base, line_off, conn_off, srows);
}
void launch_gen_rows(hipStream_t st, long long P, int n_blk, long long max_rows, const GenCfg &cfg,
                     const long long *m_off, const int *m_pairs, const int *blk_img, const int *blk_nb,
                     const int *blk_slot, const long long *seg_off, const Cam *cams, const Seg *segs,
                     const PairRec *pairs, Cand *st_c, CandLite *st_l, unsigned char *flag8, unsigned *n_tris) {
  if (P <= 0 || n_blk <= 0 || max_rows <= 0) return;
  const long long rows_per_wg = 4ll * 64 * kGenChunks;
  hipLaunchKernelGGL(k_gen_rows, dim3(nblk2(max_rows, (int)rows_per_wg), n_blk), dim3(256), 0, st, cfg, m_off,
                     m_pairs, blk_img, blk_nb, blk_slot, seg_off, cams, segs, pairs, st_c, st_l, flag8, n_tris);
}
void launch_node_fill(hipStream_t st, long long G, const long long *conn_off, const unsigned *srows,
                      const unsigned char *flag8, const long long *tri_off, const Cand *st_c, const CandLite *st_l,
                      Cand *cand, CandLite *lite, unsigned *cand_node) {
  if (G > 0)
    hipLaunchKernelGGL(k_node_fill, dim3(nblk2(G * 64, 256)), dim3(256), 0, st, G, conn_off, srows, flag8, tri_off,
                       st_c, st_l, cand, lite, cand_node);
}
void launch_cand_node(hipStream_t st, long long G, const long long *tri_off, unsigned *cand_node) {
  if (G > 0)
    hipLaunchKernelGGL(k_cand_node, dim3(nblk2(G * 64, 256)), dim3(256), 0, st, G, tri_off, cand_node);
}
size_t score3_lds_bytes(int max_nb) {
  return 9 * kWin * 8 + (size_t)max_nb * 64 * 8 + 64 * 8 + kWin * 4 + kSQCap * 4;
}
void launch_score3(hipStream_t st, long long C, long long G, const long long *tri_off, const unsigned *cand_node,
                   const Cand *cand, const CandLite *lite, const int *node_img, const long long *nb_off,
                   const int *blk_order, const Cam *cams, double *score, unsigned long long *pair_counter,
                   int max_nb, const ScoreCfg &cfg, double scaleinv_guard2) {
  if (C <= 0) return;
  Score3Args a;
  a.G = G; a.tri_off = tri_off; a.cand_node = cand_node; a.cand = cand; a.lite = lite; a.node_img = node_img;
  a.nb_off = nb_off; a.blk_order = blk_order; a.cams = cams; a.score = score; a.pair_counter = pair_counter;
  a.max_nb = max_nb;
  hipLaunchKernelGGL(k_score3, dim3(nblk2(C, 64)), dim3(64), score3_lds_bytes(max_nb), st, a, cfg, scaleinv_guard2);
}

}  // namespace lt
