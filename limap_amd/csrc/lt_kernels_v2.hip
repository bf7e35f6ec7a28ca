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
//   k_score2
//       HOT LOOP 2, candidate-major: lane = candidate (nodes packed densely into waves), sweep over
//       the candidates of the lane's own node with a two-level conservative early exit (cosine, then
//       squared scale-invariant endpoint distance), survivors evaluated densely from an LDS queue.
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
        o.l.nb_slot = slot;
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
// HOT LOOP 2, candidate-major
// ---------------------------------------------------------------------------------------------
constexpr int kSQCap = 192;

struct Score2Args {
  long long C_cap;                 // launch covers candidates [0, C_cap); C = *c_total
  const long long *tri_off;        // tri_off[G] = C
  long long G;
  const unsigned *cand_node;
  const Cand *cand;
  const CandLite *lite;
  const int *node_img;
  const long long *nb_off;
  const int *blk_nb;
  const int *blk_order;
  const long long *seg_off;
  const Seg *segs;
  const Cam *cams;
  double *score;
  int max_nb;
};

__global__ void __launch_bounds__(256)
k_score2(Score2Args a, ScoreCfg cfg, double scaleinv_guard2) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  const int wave = threadIdx.x >> 6;
  const int lane = lane_id();
  // per-wave LDS: S[max_nb][64] u64 | q[kSQCap] u32 | woff[64] i64
  const size_t per_wave = (size_t)a.max_nb * 64 * 8 + kSQCap * 4 + 64 * 8;
  unsigned char *base = smem_raw + per_wave * wave;
  unsigned long long *S = reinterpret_cast<unsigned long long *>(base);
  unsigned *queue = reinterpret_cast<unsigned *>(base + (size_t)a.max_nb * 64 * 8);
  long long *woff = reinterpret_cast<long long *>(base + (size_t)a.max_nb * 64 * 8 + kSQCap * 4);

  const long long C = a.tri_off[a.G];
  const long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x);
  const long long wave_first = i - lane;
  if (wave_first >= C) return;
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
    sloti = li.nb_slot;
    six = ci.s[0]; siy = ci.s[1]; siz = ci.s[2];
    eix = ci.e[0]; eiy = ci.e[1]; eiz = ci.e[2];
    // conservative bound of the one-way scale-invariant endpoint gate (line_dists.cc:55-60):
    // dist / (depth + eps) > th_scaleinv (1 + 1e-6)  can never score >= score_th
    double zs = ci.depth[0] + kEps, ze = ci.depth[1] + kEps;
    gs2 = scaleinv_guard2 * zs * zs;
    ge2 = scaleinv_guard2 * ze * ze;
    if (!(zs > 0.0)) gs2 = 1e300;  // non-positive depth: leave the decision to the exact path
    if (!(ze > 0.0)) ge2 = 1e300;
  }
  woff[lane] = off;
  for (int k = 0; k < a.max_nb; ++k) S[k * 64 + lane] = 0ull;
  int nmax = n;
  for (int d = 32; d >= 1; d >>= 1) nmax = max(nmax, __shfl_xor(nmax, d));
  int qn = 0;
  wave_lds_sync();

  auto drain = [&]() {
    wave_lds_sync();
    for (int q0 = 0; q0 < qn; q0 += 64) {
      int p = q0 + lane;
      if (p < qn) {
        unsigned e = queue[p];
        int il = (int)(e >> 26);
        long long oi = woff[il];
        long long j = oi + (long long)(e & 0x3FFFFFFu);
        long long ii = wave_first + il;
        const Cand ci = a.cand[ii];
        const CandLite li = a.lite[ii];
        double ti[11] = {ci.s[0], ci.s[1], ci.s[2], ci.e[0], ci.e[1], ci.e[2], li.dir[0], li.dir[1], li.dir[2],
                         ci.depth[0], ci.depth[1]};
        const CandLite lj = a.lite[j];
        const Cand cj = a.cand[j];
        const unsigned gi = a.cand_node[ii];
        const long long nbi = a.nb_off[a.node_img[gi]];
        const int imgj = a.blk_nb[nbi + lj.nb_slot];
        double sc = pair_score(cfg, ti, cj, a.cams[imgj], a.segs[a.seg_off[imgj] + lj.ng_line]);
        if (sc > 0.0) atomicMax(&S[lj.nb_slot * 64 + il], (unsigned long long)__double_as_longlong(sc));
      }
    }
    qn = 0;
    wave_lds_sync();
  };

  for (int jj = 0; jj < nmax; ++jj) {
    bool pass = active && (jj < n);
    if (pass) {
      const long long j = off + jj;
      const CandLite lj = a.lite[j];
      pass = (j != i) && (lj.nb_slot != sloti);
      if (pass) {
        double c = fabs((dix * lj.dir[0] + diy * lj.dir[1]) + diz * lj.dir[2]);
        pass = !(c < cfg.cos_guard);
        if (pass) {
          const Cand cj = a.cand[j];
          double ax = six - cj.s[0], ay = siy - cj.s[1], az = siz - cj.s[2];
          double bx = eix - cj.e[0], by = eiy - cj.e[1], bz = eiz - cj.e[2];
          double ds2 = ax * ax + ay * ay + az * az, de2 = bx * bx + by * by + bz * bz;
          pass = !(ds2 > gs2) && !(de2 > ge2);
        }
      }
    }
    unsigned long long m = __ballot(pass);
    if (m) {
      if (pass) queue[qn + __popcll(m & lanemask_lt())] = ((unsigned)lane << 26) | (unsigned)jj;
      qn += __popcll(m);
      if (qn > kSQCap - 64) drain();
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
size_t score2_lds_bytes(int max_nb) { return 4 * ((size_t)max_nb * 64 * 8 + kSQCap * 4 + 64 * 8); }
void launch_score2(hipStream_t st, long long C_cap, long long G, const long long *tri_off, const unsigned *cand_node,
                   const Cand *cand, const CandLite *lite, const int *node_img, const long long *nb_off,
                   const int *blk_nb, const int *blk_order, const long long *seg_off, const Seg *segs,
                   const Cam *cams, double *score, int max_nb, const ScoreCfg &cfg, double scaleinv_guard2) {
  if (C_cap <= 0) return;
  Score2Args a;
  a.C_cap = C_cap; a.tri_off = tri_off; a.G = G; a.cand_node = cand_node; a.cand = cand; a.lite = lite;
  a.node_img = node_img; a.nb_off = nb_off; a.blk_nb = blk_nb; a.blk_order = blk_order; a.seg_off = seg_off;
  a.segs = segs; a.cams = cams; a.score = score; a.max_nb = max_nb;
  hipLaunchKernelGGL(k_score2, dim3(nblk2(C_cap, 256)), dim3(256), score2_lds_bytes(max_nb), st, a, cfg,
                     scaleinv_guard2);
}

}  // namespace lt
