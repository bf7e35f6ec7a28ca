// lt_kernels_v2.hip -- second-generation kernels of the matched-mode pipeline and of the scoring
// stage (see DESIGN.md section 3 for the map; lt_kernels.hip keeps the invariant builders, the
// generic radix-sort grouping, the exhaustive generation and the selection kernels).
//
//   k_gen_rows
//       HOT LOOP 1 in row (block) order: coalesced match rows, the neighbour's segment table staged
//       in LDS, stage A = cheap gates on every row, stage B = triangulation etc. only for the
//       survivors, which are first gathered in an LDS queue so that the expensive path runs on full
//       wave64s.  Valid candidates are appended in row order to per-wave lists.
//   k_node_prefix + k_place (rows of every block sorted by line id -- what limap's matchers write)
//       Sort-free placement into the reference's candidate order (neighbour-ascending, then
//       match-row order -- base_line_triangulator.cc:71-103): per-(block, line) counts -> per-node
//       prefix over the neighbour blocks -> final position of every candidate.
//   k_pack_keys + radix sort + k_permute (generic rows): stable sort of the candidates by node.
//   k_score3
//       HOT LOOP 2, candidate-major: lane = candidate (nodes packed densely into waves), LDS-staged
//       sweep over the candidates of the lane's own node with a two-level conservative early exit
//       (cosine, then squared scale-invariant endpoint distance), survivors evaluated densely from
//       an LDS queue.
// Compiled with -ffp-contract=off (see lt_geom.h).

#include "lt_devfn.h"

namespace lt {

// ---------------------------------------------------------------------------------------------
// HOT LOOP 1, row order (triangulateOneNode, base_line_triangulator.cc:161-337)
// ---------------------------------------------------------------------------------------------
// grid.y = neighbour block b = (image, neighbour): the image pair, F, baseline and both cameras are
// wave-uniform (scalar registers).  grid.x * 4 waves * kGenChunks * 64 rows cover the block's rows.
// The gate fields of the neighbour's 2D segments (plane normal, line coordinates, endpoints: 80 B)
// are staged once per workgroup in LDS (SoA), since every row of the block indexes that one table.
//   stage A: cheap gates on every row (gen_gates_fast); survivors are queued (ballot + popcount).
//   stage B: triangulation / cheirality / sensitivity / uncertainty / ranges on full wave64s.
// Valid candidates are appended IN ROW ORDER to the wave's own list st_*[r0 ...] (r0 = first row of
// the wave, capacity = rows per wave); wave_count[] holds the list lengths.  In the fast path the
// number of valid candidates per (block, line) run is counted in cnt_bl for the placement pass.
constexpr int kGenChunks = 8;   // 64-row chunks per wave
constexpr int kGenQCap = 128;   // queue entries per wave (drained whenever >= 64)
constexpr int kRowsPerWave = 64 * kGenChunks;
constexpr int kRowsPerWG = 4 * kRowsPerWave;

struct GenArgs {
  const long long *m_off;
  const int *m_pairs;
  const int *blk_img, *blk_nb, *blk_slot;
  const long long *seg_off;
  const Cam *cams;
  const Seg *segs;
  const PairRec *pairs;
  const long long *blk_line_base;
  Cand *st_c;
  CandLite *st_l;
  unsigned *st_key;       // node id of every staged candidate
  unsigned *wave_count;   // [n_blk * gridDim.x * 4]
  unsigned *cnt_bl;       // valid candidates per (block, line) or nullptr (generic path)
  int lds_segs;           // capacity (in segments) of the LDS table; 0 = read the gate fields from HBM/L2
};

__global__ void __launch_bounds__(256)
k_gen_rows(GenArgs a, GenCfg cfg) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  unsigned *q_all = reinterpret_cast<unsigned *>(smem_raw);          // [4][kGenQCap]
  double *T = reinterpret_cast<double *>(smem_raw + 4 * kGenQCap * 4);  // [10][lds_segs]
  const int wave = threadIdx.x >> 6;
  const int lane = lane_id();
  const int b = blockIdx.y;
  const long long rb = a.m_off[b], re = a.m_off[b + 1];
  const int i1 = a.blk_img[b], i2 = a.blk_nb[b], slot = a.blk_slot[b];
  const long long g1 = a.seg_off[i1], g2 = a.seg_off[i2];
  const int M2 = (int)(a.seg_off[i2 + 1] - g2);
  const long long wg_r0 = rb + (long long)blockIdx.x * kRowsPerWG;
  const unsigned lin = ((unsigned)b * gridDim.x + blockIdx.x) * 4u + (unsigned)wave;
  if (wg_r0 >= re) {  // nothing for this workgroup (grid.x is sized by the largest block)
    if (lane == 0) a.wave_count[lin] = 0;
    return;
  }
  const bool use_lds = a.lds_segs >= M2;
  const int ts = a.lds_segs;  // table stride
  if (use_lds) {
    for (int s = threadIdx.x; s < M2; s += blockDim.x) {
      const Seg &sg = a.segs[g2 + s];
      T[0 * ts + s] = sg.n[0]; T[1 * ts + s] = sg.n[1]; T[2 * ts + s] = sg.n[2];
      T[3 * ts + s] = sg.lc[0]; T[4 * ts + s] = sg.lc[1]; T[5 * ts + s] = sg.lc[2];
      T[6 * ts + s] = sg.x1; T[7 * ts + s] = sg.y1; T[8 * ts + s] = sg.x2; T[9 * ts + s] = sg.y2;
    }
    __syncthreads();
  }
  const long long r0 = wg_r0 + (long long)wave * kRowsPerWave;
  unsigned wcount = 0;
  if (r0 < re) {
    const PairRec *pr = a.pairs + b;
    const long long lbase = a.cnt_bl ? a.blk_line_base[b] : 0;
    unsigned *qr = q_all + wave * kGenQCap;
    int qn = 0;

    auto stage_b = [&](int count) {  // lanes 0..count-1 finish one surviving connection each
      bool ok = false;
      GenOut o;
      int line = 0;
      if (lane < count) {
        unsigned r = qr[lane];
        line = a.m_pairs[2 * (long long)r];
        int ng = a.m_pairs[2 * (long long)r + 1];
        ok = gen_finish(cfg, a.cams[i1], a.cams[i2], a.segs[g1 + line], a.segs[g2 + ng], pr->B, &o);
        o.l.nb_slot = lite_pack(slot, i2);
        o.l.ng_line = ng;
      }
      unsigned long long m = __ballot(ok);
      if (ok) {
        long long p = r0 + wcount + __popcll(m & lanemask_lt());
        a.st_c[p] = o.c;
        a.st_l[p] = o.l;
        a.st_key[p] = (unsigned)(g1 + line);
        if (a.cnt_bl) atomicAdd(&a.cnt_bl[lbase + line], 1u);
      }
      wcount += (unsigned)__popcll(m);
    };

    // Software pipeline over the 64-row chunks: the match rows are fetched two chunks ahead and the
    // image's own segment (gather by line id) one chunk ahead, so that neither global round trip sits
    // on the critical path of the gate arithmetic.
    auto load_rows = [&](int c, int *line, int *ng) {
      long long r = r0 + 64ll * c + lane;
      *line = -1;
      *ng = 0;
      if (c < kGenChunks && r < re) {
        const int2 v = *reinterpret_cast<const int2 *>(a.m_pairs + 2 * r);
        *line = v.x;
        *ng = v.y;
      }
    };
    struct S1 { double x1, y1, x2, y2, rs[3], re[3]; };
    auto load_s1 = [&](int line, S1 *o) {
      if (line >= 0) {
        const Seg &s = a.segs[g1 + line];
        o->x1 = s.x1; o->y1 = s.y1; o->x2 = s.x2; o->y2 = s.y2;
        o->rs[0] = s.rs[0]; o->rs[1] = s.rs[1]; o->rs[2] = s.rs[2];
        o->re[0] = s.re[0]; o->re[1] = s.re[1]; o->re[2] = s.re[2];
      }
    };
    int line_c, ng_c, line_n, ng_n, line_nn = -1, ng_nn = 0;
    S1 cur, nxt;
    load_rows(0, &line_c, &ng_c);
    load_rows(1, &line_n, &ng_n);
    load_s1(line_c, &cur);
    for (int c = 0; c < kGenChunks; ++c) {
      load_rows(c + 2, &line_nn, &ng_nn);
      load_s1(line_n, &nxt);
      const long long r = r0 + 64ll * c + lane;
      bool pass = false;
      if (line_c >= 0) {
        const int ng = ng_c;
        if (use_lds) {
          pass = gen_gates_fast(cfg, cur.x1, cur.y1, cur.x2, cur.y2, cur.rs, cur.re, T[0 * ts + ng], T[1 * ts + ng],
                                T[2 * ts + ng], T[3 * ts + ng], T[4 * ts + ng], T[5 * ts + ng], T[6 * ts + ng],
                                T[7 * ts + ng], T[8 * ts + ng], T[9 * ts + ng], pr->F, a.segs[g2 + ng]);
        } else {
          const Seg &s2 = a.segs[g2 + ng];
          pass = gen_gates_fast(cfg, cur.x1, cur.y1, cur.x2, cur.y2, cur.rs, cur.re, s2.n[0], s2.n[1], s2.n[2],
                                s2.lc[0], s2.lc[1], s2.lc[2], s2.x1, s2.y1, s2.x2, s2.y2, pr->F, s2);
        }
      }
      cur = nxt;
      line_c = line_n; ng_c = ng_n;
      line_n = line_nn; ng_n = ng_nn;
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
  if (lane == 0) a.wave_count[lin] = wcount;
}

// Fast path: exclusive prefix of cnt_bl over the neighbour blocks of every node (in place) and the
// node's candidate count.
__global__ void k_node_prefix(long long G, const int *__restrict__ node_img,
                              const long long *__restrict__ seg_off, const long long *__restrict__ nb_off,
                              const long long *__restrict__ blk_line_base, unsigned *__restrict__ cnt_bl,
                              unsigned *__restrict__ n_tris) {
  long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (g > G) return;
  unsigned run = 0;
  if (g < G) {
    int img = node_img[g];
    int line = (int)(g - seg_off[img]);
    for (long long b = nb_off[img]; b < nb_off[img + 1]; ++b) {
      long long e = blk_line_base[b] + line;
      unsigned c = cnt_bl[e];
      cnt_bl[e] = run;
      run += c;
    }
  }
  n_tris[g] = run;
}

// Fast path: move every staged candidate to its final, reference-ordered position
//   pos = tri_off[node] + (valid candidates of the node in earlier neighbour blocks) + rank in its run.
// Same 2D grid as k_gen_rows; one wave per generation wave.  Rows of a block are sorted by line id,
// so the candidates of one (block, line) run are adjacent in the row-ordered lists; the rank is
// found by looking back over equal keys (crossing into the previous wave's list if the run does).
__global__ void __launch_bounds__(256)
k_place(const long long *__restrict__ m_off, const int *__restrict__ blk_img,
        const long long *__restrict__ seg_off, const long long *__restrict__ blk_line_base,
        const unsigned *__restrict__ base_bl, const unsigned *__restrict__ wave_count,
        const long long *__restrict__ tri_off, const Cand *__restrict__ st_c,
        const CandLite *__restrict__ st_l, const unsigned *__restrict__ st_key, Cand *__restrict__ cand,
        CandLite *__restrict__ lite, unsigned *__restrict__ cand_node) {
  const int wave = threadIdx.x >> 6;
  const int lane = lane_id();
  const int b = blockIdx.y;
  const long long rb = m_off[b], re = m_off[b + 1];
  const long long r0 = rb + (long long)blockIdx.x * kRowsPerWG + (long long)wave * kRowsPerWave;
  if (r0 >= re) return;
  const unsigned lin = ((unsigned)b * gridDim.x + blockIdx.x) * 4u + (unsigned)wave;
  const unsigned count = wave_count[lin];
  if (count == 0) return;
  const long long g1 = seg_off[blk_img[b]];
  const long long lbase = blk_line_base[b];
  for (unsigned e0 = 0; e0 < count; e0 += 64) {
    unsigned e = e0 + lane;
    if (e >= count) break;
    const unsigned key = st_key[r0 + e];
    // rank within the (block, line) run
    unsigned rank = 0;
    {
      long long cur_r0 = r0;
      long long idx = (long long)e - 1;
      unsigned cur_lin = lin;
      while (true) {
        while (idx >= 0 && st_key[cur_r0 + idx] == key) {
          ++rank;
          --idx;
        }
        if (idx >= 0) break;                 // a different key precedes: run starts inside this list
        if (cur_r0 - kRowsPerWave < rb) break;  // first wave of the block
        cur_r0 -= kRowsPerWave;
        cur_lin -= 1;
        unsigned pc = wave_count[cur_lin];
        if (pc == 0) {
          // an empty list: the run can only continue further back if that whole wave range belongs
          // to the same line, which an empty list cannot tell -- keep walking (bounded by the block)
          idx = -1;
          continue;
        }
        idx = (long long)pc - 1;
      }
    }
    const long long pos = tri_off[key] + base_bl[lbase + (long long)(key - g1)] + rank;
    cand[pos] = st_c[r0 + e];
    lite[pos] = st_l[r0 + e];
    cand_node[pos] = key;
  }
}

// Generic path: pack the row-ordered wave lists into dense (key, source index) arrays for the sort
__global__ void __launch_bounds__(256)
k_pack_keys(const long long *__restrict__ m_off, const unsigned *__restrict__ wave_count,
            const long long *__restrict__ wave_pos, const unsigned *__restrict__ st_key,
            unsigned *__restrict__ keys_c, unsigned *__restrict__ src_c) {
  const int wave = threadIdx.x >> 6;
  const int lane = lane_id();
  const int b = blockIdx.y;
  const long long rb = m_off[b], re = m_off[b + 1];
  const long long r0 = rb + (long long)blockIdx.x * kRowsPerWG + (long long)wave * kRowsPerWave;
  if (r0 >= re) return;
  const unsigned lin = ((unsigned)b * gridDim.x + blockIdx.x) * 4u + (unsigned)wave;
  const unsigned count = wave_count[lin];
  const long long base = wave_pos[lin];
  for (unsigned e = lane; e < count; e += 64) {
    keys_c[base + e] = st_key[r0 + e];
    src_c[base + e] = (unsigned)(r0 + e);
  }
}

// Generic path: gather the candidates into sorted (node-major, stable) order
__global__ void k_permute(long long C, const unsigned *__restrict__ skeys, const unsigned *__restrict__ ssrc,
                          const Cand *__restrict__ st_c, const CandLite *__restrict__ st_l,
                          Cand *__restrict__ cand, CandLite *__restrict__ lite, unsigned *__restrict__ cand_node) {
  long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= C) return;
  unsigned src = ssrc[t];
  cand[t] = st_c[src];
  lite[t] = st_l[src];
  cand_node[t] = skeys[t];
}

// Per-candidate record for the scoring kernel: where its node's candidates start, how many there
// are, and the neighbour table of its image -- so that the scoring prologue is ONE load level
// instead of the chain cand_node -> tri_off / node_img -> nb_off.
struct CandMeta {
  unsigned off_lo, off_hi;  // tri_off[node] (64-bit split)
  unsigned n;               // candidates of the node
  unsigned nb;              // (nb_off[img] << 8) | number of neighbours  (nb_off < 2^24)
};
__global__ void k_cand_meta(long long C, const unsigned *__restrict__ cand_node,
                            const long long *__restrict__ tri_off, const int *__restrict__ node_img,
                            const long long *__restrict__ nb_off, CandMeta *__restrict__ meta) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= C) return;
  const unsigned g = cand_node[i];
  const long long off = tri_off[g];
  const int img = node_img[g];
  const long long nb0 = nb_off[img];
  CandMeta m;
  m.off_lo = (unsigned)(off & 0xFFFFFFFFll);
  m.off_hi = (unsigned)(off >> 32);
  m.n = (unsigned)(tri_off[g + 1] - off);
  m.nb = ((unsigned)nb0 << 8) | (unsigned)(nb_off[img + 1] - nb0);
  meta[i] = m;
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
  const CandMeta *meta;
  const Cand *cand;
  const CandLite *lite;
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
  // LDS: W[9][kWin] f64 | S[max_nb][64] u64 | woff[64] i64 | wslot[kWin] i32 | queue[kSQCap] u32 | ord[max_nb] i32
  double *W = reinterpret_cast<double *>(smem_raw);
  unsigned long long *S = reinterpret_cast<unsigned long long *>(smem_raw + 9 * kWin * 8);
  long long *woff = reinterpret_cast<long long *>(smem_raw + 9 * kWin * 8 + (size_t)a.max_nb * 64 * 8);
  int *wslot = reinterpret_cast<int *>(smem_raw + 9 * kWin * 8 + (size_t)a.max_nb * 64 * 8 + 64 * 8);
  unsigned *queue = reinterpret_cast<unsigned *>(smem_raw + 9 * kWin * 8 + (size_t)a.max_nb * 64 * 8 + 64 * 8 + kWin * 4);
  int *ordl = reinterpret_cast<int *>(smem_raw + 9 * kWin * 8 + (size_t)a.max_nb * 64 * 8 + 64 * 8 + kWin * 4 + kSQCap * 4);

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
    const CandMeta mt = a.meta[i];
    off = ((long long)mt.off_hi << 32) | (long long)mt.off_lo;
    n = (int)mt.n;
    nb0 = (long long)(mt.nb >> 8);
    n_nb = (int)(mt.nb & 0xFFu);
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
  // summation order of the first lane's image, staged once (lanes of another image read it from HBM)
  const long long wave_nb0 = __shfl(nb0, 0);
  if (lane < __shfl(n_nb, 0)) ordl[lane] = a.blk_order[wave_nb0 + lane];
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
        // one LDS round trip per iteration: fetch every field up front, then test
        const int w = w0 + t;
        const int sl = wslot[w];
        const double jx = W[0 * kWin + w], jy = W[1 * kWin + w], jz = W[2 * kWin + w];
        const double sx = W[3 * kWin + w], sy = W[4 * kWin + w], sz = W[5 * kWin + w];
        const double ex = W[6 * kWin + w], ey = W[7 * kWin + w], ez = W[8 * kWin + w];
        const double c = fabs((dix * jx + diy * jy) + diz * jz);
        const double ax = six - sx, ay = siy - sy, az = siz - sz;
        const double bx = eix - ex, by = eiy - ey, bz = eiz - ez;
        const double ds2 = ax * ax + ay * ay + az * az, de2 = bx * bx + by * by + bz * bz;
        // below the cosine guard the 3D angle score is certainly gated to 0; beyond the squared
        // distance guards the scale-invariant endpoint score is
        pass = (jlo + t != i) && (sl != sloti) && !(c < cfg.cos_guard) && !(ds2 > gs2) && !(de2 > ge2);
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
    const bool own = nb0 == wave_nb0;
    for (int r = 0; r < n_nb; ++r) {
      int k = own ? ordl[r] : a.blk_order[nb0 + r];
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

unsigned gen_grid_x(long long max_rows) { return nblk2(max_rows, kRowsPerWG); }
size_t gen_lds_bytes(int lds_segs) { return 4 * kGenQCap * 4 + (size_t)lds_segs * 80; }
void launch_gen_rows(hipStream_t st, int n_blk, long long max_rows, const GenCfg &cfg, const long long *m_off,
                     const int *m_pairs, const int *blk_img, const int *blk_nb, const int *blk_slot,
                     const long long *seg_off, const Cam *cams, const Seg *segs, const PairRec *pairs,
                     const long long *blk_line_base, Cand *st_c, CandLite *st_l, unsigned *st_key,
                     unsigned *wave_count, unsigned *cnt_bl, int lds_segs) {
  if (n_blk <= 0 || max_rows <= 0) return;
  GenArgs a;
  a.m_off = m_off; a.m_pairs = m_pairs; a.blk_img = blk_img; a.blk_nb = blk_nb; a.blk_slot = blk_slot;
  a.seg_off = seg_off; a.cams = cams; a.segs = segs; a.pairs = pairs; a.blk_line_base = blk_line_base;
  a.st_c = st_c; a.st_l = st_l; a.st_key = st_key; a.wave_count = wave_count; a.cnt_bl = cnt_bl;
  a.lds_segs = lds_segs;
  hipLaunchKernelGGL(k_gen_rows, dim3(gen_grid_x(max_rows), n_blk), dim3(256), gen_lds_bytes(lds_segs), st, a, cfg);
}
void launch_node_prefix(hipStream_t st, long long G, const int *node_img, const long long *seg_off,
                        const long long *nb_off, const long long *blk_line_base, unsigned *cnt_bl,
                        unsigned *n_tris) {
  hipLaunchKernelGGL(k_node_prefix, dim3(nblk2(G + 1, 256)), dim3(256), 0, st, G, node_img, seg_off, nb_off,
                     blk_line_base, cnt_bl, n_tris);
}
void launch_place(hipStream_t st, int n_blk, long long max_rows, const long long *m_off, const int *blk_img,
                  const long long *seg_off, const long long *blk_line_base, const unsigned *base_bl,
                  const unsigned *wave_count, const long long *tri_off, const Cand *st_c, const CandLite *st_l,
                  const unsigned *st_key, Cand *cand, CandLite *lite, unsigned *cand_node) {
  if (n_blk <= 0 || max_rows <= 0) return;
  hipLaunchKernelGGL(k_place, dim3(gen_grid_x(max_rows), n_blk), dim3(256), 0, st, m_off, blk_img, seg_off,
                     blk_line_base, base_bl, wave_count, tri_off, st_c, st_l, st_key, cand, lite, cand_node);
}
void launch_pack_keys(hipStream_t st, int n_blk, long long max_rows, const long long *m_off,
                      const unsigned *wave_count, const long long *wave_pos, const unsigned *st_key,
                      unsigned *keys_c, unsigned *src_c) {
  if (n_blk <= 0 || max_rows <= 0) return;
  hipLaunchKernelGGL(k_pack_keys, dim3(gen_grid_x(max_rows), n_blk), dim3(256), 0, st, m_off, wave_count, wave_pos,
                     st_key, keys_c, src_c);
}
void launch_permute(hipStream_t st, long long C, const unsigned *skeys, const unsigned *ssrc, const Cand *st_c,
                    const CandLite *st_l, Cand *cand, CandLite *lite, unsigned *cand_node) {
  if (C > 0)
    hipLaunchKernelGGL(k_permute, dim3(nblk2(C, 256)), dim3(256), 0, st, C, skeys, ssrc, st_c, st_l, cand, lite,
                       cand_node);
}
void launch_cand_node(hipStream_t st, long long G, const long long *tri_off, unsigned *cand_node) {
  if (G > 0)
    hipLaunchKernelGGL(k_cand_node, dim3(nblk2(G * 64, 256)), dim3(256), 0, st, G, tri_off, cand_node);
}
size_t score3_lds_bytes(int max_nb) {
  return 9 * kWin * 8 + (size_t)max_nb * 64 * 8 + 64 * 8 + kWin * 4 + kSQCap * 4 + (size_t)max_nb * 4;
}
size_t cand_meta_bytes() { return sizeof(CandMeta); }
void launch_score3(hipStream_t st, long long C, long long G, const long long *tri_off, const unsigned *cand_node,
                   void *meta, const Cand *cand, const CandLite *lite, const int *node_img, const long long *nb_off,
                   const int *blk_order, const Cam *cams, double *score, unsigned long long *pair_counter,
                   int max_nb, const ScoreCfg &cfg, double scaleinv_guard2) {
  if (C <= 0) return;
  hipLaunchKernelGGL(k_cand_meta, dim3(nblk2(C, 256)), dim3(256), 0, st, C, cand_node, tri_off, node_img, nb_off,
                     reinterpret_cast<CandMeta *>(meta));
  Score3Args a;
  a.G = G; a.tri_off = tri_off; a.meta = reinterpret_cast<const CandMeta *>(meta); a.cand = cand; a.lite = lite;
  a.blk_order = blk_order; a.cams = cams; a.score = score; a.pair_counter = pair_counter;
  a.max_nb = max_nb;
  hipLaunchKernelGGL(k_score3, dim3(nblk2(C, 64)), dim3(64), score3_lds_bytes(max_nb), st, a, cfg, scaleinv_guard2);
}

}  // namespace lt
