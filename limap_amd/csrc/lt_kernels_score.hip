// lt_kernels_score.hip -- the scoring stage (scoreOneNode, global_line_triangulator.cc:71-116): k_cand_meta (prologue
// records, tiles by cost class), k_depth_order (exhaustive mode), and scoreOneNode itself in two forms:
//   split (matched mode, round 4): k_score3<.., kSplit = true> sweeps (conservative single-precision guards over an LDS
//         window) and leaves the pairs that pass in per-tile slots in HBM, k_dense8 evaluates them (pair_score, per-image
//         maxima, ordered sums) in units of tiles with workgroups of four waves;
//   fused (exhaustive mode; the fallback when the split form's overflow store fills; LT_SCORE_FUSED=1): k_score3 does both
//         per tile.  Same bits from both (tests/test_gpu_guards.py::test_split_and_fused_scoring_agree).
// A translation unit of its own since round 3; compiled like lt_kernels_v2.hip with -mllvm -disable-machine-licm since round
// 4 (k_dense8: 128 instead of 183 registers; the fused k_score3 is 2.5 % slower for it) and -ffp-contract=off (lt_geom.h).

#include "lt_devfn.h"

#include <hip/hip_ext.h>

#include <algorithm>

namespace lt {

#ifndef LT_SCORE_WIN
#define LT_SCORE_WIN 128
#endif
#ifndef LT_SCORE_RESIDENT
#define LT_SCORE_RESIDENT 16  // persistent k_score3 workgroups (one wave each) per CU, if LDS and registers allow
#endif
#ifdef LT_SCORE_WAVES_PER_EU
#define LT_SCORE_OCC __attribute__((amdgpu_waves_per_eu(LT_SCORE_WAVES_PER_EU, LT_SCORE_WAVES_PER_EU)))
#else
#define LT_SCORE_OCC
#endif
// A wave CLAIMS its next tile (the draw's device atomic) at the current tile's LAST dense round: the round trip hides
// behind one round + the ordered sums, and no wave sits on a tile nobody else can take while the queues run dry
// (measured in round 3 at 100 x 500: claim at the start of the tile 122.7 us, at the last round 118.8, after the sums 124.9).

#ifdef LT_TRACE
// developer build: per-wave timestamps (100 MHz wall clock) of the scoring kernels; lt_debug_read_trace
// (lt_kernels_v2.hip) merges them into its own array's slices 2 and 3
__device__ unsigned long long g_trace[4 * 4 * 65536];
#define LT_TRACE_MARK(kern, id, slot) \
  if (((kern) == 3 ? threadIdx.x == 0 : lane_id() == 0) && (id) < 65536u) g_trace[(kern) * 4 * 65536 + 4 * (id) + (slot)] = wall_clock64()
#else
#define LT_TRACE_MARK(kern, id, slot)
#endif

static inline unsigned nblk2(long long n, int bs) { return (unsigned)((n + bs - 1) / bs); }

// Per-candidate record for the scoring kernel: where its node's candidates start, how many there
// are, and the neighbour table of its image -- so that the scoring prologue is ONE load level
// instead of the chain cand_node -> tri_off / node_img -> nb_off.
struct CandMeta {
  unsigned off_lo, off_hi;  // tri_off[node] (64-bit split)
  unsigned n;               // candidates of the node
  unsigned nb;              // (nb_off[img] << 8) | number of neighbours  (nb_off < 2^24)
};

// Also resets the tile draw counters of the persistent k_score3 that follows, and lists the tiles (64 consecutive
// candidates = what one wave of this kernel handles per iteration) by COST CLASS: a tile's time in k_score3 grows
// with the sizes of the nodes it touches (correlation 0.62 with the sum over its lanes of the node size), and a
// persistent grid finishes earlier when the long tiles start first -- measured on the bench scene: k_score3
// 138 -> 127 us with the tiles in descending cost order.  No sort: per draw queue kTileBuckets lists filled through
// one counter each (zeroed by k_build_pairs), drawn from the most expensive class down; the order inside a class
// is arbitrary.
constexpr int kTileQueues = 8;  // one draw counter per XCD (workgroups are dealt round-robin to the XCDs)
constexpr int kTileBuckets = 32;
__device__ __forceinline__ int tile_bucket(unsigned cost_sum_n) {  // mean node size over the 64 lanes, 2 per class
  const unsigned b = cost_sum_n >> 7;
  return (int)(b < (unsigned)(kTileBuckets - 1) ? b : (unsigned)(kTileBuckets - 1));
}
__global__ void __launch_bounds__(256)
k_cand_meta(long long G, const unsigned *__restrict__ cand_node, const long long *__restrict__ tri_off,
            const int *__restrict__ node_img, const long long *__restrict__ nb_off, CandMeta *__restrict__ meta,
            unsigned *__restrict__ draw, unsigned *__restrict__ bucket_cnt, unsigned *__restrict__ bucket_list,
            unsigned bucket_cap, const uint4 *__restrict__ node_rec, unsigned *__restrict__ pc_cnt,
            uint2 *__restrict__ fifo, unsigned fifo_cap) {
  // grid-stride over the exact candidate count tri_off[G]; the host may only know an upper bound
  const long long C = tri_off[G];
  const long long stride = (long long)gridDim.x * blockDim.x;
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (pc_cnt && i < kTileQueues * kTileBuckets) pc_cnt[i * 32] = 0;  // the sweep's pair-count lists
  if (i < 2 * kTileQueues + 1) draw[i * 32] = 0;  // 128 bytes apart; behind the draw queues: the split form's overflow-chunk counter and k_dense8's eight claim counters
  if (fifo) {
    // k_score_q: behind those the XCD seen by each queue's first workgroup (none yet) and the queues' sweep-claim counters;
    // the FIFOs of finished tiles (queue q: fifo[q * fifo_cap ...], one entry per tile of the queue) start empty
    if (i >= 2 * kTileQueues + 1 && i < 3 * kTileQueues + 1) draw[i * 32] = 0xFFFFFFFFu;
    if (i >= 3 * kTileQueues + 1 && i < 4 * kTileQueues + 1) draw[i * 32] = 0;
    const long long per = ((C + 63) / 64 + kTileQueues - 1) / kTileQueues;
    for (long long j = i; j < per * kTileQueues; j += stride)
      fifo[(size_t)(j / per) * fifo_cap + (size_t)(j % per)] = make_uint2(0xFFFFFFFFu, 0u);
  }
  const long long C_up = (C + 63) & ~63ll;  // whole waves take part in the tile's reduction
  for (; i < C_up; i += stride) {
    unsigned n = 0, w_lo = 0, w_hi = 0;
    if (i < C) {
      const unsigned g = cand_node[i];
      CandMeta m;
      if (node_rec) {  // the node's record as k_node_prefix wrote it: one 16-byte gather
        const uint4 r = node_rec[g];
        m.off_lo = r.x; m.off_hi = r.y; m.n = r.z; m.nb = r.w;
      } else {
        const long long off = tri_off[g];
        const int img = node_img[g];
        const long long nb0 = nb_off[img];
        m.off_lo = (unsigned)(off & 0xFFFFFFFFll);
        m.off_hi = (unsigned)(off >> 32);
        m.n = (unsigned)(tri_off[g + 1] - off);
        m.nb = ((unsigned)nb0 << 8) | (unsigned)(nb_off[img + 1] - nb0);
      }
      meta[i] = m;
      n = m.n;
      w_lo = m.off_lo; w_hi = m.off_lo + m.n;
    }
    if (bucket_cnt) {
      unsigned sum = n;
      for (int d = 32; d >= 1; d >>= 1) sum += (unsigned)__shfl_xor((int)sum, d);
      // the tile's window (natural order): from the node of its first candidate to the end of the node of its last one
      const long long base = i - lane_id();  // < C: the loop runs over whole waves up to C rounded up
      const int last = (int)((C - base) < 64 ? (C - base) : 64) - 1;
      const unsigned t_hi = (unsigned)__builtin_amdgcn_readlane((int)w_hi, last);
      if (lane_id() == 0) {
        // one list per (draw queue, class): 128 counters -- a single counter per class would serialise thousands
        // of device-scope atomics on one address (~15 ns each).  An entry is 16 bytes: the tile and its window bounds,
        // so that k_score3 knows what to stage from the draw alone.
        const unsigned tile = (unsigned)(i >> 6);
        const int qb = (int)(tile & (kTileQueues - 1)) * kTileBuckets + tile_bucket(sum);
        const unsigned idx = atomicAdd(&bucket_cnt[qb * 32], 1u);  // counters 128 bytes apart: one L2 line each
        if (idx < bucket_cap)
          reinterpret_cast<uint4 *>(bucket_list)[(size_t)qb * bucket_cap + idx] = uint4{tile, w_lo, t_hi, 0u};
      }
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
static __device__ __forceinline__ unsigned mt_key(long long off) { return (unsigned)off & 0xFFFFFFu; }
constexpr int kSQCap = 512;  // the queue is drained when fewer than 256 (4 sweep iterations) slots are free
constexpr int kSQCapSplit = 384;  // split form: 8-byte entries, emptied into the tile's slot in HBM
constexpr int kWin = LT_SCORE_WIN;

struct Score3Args {
  long long G;
  const long long *tri_off;  // tri_off[G] = C
  const CandMeta *meta;
  const CRec *cand;
  const int *blk_order;
  const Cam *cams;
  double *score;
  unsigned long long *pair_counter;  // stats: pairs that reached the dense evaluation
  unsigned *draw;                    // kTileQueues draw counters, 128 bytes apart: queue q = tiles q, q + 8, ...
  const unsigned *perm;              // kSorted: candidate record at depth-sorted position t (k_depth_order)
  const unsigned *spos;              // kSorted over staged records: natural position (score index) of sorted position t
  const unsigned *rng;               // kSorted: node-relative range of sorted positions lane t has to sweep (lo | hi << 16; ~0: all)
  const unsigned *bucket_cnt;        // tiles by cost class (k_cand_meta): counts, lists of bucket_cap entries each
  const unsigned *bucket_list;       // entries of four words: tile, first and end position of its window, 0
  unsigned bucket_cap;
  int max_nb;
  int *err_flag;  // device error flag of the run
  // split form (k_score3<.., kSplit> writes the pairs that pass the sweep, k_dense8 evaluates them)
  uint4 *sp_slots;        // tile t: entries [t * sp_slot_cap, ..): x = record of j, y = record of i, z = lane of i in the tile
  unsigned *sp_cnt;       // pairs of tile t (also left in word 3 of the tile's class-list entry: k_dense8 reads that)
  unsigned *sp_ovf;       // first overflow chunk of tile t (valid when sp_cnt[t] > sp_slot_cap)
  uint4 *sp_pairs;        // overflow chunks of kChunkCap entries
  uint2 *sp_desc;         // overflow chunk: x = entries, y = next chunk of the tile or kNoChunk
  unsigned *sp_counters;  // [0] overflow chunks handed out (sweep), [32 (1 + q)] units of queue q claimed (dense)
  unsigned sp_chunk_cap;
  int sp_slot_cap;
  int sp_t_max;           // tiles per unit of k_dense8 (one table of maxima per tile in its LDS)
  int sp_wave_lds;        // LDS bytes of one wave of the sweep kernel
  // round 6: the sweep lists every finished tile by its TRUE pair count (class = pairs / 16, one list per (queue, class) as
  // for the cost classes); k_dense8 takes its units from these lists, heaviest first
  unsigned *pc_cnt;       // kTileQueues * kTileBuckets counters, 128 bytes apart (zeroed by k_cand_meta)
  uint2 *pc_list;         // lists of pc_cap entries: x = tile, y = its pair count
  unsigned pc_cap;
};
__device__ __forceinline__ int pair_bucket(unsigned pairs) {
  const unsigned b = pairs >> 4;
  return (int)(b < (unsigned)(kTileBuckets - 1) ? b : (unsigned)(kTileBuckets - 1));
}

// Split form of the scoring stage (round 4).  The sweep kernel writes the pairs that pass its guards to the SLOT of their
// tile in HBM (sp_slot_cap entries of 16 bytes; what does not fit goes to a chain of overflow chunks handed out through a
// counter); k_dense8 then gives units of sp_t_max consecutive tiles to workgroups of four waves: the rounds of pair_score
// run over the unit's pairs as one list, whatever the tile they came from, and the pairs of a heavy tile are spread over
// four SIMDs instead of running as nine rounds of one wave.
constexpr int kChunkCap = 512;
constexpr int kChunkTiles = 8;
constexpr unsigned kNoChunk = 0xFFFFFFFFu;
constexpr int kErrPairChunks = 7;  // device error flag: the overflow store is full (finish_run repeats the run fused)
constexpr int kDenseHdrBytes = 16 + kChunkTiles * 8 + 256 * 4 + 64 * 4;  // k_dense8's LDS in front of the tables

// Depth order of a node's candidates (large nodes: exhaustive matching gives ~450 candidates per node and
// 1.6e10 ordered pairs per scene, of which the sweep passes 0.05 %).  All candidates of a node are seen from
// the node's own view, and the stored depth of a start point is a linear functional with a unit-norm
// gradient (third row of R) of the point: |z_i - z_j| <= |start_i - start_j|.  A pair whose depths differ by
// more than the scale-invariant guard radius of i can therefore not pass the sweep's distance guard -- in
// depth-sorted order candidate i only has to sweep a contiguous range of positions.  One wave per node:
// bitonic sort of (float depth, index) in registers, then a range for every four positions by bisection.  The float
// keys only steer the pruning (radius widened by their rounding); nodes above kSortMax candidates or with a
// non-finite depth keep the identity order and the full range.  Which pairs reach the dense evaluation is
// unchanged, so is every result (LT_TEST_SCORE_UNSORTED: the plain sweep).
constexpr int kSortMax = 2048;  // the index takes the 11 low bits of the sort word

// Value of lane (lane ^ kX) for the lane distances a blocked bitonic network takes.  Within a row of 16 lanes a DPP move (no
// trip through the LDS pipe, which ds_bpermute is); across rows gfx950's row / half swaps; the two mirrored distances that
// cross rows (31, 63: once per sort each) stay with ds_bpermute.
template <int kX>
static __device__ __forceinline__ unsigned lane_xor(unsigned x, int lane) {
  if constexpr (kX == 1) return (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0xB1, 0xF, 0xF, true);         // quad_perm [1,0,3,2]
  else if constexpr (kX == 2) return (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x4E, 0xF, 0xF, true);    // quad_perm [2,3,0,1]
  else if constexpr (kX == 3) return (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x1B, 0xF, 0xF, true);    // quad_perm [3,2,1,0]
  else if constexpr (kX == 7) return (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x141, 0xF, 0xF, true);   // row_half_mirror
  else if constexpr (kX == 4) return lane_xor<3>(lane_xor<7>(x, lane), lane);
  else if constexpr (kX == 8) return (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x128, 0xF, 0xF, true);   // row_ror:8
  else if constexpr (kX == 15) return (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x140, 0xF, 0xF, true);  // row_mirror
  else if constexpr (kX == 16) {
    const auto p = __builtin_amdgcn_permlane16_swap(x, x, false, false);  // [0] = rows 0,0,2,2 of x; [1] = rows 1,1,3,3
    return (lane & 16) ? p[0] : p[1];
  } else if constexpr (kX == 32) {
    const auto p = __builtin_amdgcn_permlane32_swap(x, x, false, false);  // [0] = lower half twice; [1] = upper half twice
    return (lane & 32) ? p[0] : p[1];
  } else {
    return (unsigned)__shfl_xor((int)x, kX);
  }
}

// Bitonic sort of R * 64 packed 32-bit words (21 key bits | 11 index bits) held R per lane, BLOCKED: element e = lane * R + r.
// A compare-exchange distance below R pairs two registers of the same lane with the direction known at compile time (two
// instructions per pair); a larger one the same register of two lanes.  Every merge opens with the mirrored exchange
// (e against e ^ (k - 1)), so that all comparators point upwards and no lane-dependent direction is left.  No LDS traffic,
// every loop unrolled.  (Round 5's form held element r * 64 + lane in register r: 39 of the 45 steps of 512 elements went
// between lanes, through ds_bpermute; here 21 do, 15 of them as DPP moves.  0.18 ms of k_depth_order then.)
template <int R, int kStep, bool kMirror>
static __device__ __forceinline__ void bitonic_lanes(unsigned (&v)[R], int lane) {
  // exchange with lane ^ kStep; kMirror (the opening exchange of a merge; kStep = 2^m - 1): with its mirrored register.
  // (min / max with the DPP operand folded in by inline assembly -- three instructions per element instead of the
  // compiler's four -- was measured: no difference, the kernel is not bound by its instruction count.)
  const bool lower = (lane & (kMirror ? (kStep + 1) >> 1 : kStep)) == 0;
  unsigned mn[R], mx[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    if constexpr (!kMirror && (kStep == 16 || kStep == 32)) {
      // rows 0,0,2,2 / 1,1,3,3 (halves 0,0 / 1,1) of the word: the two partners, in both their lanes
      const auto p = kStep == 16 ? __builtin_amdgcn_permlane16_swap(v[r], v[r], false, false)
                                 : __builtin_amdgcn_permlane32_swap(v[r], v[r], false, false);
      mn[r] = min(p[0], p[1]); mx[r] = max(p[0], p[1]);
    } else {
      const unsigned o = lane_xor<kStep>(v[kMirror ? R - 1 - r : r], lane);
      mn[r] = min(v[r], o); mx[r] = max(v[r], o);
    }
  }
#pragma unroll
  for (int r = 0; r < R; ++r) v[r] = lower ? mn[r] : mx[r];
}
template <int R, int K>
static __device__ __forceinline__ void bitonic_merge(unsigned (&v)[R], int lane) {
  // merge of sorted runs of K / 2 into runs of K
  if constexpr (K <= R) {
#pragma unroll
    for (int r = 0; r < R; ++r)
      if ((r & (K >> 1)) == 0) {
        const unsigned x = v[r], y = v[r ^ (K - 1)];
        v[r] = min(x, y);
        v[r ^ (K - 1)] = max(x, y);
      }
  } else {
    bitonic_lanes<R, K / R - 1, true>(v, lane);
  }
  if constexpr (K >= 4) {
    // the remaining distances K / 4 ... 1, all upwards
    if constexpr (K / 4 >= 32 * R) bitonic_lanes<R, 32, false>(v, lane);
    if constexpr (K / 4 >= 16 * R) bitonic_lanes<R, 16, false>(v, lane);
    if constexpr (K / 4 >= 8 * R) bitonic_lanes<R, 8, false>(v, lane);
    if constexpr (K / 4 >= 4 * R) bitonic_lanes<R, 4, false>(v, lane);
    if constexpr (K / 4 >= 2 * R) bitonic_lanes<R, 2, false>(v, lane);
    if constexpr (K / 4 >= R) bitonic_lanes<R, 1, false>(v, lane);
#pragma unroll
    for (int j = (K / 4 < R ? K / 4 : R / 2); j > 0; j >>= 1) {
#pragma unroll
      for (int r = 0; r < R; ++r)
        if ((r & j) == 0) {
          const unsigned x = v[r], y = v[r | j];
          v[r] = min(x, y);
          v[r | j] = max(x, y);
        }
    }
  }
}
template <int R, int K>
static __device__ __forceinline__ void bitonic_from(unsigned (&v)[R], int lane) {
  bitonic_merge<R, K>(v, lane);
  if constexpr (K < R * 64) bitonic_from<R, 2 * K>(v, lane);
}
template <int R>
static __device__ __forceinline__ void wave_bitonic(unsigned (&v)[R], int lane) {
  bitonic_from<R, 2>(v, lane);
}

constexpr unsigned kDepthKeyMask = 0xFFFFF800u;
#ifndef LT_DEPTH_GROUP
#define LT_DEPTH_GROUP 4
#endif
constexpr int kDepthGroup = LT_DEPTH_GROUP;  // sorted positions that share a sweep range (power of two)
static __device__ __forceinline__ float depth_key(unsigned w) { return __uint_as_float(w & kDepthKeyMask); }
// first position of the sorted words kw[a0 .. n) whose key is >= t (kStrict: > t)
template <bool kStrict>
static __device__ __forceinline__ int depth_bound(const unsigned *kw, int a, int b, float t) {
  while (a < b) {
    const int m = (a + b) >> 1;
    const float k = depth_key(kw[m]);
    if (kStrict ? (k <= t) : (k < t)) a = m + 1; else b = m;
  }
  return a;
}
// Sorts the node's (depth, index) words, then gives every sorted position its record and its sweep range.  Lane l ends
// up with the sorted positions l * R .. l * R + R - 1 in its registers; the words also go to LDS (kw), where the lanes
// look up each other's keys.  A range is found once per kDepthGroup consecutive positions (from the smallest lower and the
// largest upper threshold among them: a superset of each one's own range, which is all the sweep needs) and written for
// all of them, packed into one word, in a coalesced pass.  (Round 5 bisected twice for every position -- 0.32 ms of the
// kernel's 0.68 -- and wrote two words.)
template <int R>
static __device__ __forceinline__ bool depth_sort_node(const CRec *__restrict__ cand, long long off, int n, int lane,
                                                       unsigned *kw, unsigned *__restrict__ perm,
                                                       const unsigned *__restrict__ place, unsigned *__restrict__ rec,
                                                       const float *__restrict__ st_z, double guard,
                                                       unsigned *__restrict__ rng) {
  // word = the float key with its 11 low mantissa bits replaced by the index (n <= 2048): half the compare-exchange
  // and shuffle work of a (key, index) pair of words; the keys only steer the pruning, the radius below is widened
  // by the 2^-12 the truncation can cost.  (Loaded striped -- coalesced -- and sorted as if blocked: the network does
  // not care where a word starts.)
  unsigned v[R];
  bool bad = false;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int e = r * 64 + lane;
    v[r] = ~0u;  // padding sorts to the end
    if (e < n) {
      // st_z: the same key, already rounded to single precision, from the compact per-slot array of k_tri_ex
      // (k_place_ex scattering the keys to the candidates' positions next to the slot numbers, for a streamed read here:
      // 54 us less here, 88 more there)
      const double z = st_z ? (double)st_z[place[off + e]] : cand[place ? (long long)place[off + e] : off + e].depth[0];
      const float kf = (float)z;
      bad = bad || !(z > 0.0 && z < 1e30);  // non-positive / NaN / inf / absurd depth: no pruning for this node
      v[r] = (__float_as_uint(kf) & kDepthKeyMask) | (unsigned)e;  // positive floats order like their bits
    }
  }
  if (__ballot(bad)) return false;
  wave_bitonic<R>(v, lane);
  const int base = lane * R;
#pragma unroll
  for (int r = 0; r < R; ++r) kw[base + r] = v[r];
  wave_lds_sync();
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int e = r * 64 + lane;
    if (e < n) {
      const long long nat = off + (long long)(kw[e] & 0x7FFu);
      perm[off + e] = (unsigned)nat;
      if (place) rec[off + e] = place[nat];
    }
  }
  constexpr int S = R < kDepthGroup ? R : kDepthGroup;
  unsigned res[R / S];
#pragma unroll
  for (int q = 0; q < R / S; ++q) {
    float lo_v = __builtin_inff(), hi_v = -__builtin_inff();
    bool full = false;
#pragma unroll
    for (int r = q * S; r < q * S + S; ++r) {
      if (base + r < n) {
        const float z = depth_key(v[r]);
        const double zz = (double)z + kEps;
        // radius of the sweep's distance guard for this candidate, widened by what the keys lost: a key is the depth
        // truncated to 13 mantissa bits, k <= z < k (1 + 2^-12), so |k_i - k_j| <= (rad(z_i) + 2.5e-4 k_i)(1 + 2.6e-4)
        const double rad = guard * zz * 1.001 + 3e-4 * (double)z + 1e-30;
        full = full || !(rad < 1e299);
        lo_v = fminf(lo_v, (float)((double)z - rad));
        hi_v = fmaxf(hi_v, (float)((double)z + rad));
      }
    }
    res[q] = 0xFFFFFFFFu;
    if (base + q * S < n && !full) {
      const int a = depth_bound<false>(kw, 0, n, lo_v);  // first position with key >= lo_v
      const int b = depth_bound<true>(kw, a, n, hi_v);   // first position with key > hi_v
      // (float)(z -+ rad) rounds either way: one more position on each side
      res[q] = (unsigned)max(a - 1, 0) | ((unsigned)min(b + 1, n) << 16);
    }
  }
  wave_lds_sync();  // every lane is done with the keys
#pragma unroll
  for (int q = 0; q < R / S; ++q) kw[lane * (R / S) + q] = res[q];
  wave_lds_sync();
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int e = r * 64 + lane;
    if (e < n) rng[off + e] = kw[e / S];
  }
  return true;
}

// One wave per node and per workgroup (four nodes to a workgroup held the LDS of a finished node until the largest of the four
// was done: 428 against 318 us).  Two launches: nodes up to kSortSmall candidates with a quarter of the LDS and half the
// registers of the network for 2 048 -- twice the waves in flight for the bulk of the nodes -- and the rest.
constexpr int kSortSmall = 512;
template <bool kBig>
__global__ void __launch_bounds__(64)
k_depth_order(long long G, const long long *__restrict__ tri_off, const CRec *__restrict__ cand, double guard,
              unsigned *__restrict__ perm, unsigned *__restrict__ rng, const unsigned *__restrict__ place,
              unsigned *__restrict__ rec, const float *__restrict__ st_z) {
  // place != nullptr: the records are staged (one-pass exhaustive mode), the candidate at natural position p is record
  // place[p]; perm then holds the natural position and rec the record of every depth-sorted position
  __shared__ unsigned kw[kBig ? kSortMax : kSortSmall];
  const int lane = lane_id();
  const long long g = (long long)blockIdx.x;
  const long long off = tri_off[g];
  const int n = (int)(tri_off[g + 1] - off);
  if (n <= 0 || (n > kSortSmall) != kBig) return;
  bool sorted = false;
  if constexpr (!kBig) {
    if (n <= 64) sorted = depth_sort_node<1>(cand, off, n, lane, kw, perm, place, rec, st_z, guard, rng);
    else if (n <= 128) sorted = depth_sort_node<2>(cand, off, n, lane, kw, perm, place, rec, st_z, guard, rng);
    else if (n <= 256) sorted = depth_sort_node<4>(cand, off, n, lane, kw, perm, place, rec, st_z, guard, rng);
    else sorted = depth_sort_node<8>(cand, off, n, lane, kw, perm, place, rec, st_z, guard, rng);
  } else {
    if (n <= 1024) sorted = depth_sort_node<16>(cand, off, n, lane, kw, perm, place, rec, st_z, guard, rng);
    else if (n <= 2048) sorted = depth_sort_node<32>(cand, off, n, lane, kw, perm, place, rec, st_z, guard, rng);
  }
  if (!sorted) {
    for (int r = lane; r < n; r += 64) {
      perm[off + r] = (unsigned)(off + r);
      if (place) rec[off + r] = place[off + r];
      rng[off + r] = 0xFFFFFFFFu;  // the whole node
    }
  }
}

// kF32 (default): the sweep's early exit in single precision on coordinates relative to a wave-local origin,
// with the rounding of that form bounded and added to the guards, so that it rejects a subset of what the
// double test rejects, never more: 1e-6 R on a distance (R = largest coordinate magnitude in the window; the
// bound is 4 sqrt(3) 2^-24 R = 4.2e-7 R), 2e-6 on a cosine of unit vectors (bound 3e-7).  NaN / inf compare
// false and fall through to the exact evaluation.  Which pairs reach the dense evaluation changes slightly,
// no result does (tests: LT_TEST_SCORE_F64 = the double-precision sweep, bit-identical outputs).  The window
// is AoS (3 x float4 per candidate: direction + slot, start, end: three 128-bit LDS reads per pair, broadcast
// when the lanes of a node walk in step) and the sweep is unrolled by four with the reads hoisted.
// (A table-free form of the per-image maxima for jobs whose neighbour lists are in ascending image id --
// running (slot, max, sum) per lane in registers, evaluated pairs handed to their owner lanes by ballot +
// readlane in queue order -- was measured: 195 us against 142 us, the serial hand-off costs more than the
// 10 KB of LDS it frees.  Evaluating pair_score without its early returns, for ILP: no difference.)
// kSorted: the tile is 64 consecutive DEPTH-SORTED positions of the candidate array (k_depth_order); lane t
// owns candidate perm[t] and sweeps only the sorted positions rng[t] of its node.
// kPerm: the candidate at position t of the (virtual) compact array is record perm[t] of a.cand / a.lite (the
// staging lists of stage B: k_place wrote only the permutation); scores are indexed by position.
// (split form with four independent waves to a workgroup, 1 024 workgroups instead of 4 096: resident 6 us earlier, the first
// tiles' prologues slower by as much -- 41.3 against 40.2 us; one wave per workgroup stays)
constexpr int kSweepWaves = 1;
template <bool kF32, bool kSorted, bool kPerm, bool kSplit>
__global__ void __launch_bounds__(kSplit ? 64 * kSweepWaves : 64) LT_SCORE_OCC
k_score3(Score3Args a, ScoreCfg cfg, double scaleinv_guard2) {
  static_assert(!kSplit || kF32, "the split form exists for the single-precision sweep");
  constexpr bool kInd = kSorted || kPerm;  // positions are mapped through a.perm
  constexpr int kQCap = kSplit ? kSQCapSplit : kSQCap;
  extern __shared__ __align__(16) unsigned char smem_all[];
  const int lane = (int)(threadIdx.x & 63u);
  // kSplit: the waves of a workgroup are independent of each other (own LDS part, own tiles)
  const unsigned wave_in_wg = kSplit ? (unsigned)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : 0u;
  const unsigned wave_id = kSplit ? blockIdx.x * (unsigned)kSweepWaves + wave_in_wg : blockIdx.x;
  const unsigned n_waves = kSplit ? gridDim.x * (unsigned)kSweepWaves : gridDim.x;
  unsigned char *smem_raw = smem_all + (kSplit ? (size_t)wave_in_wg * (size_t)a.sp_wave_lds : (size_t)0);
  // LDS: window (f32: float4[kWin][3]; f64: W[9][kWin] f64 + wslot[kWin] i32) | woff[64] i64 | queue[kSQCap] u32 |
  //      ord[max_nb] i32 | S[max_nb][64] u64
  constexpr size_t kWBytes = kF32 ? (size_t)kWin * 48 : (size_t)9 * kWin * 8 + (size_t)kWin * 4;
  double *W = reinterpret_cast<double *>(smem_raw);
  int *wslot = reinterpret_cast<int *>(smem_raw + (size_t)9 * kWin * 8);
  float4 *W4 = reinterpret_cast<float4 *>(smem_raw);
  long long *woff = reinterpret_cast<long long *>(smem_raw + kWBytes);
  unsigned *queue = reinterpret_cast<unsigned *>(smem_raw + kWBytes + 64 * 8);
  uint2 *queue2 = reinterpret_cast<uint2 *>(smem_raw + kWBytes + 64 * 8);  // kSplit: (record of j, lane of i)
  unsigned *reci_l = reinterpret_cast<unsigned *>(smem_raw + kWBytes + 64 * 8 + (size_t)kSQCapSplit * 8);  // kSplit: records of the lanes
  int *ordl = reinterpret_cast<int *>(smem_raw + kWBytes + 64 * 8 + kSQCap * 4);
  unsigned long long *S = reinterpret_cast<unsigned long long *>(
      smem_raw + ((kWBytes + 64 * 8 + kSQCap * 4 + (size_t)a.max_nb * 4 + 15) & ~(size_t)15));

  const long long C = a.tri_off[a.G];
  const unsigned n_tiles = (unsigned)((C + 63) >> 6);
  // Persistent wave: tiles (64 consecutive candidates) are drawn through kTileQueues counters -- queue q
  // holds the tiles q, q + 8, ... and is served by the workgroups of one XCD (round-robin dispatch), an empty
  // queue sends its waves to the next one.  The draw for the next tile is issued at the current tile's last dense
  // round and read after its sums.  (Listing the tiles by the size of their largest node, longest first, was
  // measured: no gain -- a tile's time is set by how many of its pairs survive the sweep, which neither the
  // largest node nor the number of pairs of the tile predicts: correlation 0.6.)
  int q = (int)(blockIdx.x & (kTileQueues - 1)), tried = 0;
  unsigned long long n_eval_total = 0;
  unsigned k_raw = 0;
  // (the static schedule of the split form applied to the fused kernel over the exhaustive mode's natural tile order --
  // wave w takes tiles w, w + waves, ... --: 5.65 against 3.20 ms for the stage, the tiles' costs are not spread that evenly)
  if (!kSplit && lane == 0) k_raw = atomicAdd(&a.draw[q * 32], 1u);
  // draws are mapped to tiles through the queue's cost-class lists, most expensive class first: lane b < kTileBuckets
  // holds the size of class (kTileBuckets - 1 - b) of queue q and the inclusive prefix of the sizes in that order
  unsigned cls_cnt = 0, cls_incl = 0, q_tiles = 0;
  auto load_classes = [&]() {
    cls_cnt = lane < kTileBuckets ? a.bucket_cnt[(q * kTileBuckets + (kTileBuckets - 1 - lane)) * 32] : 0u;
    cls_incl = cls_cnt;
#pragma unroll
    for (int d = 1; d < kTileBuckets; d <<= 1) {
      const unsigned t = (unsigned)__shfl_up((int)cls_incl, d);
      if (lane >= d) cls_incl += t;
    }
    q_tiles = (unsigned)__builtin_amdgcn_readlane((int)cls_incl, kTileBuckets - 1);
  };
  if (!kSplit && a.bucket_cnt) load_classes();
  // a draw resolves to the tile and -- from the class lists -- the bounds of its window (x: tile, 0xFFFFFFFF when
  // every queue is empty; y, z: first and end position of the window, z == 0: not known, derived from the lanes' nodes)
  auto resolve = [&]() -> uint4 {
    for (;;) {
      const unsigned k = (unsigned)__builtin_amdgcn_readfirstlane((int)k_raw);
      if (a.bucket_cnt) {
        if (k < q_tiles) {
          const unsigned long long m = __ballot(lane < kTileBuckets && cls_incl > k);
          const int bl = __builtin_ctzll(m);  // k < q_tiles: some class holds it
          const unsigned base = (unsigned)__builtin_amdgcn_readlane((int)(cls_incl - cls_cnt), bl);
          return reinterpret_cast<const uint4 *>(
              a.bucket_list)[(size_t)(q * kTileBuckets + (kTileBuckets - 1 - bl)) * a.bucket_cap + (k - base)];
        }
      } else {
        const unsigned long long e = (unsigned long long)k * kTileQueues + (unsigned)q;
        if (e < n_tiles) return uint4{(unsigned)e, 0u, 0u, 0u};
      }
      // This queue is exhausted.  PEEK at all eight counters (plain loads; a counter only grows, so a queue that looks
      // exhausted is) and draw only from one that looks open: without this every wave ended with eight failing device
      // atomics -- dependent round trips behind the kernel's last tiles.
      if (++tried > 4 * kTileQueues) return uint4{0xFFFFFFFFu, 0u, 0u, 0u};
      unsigned peek = 0xFFFFFFFFu, cap_l = 0;
      if (lane < kTileQueues) {
        peek = __hip_atomic_load(&a.draw[lane * 32], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        cap_l = (n_tiles + (unsigned)(kTileQueues - 1 - lane)) / (unsigned)kTileQueues;  // tiles lane, lane + 8, ...
      }
      const unsigned long long open_q = __ballot(lane < kTileQueues && lane != q && peek < cap_l);
      if (!open_q) return uint4{0xFFFFFFFFu, 0u, 0u, 0u};
      const unsigned long long after = open_q & ~((2ull << q) - 1ull);
      q = __builtin_ctzll(after ? after : open_q);
      if (a.bucket_cnt) load_classes();
      if (lane == 0) k_raw = atomicAdd(&a.draw[q * 32], 1u);
    }
  };
  // First level of a tile's loads, issued as soon as its draw is resolved: node record, record index and -- the class
  // lists carry the window bounds -- the record indices of the first window chunk, so that the chain to the first
  // window entry in LDS is draw -> {node record, indices} -> records instead of draw -> node record -> (bounds by
  // shuffles) -> indices -> records.  (Issuing this level BEFORE the previous tile's final dense rounds, so that it lands
  // while they run, was measured twice -- rounds 2 and 3: 133 against 122 us; the dense rounds' own loads queue behind
  // it on the in-order vmcnt.)
  CandMeta p_mt = {0u, 0u, 0u, 0u};
  unsigned p_rg = 0u;
  unsigned p_i = 0, p_w0 = 0, p_w1 = 0;
  static_assert(kWin <= 128, "the first window chunk's record indices are two per lane");
  auto load_first_level = [&](const uint4 h) {
    const long long tp = (long long)h.x * 64 + lane;
    if (tp < C) {
      p_mt = a.meta[tp];  // a position and its candidate belong to the same node
      p_i = kInd ? a.perm[tp] : (unsigned)tp;
      if (kSorted) p_rg = a.rng[tp];
    }
    if (kInd && !kSorted && h.z > h.y) {
      const unsigned w = (h.z - h.y) < (unsigned)kWin ? (h.z - h.y) : (unsigned)kWin;
      if ((unsigned)lane < w) p_w0 = a.perm[(size_t)h.y + lane];
      if ((unsigned)lane + 64u < w) p_w1 = a.perm[(size_t)h.y + 64 + lane];
    }
  };
  bool ch_dead = false;  // kSplit: the overflow store is full, the run is repeated with the fused kernel
  auto ch_alloc = [&]() -> unsigned {
    unsigned id = 0;
    if (lane == 0) id = atomicAdd(&a.sp_counters[0], 1u);
    id = (unsigned)__builtin_amdgcn_readfirstlane((int)id);
    if (id >= a.sp_chunk_cap) {
      ch_dead = true;
      if (lane == 0 && a.err_flag) atomicCAS(a.err_flag, 0, kErrPairChunks);
      return kNoChunk;
    }
    return id;
  };
  // kSplit: STATIC schedule.  Without the dense rounds a tile's time follows its cost class closely, so wave w takes
  // entries w, w + waves, w + 2 waves, ... of the tiles in class order (all queues' lists of the most expensive class first):
  // one expensive, one medium and one cheap tile each, no draw and no round trip between tiles -- the next tile's list
  // entry is loaded at the start of the current one and its first load level while the current one sweeps (measured with
  // draws: 1 300 of 3 072 waves inside a tile at any time, the rest between tiles).
  // flat order f = class rank * kTileQueues + queue (rank 0 = most expensive); lane l holds f = 4 l .. 4 l + 3
  unsigned o_cnt[4] = {0u, 0u, 0u, 0u}, o_tot = 0, o_incl = 0, n_order = n_tiles;
  if (kSplit && a.bucket_cnt) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int f = 4 * lane + u;
      o_cnt[u] = a.bucket_cnt[((f & (kTileQueues - 1)) * kTileBuckets + (kTileBuckets - 1 - (f >> 3))) * 32];
      o_tot += o_cnt[u];
    }
    o_incl = o_tot;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const unsigned t = (unsigned)__shfl_up((int)o_incl, d);
      if (lane >= d) o_incl += t;
    }
    n_order = (unsigned)__builtin_amdgcn_readlane((int)o_incl, 63);
  }
  static_assert(kTileQueues * kTileBuckets == 256, "four (queue, class) lists per lane");
  auto fetch_order = [&](unsigned k, size_t &ent) -> uint4 {  // k < n_order, wave-uniform; ent: the entry's index in the lists
    ent = 0;
    if (!a.bucket_cnt) return uint4{k, 0u, 0u, 0u};
    const int l = __builtin_ctzll(__ballot(o_incl > k));
    unsigned r = k - (unsigned)__builtin_amdgcn_readlane((int)(o_incl - o_tot), l);
    int u = 0;
#pragma unroll
    for (int v = 0; v < 3; ++v) {
      const unsigned c = (unsigned)__builtin_amdgcn_readlane((int)o_cnt[v], l);
      if (u == v && r >= c) { r -= c; u = v + 1; }
    }
    const int f = 4 * l + u;
    ent = (size_t)((f & (kTileQueues - 1)) * kTileBuckets + (kTileBuckets - 1 - (f >> 3))) * a.bucket_cap + r;
    return reinterpret_cast<const uint4 *>(a.bucket_list)[ent];
  };
  const uint4 kNoTile = uint4{0xFFFFFFFFu, 0u, 0u, 0u};
  // pass m of the schedule takes entries [m waves, (m + 1) waves) of the order, forwards for even m and backwards for odd
  // m: the wave with the most expensive tile of one pass gets the cheapest of the next
  unsigned o_pass = 0;
  auto order_index = [&](unsigned pass) -> unsigned {
    const unsigned w = (pass & 1u) ? n_waves - 1u - wave_id : wave_id;
    return pass * n_waves + w;
  };
  unsigned o_k = order_index(0);
  uint4 hdr_next = kNoTile;
  size_t ent_cur = 0, ent_next = 0;
  uint4 hdr = kSplit ? (o_k < n_order ? fetch_order(o_k, ent_cur) : kNoTile) : resolve();
  if (hdr.x != 0xFFFFFFFFu) load_first_level(hdr);
  // kSplit: a finished tile is appended to the list of its pair-count class (a.pc_*).  The append's atomic is issued at the
  // tile's end and its result used -- the entry stored -- only after the NEXT tile's first window is staged, so that the wave
  // does not sit on the round trip
  bool pc_pending = false;
  unsigned pc_idx = 0, pc_qb = 0;
  uint2 pc_ent = make_uint2(0u, 0u);
  auto pc_flush = [&]() {
    if (!pc_pending) return;
    if (lane == 0 && pc_idx < a.pc_cap) a.pc_list[(size_t)pc_qb * a.pc_cap + pc_idx] = pc_ent;
    pc_pending = false;
  };
  while (hdr.x != 0xFFFFFFFFu) {
    bool next_level_issued = false;
    if (kSplit) {
      // (a backwards pass that runs beyond the end of the order has no tile for this wave, but the pass after it may not
      // either: the order ends inside this pass)
      o_k = order_index(++o_pass);
      hdr_next = o_k < n_order ? fetch_order(o_k, ent_next) : kNoTile;
    }
    const unsigned tile = (unsigned)__builtin_amdgcn_readfirstlane((int)hdr.x);
    const unsigned h_lo = (unsigned)__builtin_amdgcn_readfirstlane((int)hdr.y);
    const unsigned h_hi = (unsigned)__builtin_amdgcn_readfirstlane((int)hdr.z);
    const bool h_bounds = !kSorted && h_hi > h_lo;
    const long long i0 = (long long)tile * 64;
    const long long tpos = i0 + lane;  // position (sorted order if kSorted)
    const bool active = tpos < C;
    const long long i = active ? (long long)p_i : tpos;  // the lane's candidate record
    const unsigned w0src = p_w0, w1src = p_w1;           // records of the first window chunk's entries lane, lane + 64
    LT_TRACE_MARK(2, tile, 0);

    long long off = 0, nb0 = 0;
    int n = 0, n_nb = 0, sloti = -1;
    int r_lo = 0, r_hi = 0;  // node-relative positions the lane sweeps
    double dix = 0, diy = 0, diz = 0;
    double six = 0, siy = 0, siz = 0, eix = 0, eiy = 0, eiz = 0, gs2 = 0, ge2 = 0;
    if (active) {
      const CandMeta mt = p_mt;
      off = ((long long)mt.off_hi << 32) | (long long)mt.off_lo;
      n = (int)mt.n;
      r_lo = 0; r_hi = n;
      if (kSorted) {
        const unsigned rg = p_rg;
        if (rg != 0xFFFFFFFFu) { r_lo = (int)(rg & 0xFFFFu); r_hi = (int)(rg >> 16); }
      }
      nb0 = (long long)(mt.nb >> 8);
      n_nb = (int)(mt.nb & 0xFFu);
      const CRec &ci = a.cand[i];
      dix = ci.dir[0]; diy = ci.dir[1]; diz = ci.dir[2];
      sloti = crec_slot(ci);
      six = ci.s[0]; siy = ci.s[1]; siz = ci.s[2];
      eix = ci.e[0]; eiy = ci.e[1]; eiz = ci.e[2];
      // dist / (depth + eps) > th_scaleinv (1 + 1e-6) can never score >= score_th (line_dists.cc:55-60)
      double zs = ci.depth[0] + kEps, ze = ci.depth[1] + kEps;
      gs2 = (zs > 0.0) ? scaleinv_guard2 * zs * zs : 1e300;  // odd depths: leave it to the exact path
      ge2 = (ze > 0.0) ? scaleinv_guard2 * ze * ze : 1e300;
    }
    woff[lane] = off;
    if (kSplit) reci_l[lane] = (unsigned)i;
    // summation order of the first lane's image, staged once (lanes of another image read it from HBM)
    const long long wave_nb0 = (long long)__builtin_amdgcn_readfirstlane((int)nb0);  // nb_off < 2^24
    if (!kSplit) {
      if (lane < __builtin_amdgcn_readfirstlane(n_nb)) ordl[lane] = a.blk_order[wave_nb0 + lane];
      for (int k = 0; k < a.max_nb; ++k) S[k * 64 + lane] = 0ull;
    }
    // kF32: wave-local origin (the first lane's start point) and this lane's own single-precision operands
    double ox = 0, oy = 0, oz = 0;
    float dixf = 0, diyf = 0, dizf = 0, sixf = 0, siyf = 0, sizf = 0, eixf = 0, eiyf = 0, eizf = 0, ri = 0;
    double gs = 0, ge = 0;
    float cosf_guard = -2.0f;
    if (kF32) {
      ox = readlane_f64(six, 0); oy = readlane_f64(siy, 0); oz = readlane_f64(siz, 0);
      dixf = (float)dix; diyf = (float)diy; dizf = (float)diz;
      sixf = (float)(six - ox); siyf = (float)(siy - oy); sizf = (float)(siz - oz);
      eixf = (float)(eix - ox); eiyf = (float)(eiy - oy); eizf = (float)(eiz - oz);
      ri = fmaxf(fmaxf(fmaxf(fabsf(sixf), fabsf(siyf)), fabsf(sizf)), fmaxf(fmaxf(fabsf(eixf), fabsf(eiyf)), fabsf(eizf)));
      if (!active) ri = 0.0f;
      gs = sqrt(gs2);
      ge = sqrt(ge2);
      cosf_guard = cfg.cos_guard > -1.0 ? (float)(cfg.cos_guard - 2e-6) : -2.0f;
    }
    // Range of positions this wave has to stage (lane 0 is always active; positions fit 32 bits).  Natural order: the
    // lanes' nodes ascend with the lane, so the range runs from the first lane's node to the end of the last active
    // lane's -- two scalar reads.  Depth-sorted ranges take a DPP reduction (no ds_bpermute butterflies: every level
    // of those is a round trip through the LDS pipe, and a tile had eighteen of them in a row).
    long long lo, hi;
    if (h_bounds) {
      lo = (long long)h_lo; hi = (long long)h_hi;
    } else if (!kSorted) {
      const int last = (int)((C - i0) < 64 ? (C - i0) : 64) - 1;
      lo = (long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(off + r_lo));
      hi = (long long)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)(off + r_hi), last);
    } else {
      lo = (long long)wave_min_u32(active ? (unsigned)(off + r_lo) : 0xFFFFFFFFu);
      hi = (long long)wave_max_u32(active ? (unsigned)(off + r_hi) : 0u);
    }
    int qn = 0;
    unsigned long long n_eval = 0;

    int t_cnt = 0, ov_fill = 0;  // kSplit: pairs of this tile so far, fill of its current overflow chunk
    unsigned ov_cur = kNoChunk;
    auto drain = [&](bool final) {
      wave_lds_sync();
      if (kSplit) {
        if (final && a.pc_cnt) {  // the list append's atomic first: its round trip runs under the stores below
          pc_flush();  // (still pending only if this tile had no window to stage)
          pc_qb = (tile & (unsigned)(kTileQueues - 1)) * (unsigned)kTileBuckets + (unsigned)pair_bucket((unsigned)(t_cnt + qn));
          pc_ent = make_uint2(tile, (unsigned)(t_cnt + qn));
          if (lane == 0) pc_idx = atomicAdd(&a.pc_cnt[pc_qb * 32], 1u);
          pc_pending = true;
        }
        const int k0 = min(qn, max(0, a.sp_slot_cap - t_cnt));
        uint4 *dst = a.sp_slots + (size_t)tile * (size_t)a.sp_slot_cap + t_cnt;
        for (int p = lane; p < k0; p += 64) {
          const uint2 e = queue2[p];
          dst[p] = uint4{e.x, reci_l[e.y], e.y, 0u};
        }
        int done = k0;
        while (done < qn && !ch_dead) {  // beyond the slot: the tile's chain of overflow chunks
          if (ov_cur == kNoChunk || ov_fill == kChunkCap) {
            const unsigned nxt = ch_alloc();
            if (ch_dead) break;
            if (lane == 0) {
              if (ov_cur == kNoChunk) a.sp_ovf[tile] = nxt;
              else a.sp_desc[ov_cur] = make_uint2((unsigned)ov_fill, nxt);
            }
            ov_cur = nxt;
            ov_fill = 0;
          }
          const int k = min(qn - done, kChunkCap - ov_fill);
          uint4 *od = a.sp_pairs + (size_t)ov_cur * kChunkCap + ov_fill;
          for (int p = lane; p < k; p += 64) {
            const uint2 e = queue2[done + p];
            od[p] = uint4{e.x, reci_l[e.y], e.y, 0u};
          }
          ov_fill += k;
          done += k;
        }
        t_cnt += qn;
        n_eval += (unsigned long long)qn;
        qn = 0;
        if (final && lane == 0) {
          a.sp_cnt[tile] = (unsigned)t_cnt;
          if (a.bucket_cnt)  // word 3 of the tile's class-list entry: k_dense8 gets tile and count with one load
            const_cast<unsigned *>(a.bucket_list)[4 * ent_cur + 3] = (unsigned)t_cnt;
          if (ov_cur != kNoChunk) a.sp_desc[ov_cur] = make_uint2((unsigned)ov_fill, kNoChunk);
        }
        wave_lds_sync();
        return;
      }
      if (final && qn == 0 && lane == 0) k_raw = atomicAdd(&a.draw[q * 32], 1u);
      for (int q0 = 0; q0 < qn; q0 += 64) {
        if (final && q0 + 64 >= qn && lane == 0) k_raw = atomicAdd(&a.draw[q * 32], 1u);
        const int p = q0 + lane;
        if (p < qn) {
          const unsigned e = queue[p];
          const int il = (int)(e >> 26);
          const long long jpos = woff[il] + (long long)(e & 0x3FFFFFFu);
          const long long j = kInd ? (long long)a.perm[jpos] : jpos;
          const long long ii = kInd ? (long long)a.perm[i0 + il] : i0 + il;
          const CRec &ci = a.cand[ii];
          const CRec &cj = a.cand[j];
          const int nbs_j = cj.nb_slot;
          const double sc = pair_score(cfg, mk3(ci.s[0], ci.s[1], ci.s[2]), mk3(ci.e[0], ci.e[1], ci.e[2]),
                                       mk3(ci.dir[0], ci.dir[1], ci.dir[2]), ci.depth[0], ci.depth[1],
                                       mk3(cj.s[0], cj.s[1], cj.s[2]), mk3(cj.e[0], cj.e[1], cj.e[2]),
                                       mk3(cj.dir[0], cj.dir[1], cj.dir[2]), cj.seg,
                                       a.cams[(int)((unsigned)nbs_j >> 8)]);
          if (sc > 0.0) atomicMax(&S[(nbs_j & 0xFF) * 64 + il], (unsigned long long)__double_as_longlong(sc));
        }
      }
      n_eval += (unsigned long long)qn;
      qn = 0;
      wave_lds_sync();
    };

    for (long long wb = lo; wb < hi; wb += kWin) {
      wave_lds_sync();
      const int wn = (int)((hi - wb) < kWin ? (hi - wb) : kWin);
      float rw = ri;
      for (int e = lane; e < wn; e += 64) {
        const long long src = !kInd ? wb + e
                              : ((h_bounds && wb == lo && e < 128) ? (long long)(e < 64 ? w0src : w1src) : (long long)a.perm[wb + e]);
        const CRec &c = a.cand[src];
        const CRec &l = c;
        if (kF32) {
          const float sx = (float)(c.s[0] - ox), sy = (float)(c.s[1] - oy), sz = (float)(c.s[2] - oz);
          const float ex = (float)(c.e[0] - ox), ey = (float)(c.e[1] - oy), ez = (float)(c.e[2] - oz);
          W4[3 * e + 0] = float4{(float)l.dir[0], (float)l.dir[1], (float)l.dir[2], __int_as_float(crec_slot(l))};
          W4[3 * e + 1] = float4{sx, ex, sy, ey};  // start / end interleaved: the sweep's packed-f32 operand pairs
          W4[3 * e + 2] = float4{sz, ez, __uint_as_float((unsigned)src), 0.0f};  // .z: the record (kSplit's pair entries)
          rw = fmaxf(rw, fmaxf(fmaxf(fmaxf(fabsf(sx), fabsf(sy)), fabsf(sz)), fmaxf(fmaxf(fabsf(ex), fabsf(ey)), fabsf(ez))));
        } else {
          W[0 * kWin + e] = l.dir[0]; W[1 * kWin + e] = l.dir[1]; W[2 * kWin + e] = l.dir[2];
          W[3 * kWin + e] = c.s[0]; W[4 * kWin + e] = c.s[1]; W[5 * kWin + e] = c.s[2];
          W[6 * kWin + e] = c.e[0]; W[7 * kWin + e] = c.e[1]; W[8 * kWin + e] = c.e[2];
          wslot[e] = crec_slot(l);
        }
      }
      float gsf = 0.0f, gef = 0.0f;
      if (kF32) {
        // R of the window -> this lane's single-precision distance guards (a NaN coordinate makes R NaN,
        // the guards NaN and every comparison false: everything goes to the exact evaluation)
        rw = wave_max_f32_nan(rw);
        const double delta = 1e-6 * (double)rw;
        gsf = (float)((gs + delta) * (gs + delta) * (1.0 + 2e-6));
        gef = (float)((ge + delta) * (ge + delta) * (1.0 + 2e-6));
      }
      wave_lds_sync();
      if (wb == lo) { LT_TRACE_MARK(2, tile, 1); }
      if (kSplit && !next_level_issued) {  // the registers of this tile's first level are free now
        next_level_issued = true;
        if (hdr_next.x != 0xFFFFFFFFu) load_first_level(hdr_next);
        pc_flush();  // the previous tile's list entry
      }
      // this lane's sub-range of the window
      long long jlo = (off + r_lo) > wb ? (off + r_lo) : wb;
      long long jhi = (off + r_hi) < (wb + wn) ? (off + r_hi) : (wb + wn);
      int cnt = (active && jhi > jlo) ? (int)(jhi - jlo) : 0;
      const int cmax = wave_max_i32(cnt);
      const int w0 = (int)(jlo - wb);
      const int jj0 = (int)(jlo - off);
      const int self_t = (int)(tpos - jlo);  // iteration at which the lane meets itself (may be out of range)
      if (kF32) {
        const int wlast = cnt > 0 ? w0 + cnt - 1 : 0;  // reads beyond the lane's range are clamped, then masked
        const int wbase = cnt > 0 ? w0 : 0;
        // (one instance of drain() in the code instead of two -- 4040 instead of 5951 lines of ISA -- was measured:
        // 128.9 against 124.7 us)
        for (int t = 0; t < cmax; t += 4) {
          float4 A[4], B[4];
          float2 E[4];
          unsigned Jr[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int w = min(wbase + t + u, wlast);
            A[u] = W4[3 * w + 0];
            B[u] = W4[3 * w + 1];
            if (kSplit) {
              const float4 e4 = W4[3 * w + 2];
              E[u] = make_float2(e4.x, e4.y);
              Jr[u] = __float_as_uint(e4.z);
            } else {
              E[u] = *reinterpret_cast<const float2 *>(&W4[3 * w + 2]);
            }
          }
          bool pass[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const float c = fabsf(__builtin_fmaf(dizf, A[u].z, __builtin_fmaf(diyf, A[u].y, dixf * A[u].x)));
            // (start, end) pairs: v_pk_add / v_pk_mul / v_pk_fma_f32 straight from the window's layout
            const float ax = sixf - B[u].x, bx = eixf - B[u].y;
            const float ay = siyf - B[u].z, by = eiyf - B[u].w;
            const float az = sizf - E[u].x, bz = eizf - E[u].y;
            const float ds2 = __builtin_fmaf(az, az, __builtin_fmaf(ay, ay, ax * ax));
            const float de2 = __builtin_fmaf(bz, bz, __builtin_fmaf(by, by, bx * bx));
            // below the cosine guard the 3D angle score is certainly gated to 0; beyond the squared
            // distance guards the scale-invariant endpoint score is.  (Bitwise &: no branch per test -- the
            // reads are clamped, everything may be evaluated.)
            pass[u] = (t + u < cnt) & (t + u != self_t) & (__float_as_int(A[u].w) != sloti) & !(c < cosf_guard) &
                      !(ds2 > gsf) & !(de2 > gef);
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const unsigned long long m = __ballot(pass[u]);
            if (m) {
              if (kSplit) {
                if (pass[u]) queue2[qn + __popcll(m & lanemask_lt())] = make_uint2(Jr[u], (unsigned)lane);
              } else {
                if (pass[u]) queue[qn + __popcll(m & lanemask_lt())] = ((unsigned)lane << 26) | (unsigned)(jj0 + t + u);
              }
              qn += __popcll(m);
            }
          }
          if (qn > kQCap - 256) drain(false);
        }
      } else {
        for (int t = 0; t < cmax; ++t) {
          bool pass = t < cnt;
          if (pass) {
            const int w = w0 + t;
            const int sl = wslot[w];
            const double jx = W[0 * kWin + w], jy = W[1 * kWin + w], jz = W[2 * kWin + w];
            const double sx = W[3 * kWin + w], sy = W[4 * kWin + w], sz = W[5 * kWin + w];
            const double ex = W[6 * kWin + w], ey = W[7 * kWin + w], ez = W[8 * kWin + w];
            const double c = fabs((dix * jx + diy * jy) + diz * jz);
            const double ax = six - sx, ay = siy - sy, az = siz - sz;
            const double bx = eix - ex, by = eiy - ey, bz = eiz - ez;
            const double ds2 = ax * ax + ay * ay + az * az, de2 = bx * bx + by * by + bz * bz;
            pass = (t != self_t) && (sl != sloti) && !(c < cfg.cos_guard) && !(ds2 > gs2) && !(de2 > ge2);
          }
          const unsigned long long m = __ballot(pass);
          if (m) {
            if (pass) queue[qn + __popcll(m & lanemask_lt())] = ((unsigned)lane << 26) | (unsigned)(jj0 + t);
            qn += __popcll(m);
            if (qn > kSQCap - 256) drain(false);
          }
        }
      }
    }
    LT_TRACE_MARK(2, tile, 2);
    drain(true);
    LT_TRACE_MARK(2, tile, 3);

    if (!kSplit && active) {
      double sum = 0.0;
      const bool own = nb0 == wave_nb0;
      for (int r = 0; r < n_nb; ++r) {
        int k = own ? ordl[r] : a.blk_order[nb0 + r];
        sum += __longlong_as_double((long long)S[k * 64 + lane]);
      }
      a.score[kPerm ? tpos : (kSorted && a.spos ? (long long)a.spos[tpos] : i)] = sum;
    }
    n_eval_total += n_eval;
    wave_lds_sync();  // the tables are reused by the next tile
    if (kSplit) {
      if (!next_level_issued && hdr_next.x != 0xFFFFFFFFu) load_first_level(hdr_next);
      hdr = hdr_next;
      ent_cur = ent_next;
    } else {
      hdr = resolve();
      if (hdr.x != 0xFFFFFFFFu) load_first_level(hdr);
    }
  }
  if (kSplit) pc_flush();
  // (kSplit: the pair statistic is summed by k_dense8 from the tiles' counts -- here every wave ends at about the same time,
  // and three thousand atomics on one address in a burst held up the loads of the waves that still had a tile: 20-40 us)
  if (!kSplit && lane == 0 && a.pair_counter && n_eval_total) atomicAdd(a.pair_counter, n_eval_total);
}

// Dense half of the split form: one workgroup of kWaves waves per unit of sp_t_max tiles.  Tables of per-image maxima as in
// the fused kernel, one per tile of the unit; the sums in image-id order.  Units in the sweep's class order, most expensive
// class first (a unit = T consecutive entries of one (queue, class) list: tile and pair count in one 16-byte load); the first
// unit of a workgroup is static, the later ones claimed through counters.  A unit starts with one load level (pair entries +
// per-candidate records, the header was fetched during the previous unit's sums) followed by the record gathers (both
// records of a pair from the entry).  Two barriers per unit: tables clean -> rounds -> every pair in -> sums (wave w sums and
// zeroes the table of tile w while the next unit's header and first load level are already under way).
// (Fewer tiles per unit in the expensive classes -- one from class 22 / two from class 12, from 26 / 18, from 30 / 22: no
// gain over T everywhere.  Workgroups of two waves / one wave with their own tables: 133 / 195 us for the stage against 106.)
constexpr int kDenseWaves = 4;
// (round 5, k_dense8 at 113 registers: four workgroups per CU with units of three tiles -- 36 KB of tables at 20
// neighbours -- 107.5 us for the stage against 108.9 with three workgroups and units of four tiles (48 KB); three
// workgroups with units of three: 110.6)
#ifndef LT_DENSE_PER_CU
#define LT_DENSE_PER_CU 4
#endif
#ifndef LT_DENSE_LDS_KB
#define LT_DENSE_LDS_KB 36
#endif
#ifndef LT_DENSE_SUM_CHUNK
#define LT_DENSE_SUM_CHUNK 4
#endif
constexpr int kSumChunk = LT_DENSE_SUM_CHUNK;
constexpr int kDensePerCU = LT_DENSE_PER_CU;                  // workgroups of k_dense8 per CU
constexpr int kDenseLdsBudget = LT_DENSE_LDS_KB * 1024;      // LDS for a workgroup's tables of maxima
// (round 6, measured and not kept: a STATIC schedule -- workgroup b takes units b, 2 G - 1 - b, 2 G + b, ... of the pair-count
// order, no claims, so that the header of unit j + 2 and every thread's first entry of unit j + 1 can be fetched a unit ahead,
// behind the rounds' own gathers -- 98.3 us for the stage against 95.5 with the claims below: what the prefetch saves per unit
// the fixed assignment loses in balance, 286 of 1 024 workgroups still busy at 40 us of 49.)
// (... and claims ONE UNIT AHEAD -- second unit static as well, the claim for unit j + 2 issued at the start of unit j, the
// header of unit j + 2 fetched during unit j's sums and every thread's first entry of unit j + 1 at the end of unit j's
// rounds --: 100.0 against 96.2 us, same bits.  The kernel without its arithmetic and without its record gathers
// (LT_ABL_DENSE=2) still takes 32 of its 48 us, but not as a chain of exposed round trips at the start of a unit: taking
// those away does not shorten it.)
#ifdef LT_DENSE_WPE
#define LT_DENSE_OCC __attribute__((amdgpu_waves_per_eu(LT_DENSE_WPE, LT_DENSE_WPE)))
#else
#define LT_DENSE_OCC
#endif
template <bool kFast>  // kFast: pair_score_fused / pair_score_terms (ScoreCfg::fast)
__global__ void __launch_bounds__(64 * kDenseWaves) LT_DENSE_OCC
k_dense8(Score3Args a, ScoreCfg cfg, int score_by) {  // score_by: 0 = record of the candidate, 1 = position, 2 = spos[position]
  extern __shared__ __align__(16) unsigned char smem_raw[];
  unsigned *s_next = reinterpret_cast<unsigned *>(smem_raw);  // [4]
  unsigned *s_cnt = s_next + 4;                               // [kChunkTiles] pairs of the unit's tiles
  unsigned *s_tile = s_cnt + kChunkTiles;                     // [kChunkTiles] the unit's tiles
  unsigned *s_ocnt = s_tile + kChunkTiles;                    // [256] tiles of the (queue, class) lists in flat order
  unsigned *s_uincl = s_ocnt + 256;                           // [64] inclusive prefix of the lists' units, four lists per entry
  unsigned long long *S = reinterpret_cast<unsigned long long *>(smem_raw + kDenseHdrBytes);
  if (a.err_flag && *a.err_flag == kErrPairChunks) return;  // the overflow store was full: the run is repeated
  const long long C = a.tri_off[a.G];
  const unsigned n_tiles = (unsigned)((C + 63) >> 6);
  const int T = a.sp_t_max;
  constexpr int kWaves = kDenseWaves;
  constexpr int kThreads = 64 * kWaves;
  static_assert(kChunkTiles * 64 <= 2 * kThreads, "two table rows per thread");
  const int tid = threadIdx.x;
  const int lane = lane_id();
  // flat order f = class rank * kTileQueues + queue as in the sweep; entry l of s_uincl covers the lists f = 4 l .. 4 l + 3
  // (the lists' counts live in LDS: ten more registers held across the rounds cost the third wave per SIMD)
  unsigned n_units = (n_tiles + (unsigned)T - 1) / (unsigned)T;
  // (fewer tiles per unit in the most expensive classes -- one tile from class 30 / 28 / 24 / 20, two from 24 / 20 / 16 / 12:
  // 65.8 / 67.1 / 68.4 / 69.3 us against 66.0 with T everywhere; four workgroups per CU with three tiles: 65.2)
  // round 6: the lists are the sweep's PAIR-COUNT classes when it wrote them (a.pc_cnt) -- a unit's time is its pairs, and the
  // cost classes of k_cand_meta (sum of node sizes) predict them with a correlation of 0.6: heavy units started late, 17 us
  // of the kernel's 61 were its tail (trace in profiles/r06_score_experiments.txt)
  const bool use_pc = a.pc_cnt != nullptr;
  // (heavy tiles as units of their own -- one tile per unit from 192 pairs, two from 96: 115 us for the stage against 107 with T
  // everywhere, 3 878 units instead of 3 091 and half-empty second rounds; from 256 / 128: 110)
  auto tiles_per_unit = [&](int) -> int { return T; };
  const unsigned *list_cnt = use_pc ? a.pc_cnt : a.bucket_cnt;
  const unsigned list_cap = use_pc ? a.pc_cap : a.bucket_cap;
  if (list_cnt) {
    if (tid < 64) {
      unsigned u_tot = 0;
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int f = 4 * lane + v;
        const unsigned c = list_cnt[((f & (kTileQueues - 1)) * kTileBuckets + (kTileBuckets - 1 - (f >> 3))) * 32];
        s_ocnt[f] = c;
        const unsigned tf = (unsigned)tiles_per_unit(f);
        u_tot += (c + tf - 1) / tf;
      }
      unsigned u_incl = u_tot;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const unsigned t = (unsigned)__shfl_up((int)u_incl, d);
        if (lane >= d) u_incl += t;
      }
      s_uincl[lane] = u_incl;
    }
    __syncthreads();
    n_units = s_uincl[63];
  }
  const int max_nb = a.max_nb;
  const int cap = a.sp_slot_cap;
  // unit -> its tiles: number and the index of its first class-list entry (natural order: its first tile)
  auto locate = [&](unsigned uu, int &nt_o, size_t &base_o) {
    nt_o = 0;
    base_o = 0;
    if (uu >= n_units) return;
    if (list_cnt) {
      const int l = __builtin_ctzll(__ballot(s_uincl[lane] > uu));
      unsigned r = uu - (l > 0 ? s_uincl[l - 1] : 0u);
      int v = 0;
#pragma unroll
      for (int w = 0; w < 3; ++w) {
        const unsigned c = (s_ocnt[4 * l + w] + (unsigned)tiles_per_unit(4 * l + w) - 1u) / (unsigned)tiles_per_unit(4 * l + w);  // units of the list
        if (v == w && r >= c) { r -= c; v = w + 1; }
      }
      const int f = 4 * l + v;
      const unsigned tf = (unsigned)tiles_per_unit(f);
      const unsigned first = r * tf;
      nt_o = (int)min(tf, s_ocnt[f] - first);
      base_o = (size_t)((f & (kTileQueues - 1)) * kTileBuckets + (kTileBuckets - 1 - (f >> 3))) * list_cap + first;
    } else {
      const unsigned t0 = uu * (unsigned)T;
      nt_o = (int)min((unsigned)T, n_tiles - t0);
      base_o = t0;
    }
  };
  auto load_hdr = [&](int who, int nt_x, size_t base_x, unsigned &tile_o, unsigned &cnt_o) {  // who < nt_x: tile and its pair count
    tile_o = 0;
    cnt_o = 0;
    if (who < nt_x) {
      if (use_pc) {
        const uint2 e = a.pc_list[base_x + who];
        tile_o = e.x;
        cnt_o = e.y;
      } else if (a.bucket_cnt) {
        const uint4 e = reinterpret_cast<const uint4 *>(a.bucket_list)[base_x + who];
        tile_o = e.x;
        cnt_o = e.w;
      } else {
        tile_o = (unsigned)base_x + (unsigned)who;
        cnt_o = a.sp_cnt[tile_o];
      }
    }
  };
  unsigned long long n_pairs_wg = 0;  // pair statistic (lanes < T of the last wave; the first unit: of the first)
  // Unit queues: unit u belongs to queue u % 8; a workgroup's first unit is static (its index; gridDim.x is a multiple of
  // 8, so it is of queue index % 8), all later ones are claims on the counters: claim c of queue q is unit
  // (gridDim.x / 8 + c) * 8 + q.  An exhausted queue sends the workgroup to the next one that still looks open (plain
  // loads of the counters; a counter only grows).
  // The claim is LATE (issued by the last wave before its last round of the current unit, a unit claimed early sits with a
  // workgroup that may be in a 45 us unit while the others run dry -- the claim two units ahead that would hide every
  // latency cost 10 us of tail); the last wave resolves it behind the barrier that ends the rounds and fetches the next
  // unit's header while it does its share of the sums.
  int myq = (int)(blockIdx.x & 7u);
  auto resolve_claim = [&](unsigned c, int &q_io) -> unsigned {  // one lane
    unsigned un = ((gridDim.x >> 3) + c) * 8u + (unsigned)q_io;
    int tries = 0;
    while (un >= n_units && tries < 16) {
      ++tries;
      int q2 = -1;
      for (int d = 1; d < 8; ++d) {
        const int qq = (q_io + d) & 7;
        const unsigned seen = __hip_atomic_load(&a.sp_counters[32 * (1 + qq)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (((gridDim.x >> 3) + seen) * 8u + (unsigned)qq < n_units) { q2 = qq; break; }
      }
      if (q2 < 0) break;
      q_io = q2;
      const unsigned c3 = atomicAdd(&a.sp_counters[32 * (1 + q2)], 1u);
      un = ((gridDim.x >> 3) + c3) * 8u + (unsigned)q2;
    }
    return un;
  };
  const bool last_wave = tid >= kThreads - 64;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  (void)last_wave;
  unsigned u_cur = blockIdx.x;
  // the unit's header (tile and pair count of its tiles) sits in lanes < nt of EVERY wave: no LDS copy, no barrier for it
  int nt;
  unsigned h_tile, h_cnt;
  {
    size_t base;
    locate(u_cur, nt, base);
    load_hdr(lane, nt, base, h_tile, h_cnt);
    if (tid < nt) n_pairs_wg += (unsigned long long)h_cnt;
  }
  // the tables are zero whenever a unit starts: cleared here once, after that every wave zeroes what it sums
  for (int k = tid; k < T * max_nb * 64; k += kThreads) S[k] = 0ull;
  while (nt > 0) {
    LT_TRACE_MARK(3, u_cur, 0);
    // the unit's pairs as one list: slot parts first (offsets from the tiles' counts), then the overflow chains
    int off[kChunkTiles + 1];
    off[0] = 0;
#pragma unroll
    for (int k = 0; k < kChunkTiles; ++k)
      off[k + 1] = off[k] + (k < nt ? min((int)(unsigned)__builtin_amdgcn_readlane((int)h_cnt, k), cap) : 0);
    auto entry_of = [&](int p, int &k) -> uint4 {
      k = 0;
#pragma unroll
      for (int m = 1; m < kChunkTiles; ++m) k += (p >= off[m]) ? 1 : 0;
      int o = off[0];
#pragma unroll
      for (int m = 1; m < kChunkTiles; ++m) o = (k >= m) ? off[m] : o;
      const unsigned tile = (unsigned)__shfl((int)h_tile, k);
      return a.sp_slots[(size_t)tile * (size_t)cap + (p - o)];
    };
    // first load level of the unit, all in flight together: this thread's first pair entry, its candidates' records
    const int total = off[kChunkTiles];
    int k0 = 0;
    uint4 e0 = uint4{0u, 0u, 0u, 0u};
    const bool has0 = tid < total;
    if (total > 0) {
      int kk = 0;
      const uint4 ee = entry_of(has0 ? tid : 0, kk);  // (every lane takes part in the shuffle; lanes without a pair read entry 0)
      if (has0) { e0 = ee; k0 = kk; }
    }
    CandMeta mt[2];
    long long pos[2];
    unsigned rrec[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int ti = wave + kWaves * h;  // the tile whose candidates (table rows) this wave sums
      pos[h] = -1;
      rrec[h] = 0;
      mt[h] = CandMeta{0u, 0u, 0u, 0u};
      if (ti < nt) {
        const long long p = (long long)(unsigned)__builtin_amdgcn_readlane((int)h_tile, ti & (kChunkTiles - 1)) * 64 + lane;
        if (p < C) {
          pos[h] = p;
          if (score_by == 0) rrec[h] = a.perm ? a.perm[p] : (unsigned)p;
          mt[h] = a.meta[p];
        }
      }
    }
    __syncthreads();  // every wave has summed and zeroed its tables of the previous unit
    LT_TRACE_MARK(3, u_cur, 1);
    // summation order (image-id order of the neighbour slots) of the image of each of this wave's tiles, one entry per lane:
    // the sums then take it by readlane instead of one global load per neighbour and lane (round 6: the sums were 23 % of a
    // unit's time, twenty dependent L1 round trips).  Lanes of another image (a tile that straddles two) read it from memory.
    // (the lanes' own neighbour-table base and count are taken out of the prologue records HERE: a first use after the next
    // unit's header loads were issued made the sums wait for those loads -- the memory counter is in order)
    int ordv[2] = {0, 0};
    long long wnb0[2] = {-1, -1};
    unsigned nb0_l[2] = {0u, 0u};
    int nn_l[2] = {0, 0};
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if (wave + kWaves * h < nt) {  // (wave-uniform; lane 0 of a listed tile is a candidate)
        nb0_l[h] = mt[h].nb >> 8;
        nn_l[h] = pos[h] >= 0 ? (int)(mt[h].nb & 0xFFu) : 0;
        wnb0[h] = (long long)(unsigned)__builtin_amdgcn_readfirstlane((int)nb0_l[h]);
        const int n0 = __builtin_amdgcn_readfirstlane(nn_l[h]);
        if (lane < n0) ordv[h] = a.blk_order[wnb0[h] + lane];
        if (n0 > 64) wnb0[h] = -1;  // more neighbours than lanes: every lane reads its order from memory
      }
    }
    // iterations of the last wave over the unit's list: it claims the next unit before its last one (eight counters: one
    // counter took 2 300 claims in 60 us and the claims came back after up to 9 us)
    const int n_it3 = total > kThreads - 64 ? (total - (kThreads - 64) + kThreads - 1) / kThreads : 0;
    unsigned c2 = 0;
    auto claim = [&]() {
      if (tid == kThreads - 64) c2 = atomicAdd(&a.sp_counters[32 * (1 + myq)], 1u);
    };
    if (n_it3 <= 1) claim();
    auto eval = [&](const uint4 e, const int k) {
      const CRec &ci = a.cand[e.y];
      const CRec &cj = a.cand[e.x];
      const int nbs_j = cj.nb_slot;
#if defined(LT_ABL_DENSE) && LT_ABL_DENSE == 1  // developer ablation: the loads without the arithmetic
      const double sc = 0.75 + 1e-300 * (ci.s[0] + cj.seg[3] + ci.depth[1] + cj.dir[2]);
#elif defined(LT_ABL_DENSE) && LT_ABL_DENSE == 2  // neither the record gathers nor the arithmetic
      const double sc = 0.75;
#else
      const d3 si_ = mk3(ci.s[0], ci.s[1], ci.s[2]), ei_ = mk3(ci.e[0], ci.e[1], ci.e[2]), di_ = mk3(ci.dir[0], ci.dir[1], ci.dir[2]);
      const d3 sj_ = mk3(cj.s[0], cj.s[1], cj.s[2]), ej_ = mk3(cj.e[0], cj.e[1], cj.e[2]), dj_ = mk3(cj.dir[0], cj.dir[1], cj.dir[2]);
      const Cam &camj = a.cams[(int)((unsigned)nbs_j >> 8)];
      double sc;
      if constexpr (kFast) sc = pair_score_fused(cfg, si_, ei_, di_, ci.depth[0], ci.depth[1], sj_, ej_, dj_, cj.seg, camj);
      else sc = pair_score_terms(cfg, si_, ei_, di_, ci.depth[0], ci.depth[1], sj_, ej_, dj_, cj.seg, camj);
#endif
      if (sc > 0.0)
        atomicMax(&S[(k * max_nb + (nbs_j & 0xFF)) * 64 + (int)e.z], (unsigned long long)__double_as_longlong(sc));
    };
    // (ONE inlined instance of pair_score for the unit's list: the first entry was fetched with the unit's first load level)
    const int n_it = total > (wave << 6) ? (total - (wave << 6) + kThreads - 1) / kThreads : 0;  // of this wave (wave-uniform)
    for (int it = 0; it < n_it; ++it) {
      if (it > 0 && it == n_it3 - 1) claim();
      const int p = tid + it * kThreads;
      int k = k0;
      uint4 e = e0;
      if (it > 0) e = entry_of(p < total ? p : 0, k);
      if (p < total) eval(e, k);
    }
    for (int k = 0; k < nt; ++k) {
      if ((int)(unsigned)__builtin_amdgcn_readlane((int)h_cnt, k & (kChunkTiles - 1)) <= cap) continue;
      unsigned cc = a.sp_ovf[(unsigned)__builtin_amdgcn_readlane((int)h_tile, k & (kChunkTiles - 1))];
      while (cc != kNoChunk) {
        const uint2 d = a.sp_desc[cc];
        for (int p = tid; p < (int)d.x; p += kThreads) eval(a.sp_pairs[(size_t)cc * kChunkCap + p], k);
        cc = d.y;
      }
    }
    if (tid == kThreads - 64) {  // the claimed unit, for everybody behind the barrier
      int q_n = myq;
      s_next[0] = resolve_claim(c2, q_n);
      s_next[1] = (unsigned)q_n;
    }
    __syncthreads();  // every pair of the unit is in the tables
    LT_TRACE_MARK(3, u_cur, 2);
    // the next unit's header: in flight during the sums.  (The order vectors were loaded before the rounds and are first read
    // in the sums: without this use in front of the header loads the compiler waits for THEM there -- vmcnt(0), the counter
    // is in order -- and the sums started with the header's round trip, 2 us per unit.)
    asm volatile("" : "+v"(ordv[0]), "+v"(ordv[1]));
    const unsigned un = (unsigned)__builtin_amdgcn_readfirstlane((int)s_next[0]);
    myq = __builtin_amdgcn_readfirstlane((int)s_next[1]);
    int nt_n;
    size_t base_n;
    unsigned n_tile, n_cnt;
    locate(un, nt_n, base_n);
    load_hdr(lane, nt_n, base_n, n_tile, n_cnt);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int ti = wave + kWaves * h;
      if (ti >= nt) continue;  // wave-uniform
      const bool act = pos[h] >= 0;
      const long long nb0 = (long long)nb0_l[h];
      const int n_nb = nn_l[h];
      const bool own = nb0 == wnb0[h];
      const int n_max = wave_max_i32(n_nb);  // the loop is wave-uniform: readlane takes the order whatever the lanes' state
      // kSumChunk cells at a time: the reads first (independent of each other), then the ordered additions, then the zeroes -- as
      // one read / add / write per neighbour the compiler kept every read behind the previous write (LDS, may alias): twenty
      // LDS round trips in a row, 2.4 us per unit.  Cells beyond the lane's own neighbours add +0.0 (exact).
      double sum = 0.0;
      // (two copies of the loop: the common one -- every lane of the tile belongs to the image whose order the wave holds --
      // contains no global load, so nothing in it waits on the memory counter, on which the NEXT unit's header loads issued
      // just above are still outstanding: with the order read from memory inside the loop the sums began with a wait for
      // that header, 2 us per unit)
      if (!__any(act && !own)) {
        for (int k0 = 0; k0 < n_max; k0 += kSumChunk) {
          unsigned long long v[kSumChunk];
          int sl[kSumChunk];
#pragma unroll
          for (int u = 0; u < kSumChunk; ++u) {
            const int k = k0 + u;
            sl[u] = __builtin_amdgcn_readlane(ordv[h], k & 63);
            v[u] = k < n_nb ? S[(ti * max_nb + sl[u]) * 64 + lane] : 0ull;
          }
#pragma unroll
          for (int u = 0; u < kSumChunk; ++u) sum += __longlong_as_double((long long)v[u]);
#pragma unroll
          for (int u = 0; u < kSumChunk; ++u)  // (a candidate's pairs only touch the slots of its image's neighbours: the table is clean again)
            if (k0 + u < n_nb) S[(ti * max_nb + sl[u]) * 64 + lane] = 0ull;
        }
      } else {
        for (int k = 0; k < n_nb; ++k) {
          unsigned long long *cell = &S[(ti * max_nb + a.blk_order[nb0 + k]) * 64 + lane];
          sum += __longlong_as_double((long long)*cell);
          *cell = 0ull;
        }
      }
      if (act) a.score[score_by == 1 ? pos[h] : (score_by == 2 ? (long long)a.spos[pos[h]] : (long long)rrec[h])] = sum;
    }
    LT_TRACE_MARK(3, u_cur, 3);
    u_cur = un;
    nt = nt_n;
    h_tile = n_tile;
    h_cnt = n_cnt;
    if (tid < nt) n_pairs_wg += (unsigned long long)h_cnt;
  }
  if (a.pair_counter && tid < 64) {
    for (int d = 32; d >= 1; d >>= 1) n_pairs_wg += (unsigned long long)__shfl_xor((long long)n_pairs_wg, d);
    if (tid == 0 && n_pairs_wg) atomicAdd(a.pair_counter, n_pairs_wg);
  }
}

// k_dense8 for the depth-sorted tiles of the exhaustive mode (round 6, last part).  A tile there carries ~25 pairs on ~20 of
// its 64 candidates; with one 64 x max_nb table of maxima per tile (10 KB) a unit was three tiles -- 75 pairs for four waves, and
// 1.1e5 units each paying its header, entries and sums (1.13 ms, 128 ns per 1 000 pairs against 67 in matched mode).  Here a
// unit is kChunkTiles consecutive tiles and the table has a ROW only for a candidate that has a pair: a first pass over the
// unit's entries sets the tiles' 64-bit masks, row = (rows of the tiles before) + (mask bits below the lane); the rows of a
// unit (~160) fit the same 36 KB.  A unit whose rows do not fit is worked off in groups of tiles that do.  Units in natural
// order (the exhaustive mode has no cost-class lists), claimed as in k_dense8.  Same arithmetic, same order of the sums.
constexpr int kRowsHdrBytes = 16 + kChunkTiles * 8 + (kChunkTiles + 8) * 4;
template <bool kFast>
__global__ void __launch_bounds__(64 * kDenseWaves) LT_DENSE_OCC
k_dense_rows(Score3Args a, ScoreCfg cfg, int score_by, int rows_cap) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  unsigned *s_next = reinterpret_cast<unsigned *>(smem_raw);                              // [4]
  unsigned long long *s_mask = reinterpret_cast<unsigned long long *>(smem_raw + 16);     // [kChunkTiles] candidates with a pair
  unsigned long long *S = reinterpret_cast<unsigned long long *>(smem_raw + ((kRowsHdrBytes + 15) & ~15));
  if (a.err_flag && *a.err_flag == kErrPairChunks) return;  // the overflow store was full: the run is repeated
  const long long C = a.tri_off[a.G];
  const unsigned n_tiles = (unsigned)((C + 63) >> 6);
  constexpr int T = kChunkTiles;
  constexpr int kWaves = kDenseWaves;
  constexpr int kThreads = 64 * kWaves;
  static_assert(kChunkTiles * 64 <= 2 * kThreads, "two tiles per wave in the sums");
  const int tid = threadIdx.x;
  const int lane = lane_id();
  const unsigned n_units = (n_tiles + (unsigned)T - 1) / (unsigned)T;
  const int max_nb = a.max_nb;
  const int cap = a.sp_slot_cap;
  unsigned long long n_pairs_wg = 0;
  int myq = (int)(blockIdx.x & 7u);
  auto resolve_claim = [&](unsigned c, int &q_io) -> unsigned {  // one lane; as in k_dense8
    unsigned un = ((gridDim.x >> 3) + c) * 8u + (unsigned)q_io;
    int tries = 0;
    while (un >= n_units && tries < 16) {
      ++tries;
      int q2 = -1;
      for (int d = 1; d < 8; ++d) {
        const int qq = (q_io + d) & 7;
        const unsigned seen = __hip_atomic_load(&a.sp_counters[32 * (1 + qq)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (((gridDim.x >> 3) + seen) * 8u + (unsigned)qq < n_units) { q2 = qq; break; }
      }
      if (q2 < 0) break;
      q_io = q2;
      const unsigned c3 = atomicAdd(&a.sp_counters[32 * (1 + q2)], 1u);
      un = ((gridDim.x >> 3) + c3) * 8u + (unsigned)q2;
    }
    return un;
  };
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  auto header = [&](unsigned uu, int &nt_o, unsigned &tile_o, unsigned &cnt_o) {  // lanes < nt of every wave: tile and pair count
    nt_o = 0; tile_o = 0; cnt_o = 0;
    if (uu >= n_units) return;
    const unsigned t0 = uu * (unsigned)T;
    nt_o = (int)min((unsigned)T, n_tiles - t0);
    if (lane < nt_o) {
      tile_o = t0 + (unsigned)lane;
      cnt_o = a.sp_cnt[tile_o];
    }
  };
  unsigned u_cur = blockIdx.x;
  int nt;
  unsigned h_tile, h_cnt;
  header(u_cur, nt, h_tile, h_cnt);
  if (tid < nt) n_pairs_wg += (unsigned long long)h_cnt;
  for (int k = tid; k < rows_cap * max_nb; k += kThreads) S[k] = 0ull;
  while (nt > 0) {
    int off[kChunkTiles + 1];
    off[0] = 0;
#pragma unroll
    for (int k = 0; k < kChunkTiles; ++k)
      off[k + 1] = off[k] + (k < nt ? min((int)(unsigned)__builtin_amdgcn_readlane((int)h_cnt, k), cap) : 0);
    auto entry_of = [&](int p, int &k) -> uint4 {
      k = 0;
#pragma unroll
      for (int m = 1; m < kChunkTiles; ++m) k += (p >= off[m]) ? 1 : 0;
      int o = off[0];
#pragma unroll
      for (int m = 1; m < kChunkTiles; ++m) o = (k >= m) ? off[m] : o;
      return a.sp_slots[(size_t)(u_cur * (unsigned)T + (unsigned)k) * (size_t)cap + (p - o)];  // (tiles of a unit are consecutive)
    };
    const int total = off[kChunkTiles];
    if (tid < kChunkTiles) s_mask[tid] = 0ull;
    // the candidates (table rows) this wave sums: tiles wave and wave + kWaves of the unit
    CandMeta mt[2];
    long long pos[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int ti = wave + kWaves * h;
      pos[h] = -1;
      mt[h] = CandMeta{0u, 0u, 0u, 0u};
      if (ti < nt) {
        const long long p = (long long)(unsigned)__builtin_amdgcn_readlane((int)h_tile, ti & (kChunkTiles - 1)) * 64 + lane;
        if (p < C) {
          pos[h] = p;
          mt[h] = a.meta[p];
        }
      }
    }
    __syncthreads();  // masks cleared; every wave has summed and zeroed its rows of the previous unit
    // ---- pass 1: which candidates have a pair (the entries are read again in pass 2: L2) ----
    for (int p = tid; p < total; p += kThreads) {
      int k;
      const uint4 e = entry_of(p, k);
      atomicOr(&s_mask[k], 1ull << (e.z & 63u));
    }  // (entry_of holds no cross-lane operation: the loop may diverge)
    for (int k = 0; k < nt; ++k) {
      if ((int)(unsigned)__builtin_amdgcn_readlane((int)h_cnt, k & (kChunkTiles - 1)) <= cap) continue;
      unsigned cc = a.sp_ovf[(unsigned)__builtin_amdgcn_readlane((int)h_tile, k & (kChunkTiles - 1))];
      while (cc != kNoChunk) {
        const uint2 d = a.sp_desc[cc];
        for (int p = tid; p < (int)d.x; p += kThreads) atomicOr(&s_mask[k], 1ull << (a.sp_pairs[(size_t)cc * kChunkCap + p].z & 63u));
        cc = d.y;
      }
    }
    int ordv[2] = {0, 0};
    long long wnb0[2] = {-1, -1};
    unsigned nb0_l[2] = {0u, 0u};
    int nn_l[2] = {0, 0};
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if (wave + kWaves * h < nt) {
        nb0_l[h] = mt[h].nb >> 8;
        nn_l[h] = pos[h] >= 0 ? (int)(mt[h].nb & 0xFFu) : 0;
        wnb0[h] = (long long)(unsigned)__builtin_amdgcn_readfirstlane((int)nb0_l[h]);
        const int n0 = __builtin_amdgcn_readfirstlane(nn_l[h]);
        if (lane < n0) ordv[h] = a.blk_order[wnb0[h] + lane];
        if (n0 > 64) wnb0[h] = -1;
      }
    }
    __syncthreads();  // the masks are complete
    unsigned long long mk[kChunkTiles];
    int rb[kChunkTiles + 1];  // rows of the unit's tiles before tile k
    rb[0] = 0;
#pragma unroll
    for (int k = 0; k < kChunkTiles; ++k) {
      mk[k] = k < nt ? s_mask[k] : 0ull;
      rb[k + 1] = rb[k] + __popcll(mk[k]);
    }
    unsigned c2 = 0;
    bool claimed = false;
    auto eval = [&](const uint4 e, const int k, const int row0) {
      const CRec &ci = a.cand[e.y];
      const CRec &cj = a.cand[e.x];
      const int nbs_j = cj.nb_slot;
      const d3 si_ = mk3(ci.s[0], ci.s[1], ci.s[2]), ei_ = mk3(ci.e[0], ci.e[1], ci.e[2]), di_ = mk3(ci.dir[0], ci.dir[1], ci.dir[2]);
      const d3 sj_ = mk3(cj.s[0], cj.s[1], cj.s[2]), ej_ = mk3(cj.e[0], cj.e[1], cj.e[2]), dj_ = mk3(cj.dir[0], cj.dir[1], cj.dir[2]);
      const Cam &camj = a.cams[(int)((unsigned)nbs_j >> 8)];
      double sc;
      if constexpr (kFast) sc = pair_score_fused(cfg, si_, ei_, di_, ci.depth[0], ci.depth[1], sj_, ej_, dj_, cj.seg, camj);
      else sc = pair_score_terms(cfg, si_, ei_, di_, ci.depth[0], ci.depth[1], sj_, ej_, dj_, cj.seg, camj);
      if (sc > 0.0) {
        unsigned long long m = 0ull;
        int r0 = 0;
#pragma unroll
        for (int q = 0; q < kChunkTiles; ++q) {
          m = (k == q) ? mk[q] : m;
          r0 = (k == q) ? rb[q] : r0;
        }
        const int row = r0 - row0 + __popcll(m & ((1ull << (e.z & 63u)) - 1ull));
        atomicMax(&S[row * max_nb + (nbs_j & 0xFF)], (unsigned long long)__double_as_longlong(sc));
      }
    };
    // groups of consecutive tiles whose rows fit the table (nearly always the whole unit)
    int ka = 0;
    while (ka < nt) {
      int kb = ka + 1;
#pragma unroll
      for (int q = 1; q <= kChunkTiles; ++q)
        if (q > ka && q <= nt && rb[q] - rb[ka] <= rows_cap) kb = q;  // (the largest prefix that fits; one tile always does)
      int p_lo = 0, p_hi = 0, row0 = 0;
#pragma unroll
      for (int q = 0; q <= kChunkTiles; ++q) {
        p_lo = (q == ka) ? off[q] : p_lo;
        p_hi = (q == kb) ? off[q] : p_hi;
        row0 = (q == ka) ? rb[q] : row0;
      }
      if (kb >= nt && !claimed) {  // the last group: claim the next unit under its rounds
        claimed = true;
        if (tid == kThreads - 64) c2 = atomicAdd(&a.sp_counters[32 * (1 + myq)], 1u);
      }
      for (int p = p_lo + tid; p < p_hi; p += kThreads) {
        int k;
        const uint4 e = entry_of(p, k);
        eval(e, k, row0);
      }
      for (int k = ka; k < kb; ++k) {
        if ((int)(unsigned)__builtin_amdgcn_readlane((int)h_cnt, k & (kChunkTiles - 1)) <= cap) continue;
        unsigned cc = a.sp_ovf[(unsigned)__builtin_amdgcn_readlane((int)h_tile, k & (kChunkTiles - 1))];
        while (cc != kNoChunk) {
          const uint2 d = a.sp_desc[cc];
          for (int p = tid; p < (int)d.x; p += kThreads) eval(a.sp_pairs[(size_t)cc * kChunkCap + p], k, row0);
          cc = d.y;
        }
      }
      if (kb >= nt && tid == kThreads - 64) {
        int q_n = myq;
        s_next[0] = resolve_claim(c2, q_n);
        s_next[1] = (unsigned)q_n;
      }
      __syncthreads();  // every pair of the group is in the table
      // ---- the sums of the group's tiles: a lane with a row reads it in its image's order and leaves it zero ----
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int ti = wave + kWaves * h;
        if (ti >= nt || ti < ka || ti >= kb) continue;  // wave-uniform
        const bool act = pos[h] >= 0;
        unsigned long long m = 0ull;
        int r0 = 0;
#pragma unroll
        for (int q = 0; q < kChunkTiles; ++q) {
          m = (ti == q) ? mk[q] : m;
          r0 = (ti == q) ? rb[q] : r0;
        }
        const bool has = ((m >> lane) & 1ull) != 0ull;
        const int row = r0 - row0 + __popcll(m & ((1ull << lane) - 1ull));
        const long long nb0 = (long long)nb0_l[h];
        const int n_nb = has ? nn_l[h] : 0;
        const bool own = nb0 == wnb0[h];
        const int n_max = wave_max_i32(n_nb);
        double sum = 0.0;
        if (!__any(has && !own)) {
          for (int k0 = 0; k0 < n_max; k0 += kSumChunk) {
            unsigned long long v[kSumChunk];
            int sl[kSumChunk];
#pragma unroll
            for (int u = 0; u < kSumChunk; ++u) {
              const int k = k0 + u;
              sl[u] = __builtin_amdgcn_readlane(ordv[h], k & 63);
              v[u] = k < n_nb ? S[row * max_nb + sl[u]] : 0ull;
            }
#pragma unroll
            for (int u = 0; u < kSumChunk; ++u) sum += __longlong_as_double((long long)v[u]);
#pragma unroll
            for (int u = 0; u < kSumChunk; ++u)
              if (k0 + u < n_nb) S[row * max_nb + sl[u]] = 0ull;
          }
        } else {
          for (int k = 0; k < n_nb; ++k) {
            unsigned long long *cell = &S[row * max_nb + a.blk_order[nb0 + k]];
            sum += __longlong_as_double((long long)*cell);
            *cell = 0ull;
          }
        }
        if (act) a.score[score_by == 1 ? pos[h] : (score_by == 2 ? (long long)a.spos[pos[h]] : (long long)(a.perm ? a.perm[pos[h]] : (unsigned)pos[h]))] = sum;
      }
      ka = kb;
      if (ka < nt) __syncthreads();  // the next group's rows start from a clean table
    }
    const unsigned un = (unsigned)__builtin_amdgcn_readfirstlane((int)s_next[0]);
    myq = __builtin_amdgcn_readfirstlane((int)s_next[1]);
    u_cur = un;
    header(u_cur, nt, h_tile, h_cnt);
    if (tid < nt) n_pairs_wg += (unsigned long long)h_cnt;
  }
  if (a.pair_counter && tid < 64) {
    for (int d = 32; d >= 1; d >>= 1) n_pairs_wg += (unsigned long long)__shfl_xor((long long)n_pairs_wg, d);
    if (tid == 0 && n_pairs_wg) atomicAdd(a.pair_counter, n_pairs_wg);
  }
}

// ---------------------------------------------------------------------------------------------
// Round 6, second half: the scoring stage as ONE kernel -- k_score_q.  The default of the matched mode.
// ---------------------------------------------------------------------------------------------
// The two kernels above run one after the other: 41 us in which every wave of the chip waits on window gathers (VALU issue
// 0.44), the last 15 of them with a growing part of the chip idle (a wave has two or three tiles), then k_dense8's start
// and 50 us of pair_score rounds behind LDS maxima and barriers (0.45).  Here ONE grid of workgroups of four waves does both:
//   sweep role   -- a workgroup sweeps its share of the queue's tiles (the code of k_score3<split>: window in LDS in single
//                   precision, two guards, the passing pairs to the tile's slot), one tile per wave at a time, and PUBLISHES
//                   every finished tile as (tile, pair count) in the FIFO of its XCD, at the tile's place in the queue's
//                   cost-class order;
//   dense role   -- when its share is swept the same workgroup turns to units of T consecutive FIFO entries (the code of
//                   k_dense8: one list of pairs over the unit's tiles, pair_score one pair per lane, per-image maxima in LDS,
//                   ordered sums), claimed through the queue's counter, most expensive tiles first, and waits -- bounded --
//                   for a unit's tiles where they are not published yet.
// The workgroups finish their shares between 25 and 40 us; the dense role starts in every slot the moment it is free, on
// tiles other workgroups published long before, and the sweep's tail and the dense kernel's start disappear.
//   Producer and consumer of a tile are ALWAYS on the same XCD (queue x = blockIdx.x % 8, the dispatch order of the
// hardware; checked against the XCC_ID register, device flag 8 sends the context back to the two-kernel form): the pair
// entries travel through that XCD's L2 with plain stores and loads, only the 8-byte FIFO entry is a device-scope store /
// load, and the overflow chains are kept in band (word 3 of an entry) because sp_ovf / sp_desc pack many tiles into cache
// lines a consumer's CU may already hold.  Nothing waits on a consumer: the slots of all tiles exist up front.  Same bits as
// the two-kernel form and as the fused kernel (tests/test_gpu_guards.py).
//   MEASURED (profiles/r06_score_experiments.txt, items 8-14): 86.0 us for the stage against 95.4 in the two-kernel form at
// 100 x 500, 0.593 against 0.659 ms at config 3.  What did NOT work on the way: workgroups ALTERNATING between the roles
// pass by pass (equal to the two kernels: they stay in step, all sweep, then all evaluate), tiles drawn from device
// counters (the draws' round trips at the start), FIFO places in the order the tiles finish (the heaviest tiles finish
// last: their units were claimed second and ended the kernel), part of the slots dense from the start.
// ---------------------------------------------------------------------------------------------
constexpr int kErrXcdMap = 8;   // device error flag: workgroups of one queue ran on different XCDs, or a FIFO wait timed out
constexpr int kQHdrBytes = 64 + 2 * kTileBuckets * 4;  // [0] the claimed unit, [1] abort; from [16]: class sizes and their inclusive prefix
constexpr int kQCap = 352;       // the sweep role's pair queue (entries of 8 bytes; emptied into the tile's slot from kQCap - 256 on)
constexpr int kQSweepWaveBytes = kWin * 48 + kQCap * 8 + 64 * 4;
constexpr unsigned kFifoEmpty = 0xFFFFFFFFu;
constexpr int kQSpinMax = 1 << 11;  // ~2 ms of waiting for one FIFO entry (a tile takes ~10 us): see the sweep role

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_void_t;
static __device__ __forceinline__ uint2 fifo_load(const uint2 *p) {
  const unsigned long long v = __hip_atomic_load(reinterpret_cast<const unsigned long long *>(p), __ATOMIC_RELAXED,
                                                 __HIP_MEMORY_SCOPE_AGENT);
  return make_uint2((unsigned)v, (unsigned)(v >> 32));
}
static __device__ __forceinline__ void fifo_store(uint2 *p, unsigned tile, unsigned cnt) {
  __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), (unsigned long long)tile | ((unsigned long long)cnt << 32),
                     __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

static __device__ __forceinline__ unsigned &q_word3(uint4 *e) { return reinterpret_cast<unsigned *>(e)[3]; }
// first load level of a tile: prologue record, record index, the record indices of the first window chunk -- loaded while
// the dense role's sums of the unit before it run (k_score_q), or at the start of the tile
struct QFirst {
  uint4 hdr;  // the tile's class-list entry: tile, first and end position of its window
  CandMeta mt;
  unsigned p_i, w0, w1;
};
template <bool kPerm>
static __device__ __forceinline__ QFirst q_first_level(const Score3Args &a, const uint4 hdr, const long long C, const int lane) {
  QFirst f;
  f.hdr = hdr;
  f.mt = CandMeta{0u, 0u, 0u, 0u};
  f.p_i = 0; f.w0 = 0; f.w1 = 0;
  const long long tpos = (long long)hdr.x * 64 + lane;
  if (tpos < C) {
    f.mt = a.meta[tpos];
    f.p_i = kPerm ? a.perm[tpos] : (unsigned)tpos;
  }
  if (kPerm && hdr.z > hdr.y) {
    const unsigned w = (hdr.z - hdr.y) < (unsigned)kWin ? (hdr.z - hdr.y) : (unsigned)kWin;
    if ((unsigned)lane < w) f.w0 = a.perm[(size_t)hdr.y + lane];
    if ((unsigned)lane + 64u < w) f.w1 = a.perm[(size_t)hdr.y + 64 + lane];
  }
  return f;
}
// sweep role: one wave, one tile (k_score3<true, false, kPerm, true> without the schedule around it)
template <bool kPerm, class Pre>
static __device__ __forceinline__ void q_sweep_tile(const Score3Args &a, const ScoreCfg &cfg, const double scaleinv_guard2,
                                                    unsigned char *smem_raw, const QFirst &fl, const long long C,
                                                    const int lane, uint2 *fifo, const unsigned f_idx, bool &ch_dead,
                                                    Pre &&prefetch_next) {  // called once, when the first window is staged
  // f_idx: the tile's place in the FIFO = its place in the queue's cost-class order (units are then runs of the ORDER, the
  // most expensive tiles first, whatever the order the tiles finish in: with places handed out at a tile's end the heaviest
  // tiles of the first pass -- the last to finish -- formed the units that were claimed second, at 45-50 us, and the longest
  // of them (40 us) ended the kernel at 88)
  const uint4 hdr = fl.hdr;
  float4 *W4 = reinterpret_cast<float4 *>(smem_raw);
  uint2 *queue2 = reinterpret_cast<uint2 *>(smem_raw + (size_t)kWin * 48);
  unsigned *reci_l = reinterpret_cast<unsigned *>(smem_raw + (size_t)kWin * 48 + (size_t)kQCap * 8);
  const unsigned tile = (unsigned)__builtin_amdgcn_readfirstlane((int)hdr.x);
  const unsigned h_lo = (unsigned)__builtin_amdgcn_readfirstlane((int)hdr.y);
  const unsigned h_hi = (unsigned)__builtin_amdgcn_readfirstlane((int)hdr.z);
  const bool h_bounds = h_hi > h_lo;
  const long long i0 = (long long)tile * 64;
  const long long tpos = i0 + lane;
  const bool active = tpos < C;
  LT_TRACE_MARK(2, tile, 0);
  const CandMeta p_mt = fl.mt;
  const unsigned p_i = fl.p_i, w0src = fl.w0, w1src = fl.w1;
  const long long i = active ? (long long)p_i : 0;
  long long off = 0;
  int n = 0, sloti = -1;
  double six = 0, siy = 0, siz = 0, eix = 0, eiy = 0, eiz = 0, gs2 = 0, ge2 = 0;
  float dixf = 0, diyf = 0, dizf = 0;
  if (active) {
    off = ((long long)p_mt.off_hi << 32) | (long long)p_mt.off_lo;
    n = (int)p_mt.n;
    const CRec &ci = a.cand[i];
    dixf = (float)ci.dir[0]; diyf = (float)ci.dir[1]; dizf = (float)ci.dir[2];
    sloti = ci.nb_slot;  // (the whole word: candidates of one node share the image, so equal slots = equal words)
    six = ci.s[0]; siy = ci.s[1]; siz = ci.s[2];
    eix = ci.e[0]; eiy = ci.e[1]; eiz = ci.e[2];
    const double zs = ci.depth[0] + kEps, ze = ci.depth[1] + kEps;
    gs2 = (zs > 0.0) ? scaleinv_guard2 * zs * zs : 1e300;
    ge2 = (ze > 0.0) ? scaleinv_guard2 * ze * ze : 1e300;
  }
  reci_l[lane] = (unsigned)i;
  const double ox = readlane_f64(six, 0), oy = readlane_f64(siy, 0), oz = readlane_f64(siz, 0);
  const float sixf = (float)(six - ox), siyf = (float)(siy - oy), sizf = (float)(siz - oz);
  const float eixf = (float)(eix - ox), eiyf = (float)(eiy - oy), eizf = (float)(eiz - oz);
  float ri = fmaxf(fmaxf(fmaxf(fabsf(sixf), fabsf(siyf)), fabsf(sizf)), fmaxf(fmaxf(fabsf(eixf), fabsf(eiyf)), fabsf(eizf)));
  if (!active) ri = 0.0f;
  const double gs = sqrt(gs2), ge = sqrt(ge2);
  const float cosf_guard = cfg.cos_guard > -1.0 ? (float)(cfg.cos_guard - 2e-6) : -2.0f;
  long long lo, hi;
  if (h_bounds) {
    lo = (long long)h_lo; hi = (long long)h_hi;
  } else {
    const int last = (int)((C - i0) < 64 ? (C - i0) : 64) - 1;
    lo = (long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)off);
    hi = (long long)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)(off + n), last);
  }
  int qn = 0, t_cnt = 0, ov_fill = 0;
  unsigned ov_cur = kNoChunk;
  bool prefetched = false;
  auto ch_alloc = [&]() -> unsigned {
    unsigned id = 0;
    if (lane == 0) id = atomicAdd(&a.sp_counters[0], 1u);
    id = (unsigned)__builtin_amdgcn_readfirstlane((int)id);
    if (id >= a.sp_chunk_cap) {
      ch_dead = true;
      if (lane == 0 && a.err_flag) atomicCAS(a.err_flag, 0, kErrPairChunks);
      return kNoChunk;
    }
    return id;
  };
  auto drain = [&](const bool final) {
    wave_lds_sync();
    const int k0 = min(qn, max(0, a.sp_slot_cap - t_cnt));
    uint4 *dst = a.sp_slots + (size_t)tile * (size_t)a.sp_slot_cap + t_cnt;
    for (int p = lane; p < k0; p += 64) {
      const uint2 e = queue2[p];
      dst[p] = uint4{e.x, reci_l[e.y & 63u], e.y, 0u};
    }
    int done = k0;
    while (done < qn && !ch_dead) {
      if (ov_cur == kNoChunk || ov_fill == kChunkCap) {
        const unsigned nxt = ch_alloc();
        if (ch_dead) break;
        if (lane == 0) {
          // the chain is kept IN BAND (word 3 of an entry is free): first chunk in entry 0 of the tile's slot, next chunk and
          // fill in entries 0 and 1 of a chunk.  sp_ovf / sp_desc pack many tiles / chunks into a cache line that a consumer's
          // CU may hold from an earlier read -- these lines belong to one tile and are read after it is published, once.
          if (ov_cur == kNoChunk) q_word3(a.sp_slots + (size_t)tile * (size_t)a.sp_slot_cap) = nxt;
          else {
            q_word3(a.sp_pairs + (size_t)ov_cur * kChunkCap) = nxt;
            q_word3(a.sp_pairs + (size_t)ov_cur * kChunkCap + 1) = (unsigned)ov_fill;
          }
        }
        ov_cur = nxt;
        ov_fill = 0;
      }
      const int k = min(qn - done, kChunkCap - ov_fill);
      uint4 *od = a.sp_pairs + (size_t)ov_cur * kChunkCap + ov_fill;
      for (int p = lane; p < k; p += 64) {
        const uint2 e = queue2[done + p];
        od[p] = uint4{e.x, reci_l[e.y & 63u], e.y, 0u};
      }
      ov_fill += k;
      done += k;
    }
    t_cnt += qn;
    qn = 0;
    if (final) {
      if (lane == 0) {
        a.sp_cnt[tile] = (unsigned)t_cnt;
        if (ov_cur != kNoChunk) {
          q_word3(a.sp_pairs + (size_t)ov_cur * kChunkCap) = kNoChunk;
          q_word3(a.sp_pairs + (size_t)ov_cur * kChunkCap + 1) = (unsigned)ov_fill;
        }
      }
      // the tile's entries (and chain) have reached the L2 before the FIFO entry can be seen.  A tile whose chain could
      // not even be started (the overflow store is full: the run is repeated) is published with what its slot holds.
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const unsigned pub = (ch_dead && ov_cur == kNoChunk) ? (unsigned)min(t_cnt, a.sp_slot_cap) : (unsigned)t_cnt;
      if (lane == 0) fifo_store(&fifo[f_idx], tile, pub);
    }
    wave_lds_sync();
  };
  for (long long wb = lo; wb < hi; wb += kWin) {
    wave_lds_sync();
    const int wn = (int)((hi - wb) < kWin ? (hi - wb) : kWin);
    float rw = ri;
    for (int e = lane; e < wn; e += 64) {
      const long long src = !kPerm ? wb + e
                            : ((h_bounds && wb == lo && e < 128) ? (long long)(e < 64 ? w0src : w1src) : (long long)a.perm[wb + e]);
      const CRec &c = a.cand[src];
      const float sx = (float)(c.s[0] - ox), sy = (float)(c.s[1] - oy), sz = (float)(c.s[2] - oz);
      const float ex = (float)(c.e[0] - ox), ey = (float)(c.e[1] - oy), ez = (float)(c.e[2] - oz);
      W4[3 * e + 0] = float4{(float)c.dir[0], (float)c.dir[1], (float)c.dir[2], __int_as_float(c.nb_slot)};
      W4[3 * e + 1] = float4{sx, ex, sy, ey};
      W4[3 * e + 2] = float4{sz, ez, __uint_as_float((unsigned)src), 0.0f};
      rw = fmaxf(rw, fmaxf(fmaxf(fmaxf(fabsf(sx), fabsf(sy)), fabsf(sz)), fmaxf(fmaxf(fabsf(ex), fabsf(ey)), fabsf(ez))));
    }
    rw = wave_max_f32_nan(rw);
    const double delta = 1e-6 * (double)rw;
    const float gsf = (float)((gs + delta) * (gs + delta) * (1.0 + 2e-6));
    const float gef = (float)((ge + delta) * (ge + delta) * (1.0 + 2e-6));
    wave_lds_sync();
    if (wb == lo) {
      LT_TRACE_MARK(2, tile, 1);
      prefetch_next();  // (the registers of this tile's first load level are free now)
      prefetched = true;
    }
    long long jlo = off > wb ? off : wb;
    long long jhi = (off + n) < (wb + wn) ? (off + n) : (wb + wn);
    const int cnt = (active && jhi > jlo) ? (int)(jhi - jlo) : 0;
    const int cmax = wave_max_i32(cnt);
    const int w0 = (int)(jlo - wb);
    const int self_t = (int)(tpos - jlo);
    const int wlast = cnt > 0 ? w0 + cnt - 1 : 0;
    const int wbase = cnt > 0 ? w0 : 0;
    for (int t = 0; t < cmax; t += 4) {
      float4 A[4], B[4];
      float2 E[4];
      unsigned Jr[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int w = min(wbase + t + u, wlast);
        A[u] = W4[3 * w + 0];
        B[u] = W4[3 * w + 1];
        const float4 e4 = W4[3 * w + 2];
        E[u] = make_float2(e4.x, e4.y);
        Jr[u] = __float_as_uint(e4.z);
      }
      bool pass[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float c = fabsf(__builtin_fmaf(dizf, A[u].z, __builtin_fmaf(diyf, A[u].y, dixf * A[u].x)));
        const float ax = sixf - B[u].x, bx = eixf - B[u].y;
        const float ay = siyf - B[u].z, by = eiyf - B[u].w;
        const float az = sizf - E[u].x, bz = eizf - E[u].y;
        const float ds2 = __builtin_fmaf(az, az, __builtin_fmaf(ay, ay, ax * ax));
        const float de2 = __builtin_fmaf(bz, bz, __builtin_fmaf(by, by, bx * bx));
        pass[u] = (t + u < cnt) & (t + u != self_t) & (__float_as_int(A[u].w) != sloti) & !(c < cosf_guard) &
                  !(ds2 > gsf) & !(de2 > gef);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const unsigned long long m = __ballot(pass[u]);
        if (m) {
          // (the neighbour word of j rides in the entry: the dense role fetches j's camera together with the records)
          if (pass[u]) queue2[qn + __popcll(m & lanemask_lt())] = make_uint2(Jr[u], (unsigned)lane | ((unsigned)__float_as_int(A[u].w) << 6));
          qn += __popcll(m);
        }
      }
      if (qn > kQCap - 256) drain(false);
    }
  }
  LT_TRACE_MARK(2, tile, 2);
  if (!prefetched) prefetch_next();
  drain(true);
  LT_TRACE_MARK(2, tile, 3);
}

#ifndef LT_Q_WAVES_PER_EU
#define LT_Q_WAVES_PER_EU 4
#endif
template <bool kFast, bool kPerm>
__global__ void __launch_bounds__(64 * kDenseWaves) __attribute__((amdgpu_waves_per_eu(LT_Q_WAVES_PER_EU, LT_Q_WAVES_PER_EU)))
k_score_q(Score3Args a, ScoreCfg cfg, double scaleinv_guard2, int score_by, unsigned n_sweep_wgs, int test_lose_tile) {
  extern __shared__ __align__(16) unsigned char smem_all[];
  unsigned *s_hdr = reinterpret_cast<unsigned *>(smem_all);
  unsigned long long *S = reinterpret_cast<unsigned long long *>(smem_all + kQHdrBytes);
  constexpr int kWaves = kDenseWaves;
  constexpr int kThreads = 64 * kWaves;
  const int tid = threadIdx.x;
  const int lane = lane_id();
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int x = (int)(blockIdx.x & (unsigned)(kTileQueues - 1));  // n_sweep_wgs is a multiple of 8: the same rule in both roles
  const long long C = a.tri_off[a.G];
  const int T = a.sp_t_max;
  const int max_nb = a.max_nb;
  const int cap = a.sp_slot_cap;
  uint2 *fifo = a.pc_list + (size_t)x * a.pc_cap;
  unsigned *q_claim = a.sp_counters + 32 * (1 + x);
  // the queue's tiles in class order, most expensive class first: entry l < kTileBuckets of the two LDS vectors is class
  // kTileBuckets - 1 - l (every wave writes the same values: no barrier needed for its own reads)
  unsigned *s_cls_cnt = s_hdr + 16, *s_cls_incl = s_hdr + 16 + kTileBuckets;
  unsigned n_x;
  {
    const unsigned cls_cnt = lane < kTileBuckets ? a.bucket_cnt[(x * kTileBuckets + (kTileBuckets - 1 - lane)) * 32] : 0u;
    const unsigned cls_incl = wave_incl_scan_u32(cls_cnt);
    n_x = (unsigned)__builtin_amdgcn_readlane((int)cls_incl, 63);
    if (lane < kTileBuckets) { s_cls_cnt[lane] = cls_cnt; s_cls_incl[lane] = cls_incl; }
    wave_lds_sync();
  }
  auto fetch_x = [&](const unsigned k) -> uint4 {  // k < n_x, wave-uniform
    const unsigned incl = lane < kTileBuckets ? s_cls_incl[lane] : 0xFFFFFFFFu;
    const int l = __builtin_ctzll(__ballot(incl > k));
    const unsigned base = s_cls_incl[l] - s_cls_cnt[l];
    return reinterpret_cast<const uint4 *>(a.bucket_list)[(size_t)(x * kTileBuckets + (kTileBuckets - 1 - l)) * a.bucket_cap + (k - base)];
  };
  unsigned n_pairs_wg = 0;  // pair statistic of this workgroup (dense role: lanes < T of the first wave)
  if (blockIdx.x < n_sweep_wgs) {
    // ---- sweep role.  The workgroup's SHARE of the queue's order is static as in k_score3<split> -- pass m takes entries
    // [m W, (m + 1) W) of the order (W = sweep waves of this queue), forwards for even m and backwards for odd m, this
    // workgroup the four entries of its four ranks -- and its four waves draw from that share through a counter in LDS (the
    // workgroup turns to the dense role when its last wave is through).  The next tile is drawn at the start of the current
    // one, its list entry fetched then and its first load level once the current window is staged.
    // (Measured and dropped: every wave drawing its tiles from per-queue device counters, 16 first-draw counters per queue
    // against the burst at the start -- no share is owed by a workgroup that is not resident, so nothing could ever wait on
    // one --: 98-100 us against 86, the draws' round trips at the start of 4 096 waves cost what the form gains.  With static
    // shares a grid that is not fully resident -- another kernel on the GPU -- can leave a unit waiting for a tile nobody
    // sweeps yet: the wait is bounded (kQSpinMax, ~2 ms), raises device flag 8 and the run is repeated in the two-kernel form.)
    const unsigned W_x = (n_sweep_wgs >> 3) * (unsigned)kWaves;
    const unsigned r0 = (blockIdx.x >> 3) * (unsigned)kWaves;
    unsigned char *smem_wave = smem_all + kQHdrBytes + (size_t)wave * kQSweepWaveBytes;
    if (tid == 0) s_hdr[2] = 0u;
    __syncthreads();
    bool ch_dead = false;
    auto draw = [&]() -> unsigned {  // -> this wave's next entry of the order, n_x when the share is used up
      for (;;) {
        unsigned idx = 0;
        if (lane == 0) idx = atomicAdd(&s_hdr[2], 1u);
        idx = (unsigned)__builtin_amdgcn_readfirstlane((int)idx);
        const unsigned pass = idx >> 2, rr = r0 + (idx & 3u);
        if ((unsigned long long)pass * W_x >= n_x) return n_x;
        const unsigned kk = pass * W_x + ((pass & 1u) ? W_x - 1u - rr : rr);
        if (kk < n_x) return kk;  // (a rank without an entry in the last pass: draw on)
      }
    };
    unsigned k = draw();
    QFirst fl;
    if (k < n_x) fl = q_first_level<kPerm>(a, fetch_x(k), C, lane);
    while (k < n_x) {
      const unsigned k_n = draw();
      uint4 hdr_n = uint4{0u, 0u, 0u, 0u};
      if (k_n < n_x) hdr_n = fetch_x(k_n);
      QFirst fl_n;
      fl_n.hdr = hdr_n; fl_n.mt = CandMeta{0u, 0u, 0u, 0u}; fl_n.p_i = 0; fl_n.w0 = 0; fl_n.w1 = 0;
      // (LT_TEST_Q_LOSE_TILE: the first tile of queue 0 is never swept -- its unit's wait runs into the bound, device flag 8)
      if (test_lose_tile && x == 0 && k == 0) {
        if (k_n < n_x) fl_n = q_first_level<kPerm>(a, hdr_n, C, lane);
      } else {
        q_sweep_tile<kPerm>(a, cfg, scaleinv_guard2, smem_wave, fl, C, lane, fifo, k, ch_dead,
                     [&]() { if (k_n < n_x) fl_n = q_first_level<kPerm>(a, hdr_n, C, lane); });
      }
      fl = fl_n;
      k = k_n;
    }
    __syncthreads();  // the four windows are done with: the tables of the dense role lie over them
  }
  {
    // ---- dense role (a sweep workgroup goes on here when its share is swept -- no new workgroup has to be dispatched into
    // its slot and set up --, a workgroup behind the sweep workgroups starts here): units of T consecutive FIFO entries, every one of them CLAIMED (claim c = unit c; the later claims are
    // issued before the last round of the unit before).  No static first unit: dense workgroups arrive as the sweep
    // workgroups end, over 15 us, and the front of the FIFO holds the most expensive tiles -- with unit = rank the heaviest
    // units waited for the workgroups that arrive last (the longest unit, 40 us, then started at 40 us).
    const unsigned n_units = (n_x + (unsigned)T - 1u) / (unsigned)T;
    const unsigned d_wgs = 0u;
    unsigned c_raw = 0;  // thread kThreads - 64: its claim
    if (tid == kThreads - 64) {
      s_hdr[1] = 0u;
      s_hdr[0] = atomicAdd(q_claim, 1u);
    }
    for (int k = tid; k < T * max_nb * 64; k += kThreads) S[k] = 0ull;
    __syncthreads();
    unsigned u_cur = (unsigned)__builtin_amdgcn_readfirstlane((int)s_hdr[0]);
    while (u_cur < n_units) {
      const unsigned long long e0 = (unsigned long long)u_cur * (unsigned)T;
      const int nt = (int)min((unsigned long long)T, (unsigned long long)n_x - e0);
      LT_TRACE_MARK(3, u_cur * 8u + (unsigned)x, 0);
      // the unit's header (tile and pair count of its tiles) in lanes < nt of EVERY wave: wait for the producers
      unsigned h_tile = 0, h_cnt = 0;
      {
        bool timed_out = false;
        if (lane < nt) {
          int spins = 0;
          for (;;) {
            const uint2 e = fifo_load(&fifo[e0 + (unsigned)lane]);
            if (e.x != kFifoEmpty) { h_tile = e.x; h_cnt = e.y; break; }
            if (++spins > kQSpinMax) { timed_out = true; break; }
            __builtin_amdgcn_s_sleep(8);
          }
        }
        if (__any(timed_out)) {
          h_tile = 0; h_cnt = 0;
          if (lane == 0) {
            s_hdr[1] = 1u;
            if (a.err_flag) atomicCAS(a.err_flag, 0, kErrXcdMap);
          }
        }
      }
      if (tid < nt) n_pairs_wg += h_cnt;
      int off[kChunkTiles + 1];
      off[0] = 0;
#pragma unroll
      for (int k = 0; k < kChunkTiles; ++k)
        off[k + 1] = off[k] + (k < nt ? min((int)(unsigned)__builtin_amdgcn_readlane((int)h_cnt, k), cap) : 0);
      auto entry_of = [&](int p, int &k) -> uint4 {
        k = 0;
#pragma unroll
        for (int m = 1; m < kChunkTiles; ++m) k += (p >= off[m]) ? 1 : 0;
        int o = off[0];
#pragma unroll
        for (int m = 1; m < kChunkTiles; ++m) o = (k >= m) ? off[m] : o;
        const unsigned tile = (unsigned)__shfl((int)h_tile, k);
        return a.sp_slots[(size_t)tile * (size_t)cap + (p - o)];
      };
      const int total = off[kChunkTiles];
      int k0e = 0;
      uint4 e0v = uint4{0u, 0u, 0u, 0u};
      const bool has0 = tid < total;
      if (total > 0) {
        int kk = 0;
        const uint4 ee = entry_of(has0 ? tid : 0, kk);
        if (has0) { e0v = ee; k0e = kk; }
      }
      CandMeta mt[2];
      long long pos[2];
      unsigned rrec[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int ti = wave + kWaves * h;
        pos[h] = -1;
        rrec[h] = 0;
        mt[h] = CandMeta{0u, 0u, 0u, 0u};
        if (ti < nt) {
          const long long p = (long long)(unsigned)__builtin_amdgcn_readlane((int)h_tile, ti & (kChunkTiles - 1)) * 64 + lane;
          if (p < C) {
            pos[h] = p;
            if (score_by == 0) rrec[h] = a.perm ? a.perm[p] : (unsigned)p;
            mt[h] = a.meta[p];
          }
        }
      }
      __syncthreads();  // the tables are clean (zeroed above, or summed and zeroed by the previous unit)
      if (s_hdr[1]) return;  // a wait timed out: the run is repeated in the two-kernel form
      LT_TRACE_MARK(3, u_cur * 8u + (unsigned)x, 1);
      int ordv[2] = {0, 0};
      long long wnb0[2] = {-1, -1};
      unsigned nb0_l[2] = {0u, 0u};
      int nn_l[2] = {0, 0};
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        if (wave + kWaves * h < nt) {
          nb0_l[h] = mt[h].nb >> 8;
          nn_l[h] = pos[h] >= 0 ? (int)(mt[h].nb & 0xFFu) : 0;
          wnb0[h] = (long long)(unsigned)__builtin_amdgcn_readfirstlane((int)nb0_l[h]);
          const int n0 = __builtin_amdgcn_readfirstlane(nn_l[h]);
          if (lane < n0) ordv[h] = a.blk_order[wnb0[h] + lane];
          if (n0 > 64) wnb0[h] = -1;
        }
      }
      const int n_it3 = total > kThreads - 64 ? (total - (kThreads - 64) + kThreads - 1) / kThreads : 0;
      auto claim = [&]() {
        if (tid == kThreads - 64) c_raw = atomicAdd(q_claim, 1u);
      };
      if (n_it3 <= 1) claim();
      auto eval = [&](const uint4 e, const int k) {
        const CRec &ci = a.cand[e.y];
        const CRec &cj = a.cand[e.x];
        const int nbs_j = (int)(e.z >> 6);  // = cj.nb_slot
        const d3 si_ = mk3(ci.s[0], ci.s[1], ci.s[2]), ei_ = mk3(ci.e[0], ci.e[1], ci.e[2]), di_ = mk3(ci.dir[0], ci.dir[1], ci.dir[2]);
        const d3 sj_ = mk3(cj.s[0], cj.s[1], cj.s[2]), ej_ = mk3(cj.e[0], cj.e[1], cj.e[2]), dj_ = mk3(cj.dir[0], cj.dir[1], cj.dir[2]);
        const Cam &camj = a.cams[(int)((unsigned)nbs_j >> 8)];
        double sc;
        if constexpr (kFast) sc = pair_score_fused(cfg, si_, ei_, di_, ci.depth[0], ci.depth[1], sj_, ej_, dj_, cj.seg, camj);
        else sc = pair_score_terms(cfg, si_, ei_, di_, ci.depth[0], ci.depth[1], sj_, ej_, dj_, cj.seg, camj);
        if (sc > 0.0)
          atomicMax(&S[(k * max_nb + (nbs_j & 0xFF)) * 64 + (int)(e.z & 63u)], (unsigned long long)__double_as_longlong(sc));
      };
      const int n_it = total > (wave << 6) ? (total - (wave << 6) + kThreads - 1) / kThreads : 0;
      for (int it = 0; it < n_it; ++it) {
        if (it > 0 && it == n_it3 - 1) claim();
        const int p = tid + it * kThreads;
        int k = k0e;
        uint4 e = e0v;
        if (it > 0) e = entry_of(p < total ? p : 0, k);
        if (p < total) eval(e, k);
      }
      for (int k = 0; k < nt; ++k) {
        if ((int)(unsigned)__builtin_amdgcn_readlane((int)h_cnt, k & (kChunkTiles - 1)) <= cap) continue;
        unsigned cc = q_word3(a.sp_slots + (size_t)(unsigned)__builtin_amdgcn_readlane((int)h_tile, k & (kChunkTiles - 1)) * (size_t)cap);
        while (cc != kNoChunk) {
          const unsigned nxt = q_word3(a.sp_pairs + (size_t)cc * kChunkCap);
          const unsigned fill = q_word3(a.sp_pairs + (size_t)cc * kChunkCap + 1);
          for (int p = tid; p < (int)fill; p += kThreads) eval(a.sp_pairs[(size_t)cc * kChunkCap + p], k);
          cc = nxt;
        }
      }
      if (tid == kThreads - 64) s_hdr[0] = d_wgs + c_raw;  // claim c of this queue is unit d_wgs + c
      __syncthreads();  // every pair of the unit is in the tables
      LT_TRACE_MARK(3, u_cur * 8u + (unsigned)x, 2);
      asm volatile("" : "+v"(ordv[0]), "+v"(ordv[1]));
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int ti = wave + kWaves * h;
        if (ti >= nt) continue;
        const bool act = pos[h] >= 0;
        const long long nb0 = (long long)nb0_l[h];
        const int n_nb = nn_l[h];
        const bool own = nb0 == wnb0[h];
        const int n_max = wave_max_i32(n_nb);
        double sum = 0.0;
        if (!__any(act && !own)) {
          for (int kk0 = 0; kk0 < n_max; kk0 += kSumChunk) {
            unsigned long long v[kSumChunk];
            int sl[kSumChunk];
#pragma unroll
            for (int u = 0; u < kSumChunk; ++u) {
              const int k = kk0 + u;
              sl[u] = __builtin_amdgcn_readlane(ordv[h], k & 63);
              v[u] = k < n_nb ? S[(ti * max_nb + sl[u]) * 64 + lane] : 0ull;
            }
#pragma unroll
            for (int u = 0; u < kSumChunk; ++u) sum += __longlong_as_double((long long)v[u]);
#pragma unroll
            for (int u = 0; u < kSumChunk; ++u)
              if (kk0 + u < n_nb) S[(ti * max_nb + sl[u]) * 64 + lane] = 0ull;
          }
        } else {
          for (int k = 0; k < n_nb; ++k) {
            unsigned long long *cell = &S[(ti * max_nb + a.blk_order[nb0 + k]) * 64 + lane];
            sum += __longlong_as_double((long long)*cell);
            *cell = 0ull;
          }
        }
        if (act) a.score[score_by == 1 ? pos[h] : (score_by == 2 ? (long long)a.spos[pos[h]] : (long long)rrec[h])] = sum;
      }
      LT_TRACE_MARK(3, u_cur * 8u + (unsigned)x, 3);
      u_cur = (unsigned)__builtin_amdgcn_readfirstlane((int)s_hdr[0]);  // (written in front of the rounds' barrier)
    }
  }
  if (tid < 64) {
    unsigned long long np = (unsigned long long)n_pairs_wg;
    for (int d = 32; d >= 1; d >>= 1) np += (unsigned long long)__shfl_xor((long long)np, d);
    if (tid == 0) {
      if (a.pair_counter && np) atomicAdd(a.pair_counter, np);
      // every workgroup of this queue on one XCD?  (looked at by the host with the run's error flag: nothing in the kernel
      // has to hold the answer)
      unsigned xcc = 0;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      xcc &= 0xFu;
      const unsigned seen = atomicCAS(&a.draw[(2 * kTileQueues + 1 + x) * 32], kFifoEmpty, xcc);
      if (seen != kFifoEmpty && seen != xcc && a.err_flag) atomicCAS(a.err_flag, 0, kErrXcdMap);
    }
  }
}

#ifdef LT_TRACE
int score_read_trace(unsigned long long *host, size_t n) {  // slices 2 and 3 of the trace array
  if (n < 4 * 4 * 65536) return -1;
  return (int)hipMemcpyFromSymbol(host + 2 * 4 * 65536, HIP_SYMBOL(g_trace), (size_t)2 * 4 * 65536 * 8,
                                  (size_t)2 * 4 * 65536 * 8, hipMemcpyDeviceToHost);
}
#endif
void launch_cand_node(hipStream_t st, long long G, const long long *tri_off, unsigned *cand_node) {
  if (G > 0)
    hipLaunchKernelGGL(k_cand_node, dim3(nblk2(G * 64, 256)), dim3(256), 0, st, G, tri_off, cand_node);
}
size_t score3_lds_bytes(int max_nb, bool f32) {
  const size_t base = (f32 ? (size_t)kWin * 48 : (size_t)9 * kWin * 8 + (size_t)kWin * 4) + 64 * 8 + kSQCap * 4;
  return ((base + (size_t)max_nb * 4 + 15) & ~(size_t)15) + (size_t)max_nb * 64 * 8;
}
size_t cand_meta_bytes() { return sizeof(CandMeta); }
// split form: tiles per unit of k_dense8 for a job's widest neighbour list (one table of maxima per tile: kDensePerCU
// workgroups per CU within kDenseLdsBudget each), bytes of an overflow chunk and of a pair entry, and the number of overflow chunks for C candidates
int score_split_t_max(int max_nb) {
  const int t = (kDenseLdsBudget) / (std::max(max_nb, 1) * 512);
  return std::max(1, std::min(kChunkTiles, t));
}
size_t score_split_chunk_bytes() { return (size_t)kChunkCap * 16; }
size_t score_split_entry_bytes() { return 16; }
long long score_split_chunks(long long C) { return C / kChunkCap + 1024; }
int score3_tile_buckets() { return kTileBuckets * kTileQueues; }  // counters (128 B apart) / lists: one per (queue, class)
// (A two-kernel form -- light sweep writing per-tile pair lists, then a dense evaluation kernel -- was
// measured: the sweep alone takes 53 us, but the evaluation does not get cheaper and the two phases no
// longer overlap across waves: 165+ us against 150 us fused.)
void launch_score3(hipStream_t st, long long C, long long G, const long long *tri_off, const unsigned *cand_node,
                   void *meta, const CRec *cand, const int *node_img, const long long *nb_off,
                   const int *blk_order, const Cam *cams, double *score, unsigned long long *pair_counter,
                   int max_nb, const ScoreCfg &cfg, double scaleinv_guard2, hipEvent_t ev_before, unsigned *draw,
                   bool f32, unsigned *perm, void *rng, bool perm_is_placement, unsigned *bucket_cnt,
                   unsigned *bucket_list, unsigned bucket_cap, const unsigned *place, unsigned *rec, const float *st_z,
                   int *err_flag, void *sp_slots, int sp_slot_cap, unsigned *sp_cnt, unsigned *sp_ovf, void *sp_pairs,
                   void *sp_desc, long long sp_chunks, hipEvent_t ev_after, const void *node_rec, unsigned *pc_cnt,
                   void *pc_list, unsigned pc_cap, int one_kernel) {
  // ev_before / ev_after: bound as the STOP events of k_cand_meta and of the stage's last kernel (hipExtLaunchKernelGGL:
  // the kernel's own completion signal carries the timestamp) -- a hipEventRecord between two kernels is a barrier packet
  // that opens a ~5.5 us gap in the stream (LT_EV_MARKERS=1: the plain records, for comparison)
  // sp_*: the pair store of the split form (sp_slots == nullptr: the fused kernel): slots of sp_slot_cap entries per tile,
  // pair counts and first overflow chunk per tile, sp_chunks overflow chunks with their (count, next) records
  // place / rec: depth-sorted sweep over STAGED records (one-pass exhaustive mode): place[natural position] = record,
  // rec (scratch, one word per candidate) receives the record of every sorted position
  if (C <= 0) return;
  // one_kernel: the split form as ONE persistent kernel (k_score_q) where it applies: single-precision sweep over the
  // placement permutation (matched mode), cost-class lists and the pair store present
  const bool dense_tables = (one_kernel & 4) != 0;  // LT_TEST_DENSE_TABLES: k_dense8 instead of k_dense_rows (exhaustive mode)
  const bool few_rows = (one_kernel & 8) != 0;      // LT_TEST_DENSE_FEW_ROWS: a table of 64 rows (units worked off in groups of tiles)
  one_kernel &= 3;
  const bool q_form = one_kernel != 0 && sp_slots != nullptr && f32 && perm != nullptr && perm_is_placement && rng == nullptr &&
                      bucket_cnt != nullptr && pc_list != nullptr;
  static int n_cu = 0;
  if (n_cu == 0) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu <= 0)
      n_cu = 256;
  }
  // C: the candidate count or an upper bound of it (the kernels read the exact count from tri_off[G])
  const long long n_tiles = (C + 63) / 64;
  static const bool ev_markers = getenv("LT_EV_MARKERS") != nullptr;
  hipExtLaunchKernelGGL(k_cand_meta, dim3((unsigned)std::min<long long>(nblk2(C, 256), 16ll * n_cu)), dim3(256), 0, st,
                        nullptr, ev_markers ? nullptr : ev_before, 0, G, cand_node, tri_off, node_img, nb_off,
                        reinterpret_cast<CandMeta *>(meta), draw, bucket_cnt, bucket_list, bucket_cap,
                        reinterpret_cast<const uint4 *>(node_rec), pc_cnt,
                        q_form ? reinterpret_cast<uint2 *>(pc_list) : nullptr, pc_cap);
  hipEvent_t ev_stop = ev_markers ? nullptr : ev_after;
  Score3Args a;
  a.G = G; a.tri_off = tri_off; a.meta = reinterpret_cast<const CandMeta *>(meta); a.cand = cand;
  a.blk_order = blk_order; a.cams = cams; a.score = score; a.pair_counter = pair_counter;
  a.draw = draw;
  a.perm = perm; a.rng = reinterpret_cast<const unsigned *>(rng);
  a.spos = nullptr;
  a.bucket_cnt = bucket_cnt; a.bucket_list = bucket_list; a.bucket_cap = bucket_cap;
  a.max_nb = max_nb;
  a.err_flag = err_flag;
  const bool split = sp_slots != nullptr && f32;
  a.sp_slots = reinterpret_cast<uint4 *>(sp_slots);
  a.sp_cnt = sp_cnt;
  a.sp_ovf = sp_ovf;
  a.sp_pairs = reinterpret_cast<uint4 *>(sp_pairs);
  a.sp_desc = reinterpret_cast<uint2 *>(sp_desc);
  a.sp_counters = draw + kTileQueues * 32;
  a.sp_chunk_cap = (unsigned)std::max<long long>(0, std::min<long long>(sp_chunks, 0x7FFFFFFFll));
  a.sp_slot_cap = sp_slot_cap;
  a.sp_t_max = score_split_t_max(max_nb);
  // pair-count lists: only with the split form over the cost-class schedule (the natural-order / fused forms do not use them)
  const bool use_pc = split && bucket_cnt != nullptr && pc_cnt != nullptr && pc_list != nullptr;
  a.pc_cnt = use_pc ? pc_cnt : nullptr;
  a.pc_list = reinterpret_cast<uint2 *>(pc_list);
  a.pc_cap = pc_cap;
  if (ev_before && ev_markers) (void)hipEventRecord(ev_before, st);
  const bool sorted = perm != nullptr && f32 && !perm_is_placement;
  if (sorted && G > 0) {  // depth order + sweep ranges per node (large nodes: exhaustive matching)
    const double guard = scaleinv_guard2 < 1e299 ? std::sqrt(scaleinv_guard2) : 1e300;
    hipLaunchKernelGGL(k_depth_order<true>, dim3((unsigned)G), dim3(64), 0, st, G, tri_off, cand, guard, perm,
                       reinterpret_cast<unsigned *>(rng), place, rec, place ? st_z : nullptr);
    hipLaunchKernelGGL(k_depth_order<false>, dim3((unsigned)G), dim3(64), 0, st, G, tri_off, cand, guard, perm,
                       reinterpret_cast<unsigned *>(rng), place, rec, place ? st_z : nullptr);
  }
  if (sorted && place) {
    a.perm = rec;
    a.spos = perm;
  }
  // persistent grid: as many single-wave workgroups as fit at once (LDS; registers allow LT_SCORE_RESIDENT per CU)
  const size_t lds = split ? (size_t)kWin * 48 + 64 * 8 + (size_t)kSQCapSplit * 8 + 64 * 4 : score3_lds_bytes(max_nb, f32);
  const long long per_cu = std::max<long long>(1, std::min<long long>(LT_SCORE_RESIDENT, (long long)(160 * 1024 / lds)));
  const dim3 grid((unsigned)std::min<long long>(n_tiles, per_cu * n_cu)), block(64);
  if (split && q_form) {
    a.pc_cnt = nullptr;  // (the FIFOs live in pc_list; no pair-count classes)
    // LDS: header | the role's area (four sweep windows + queues, or the unit's tables of maxima)
    const size_t ldsq = (size_t)kQHdrBytes + std::max((size_t)kDenseWaves * (size_t)kQSweepWaveBytes, (size_t)a.sp_t_max * (size_t)max_nb * 512);
    static int occ_q[2] = {0, 0};
    static size_t occ_q_lds[2] = {0, 0};
    const int v = cfg.fast ? 1 : 0;
    if (occ_q[v] == 0 || occ_q_lds[v] != ldsq) {
      int o = 0;
      hipError_t e = v ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&o, k_score_q<true, true>, 64 * kDenseWaves, ldsq)
                       : hipOccupancyMaxActiveBlocksPerMultiprocessor(&o, k_score_q<false, true>, 64 * kDenseWaves, ldsq);
      occ_q[v] = (e == hipSuccess && o > 0) ? o : 1;
      occ_q_lds[v] = ldsq;
    }
    // sweep workgroups first (four waves, one tile each at a time; they go on as dense workgroups when their share is swept),
    // dense-only workgroups in the slots a small job leaves: both counts multiples of 8 (queue = workgroup % 8 = XCD in
    // either role).  (Measured: 7 / 6 / 4 eighths of the slots starting in the sweep role and the rest as dense workgroups
    // from the start -- 88.5 / 89.1 / 88.9 us against 86.2 with all of them sweeping first.)
    const long long slots = std::max<long long>(8, ((long long)occ_q[v] * n_cu) & ~7ll);
    const long long n_sw = std::max<long long>(8, std::min<long long>(slots, ((n_tiles + kDenseWaves - 1) / kDenseWaves + 7) & ~7ll));
    const long long n_units_b = (n_tiles + a.sp_t_max - 1) / a.sp_t_max;
    const long long n_de = std::max<long long>(0, std::min<long long>(slots - n_sw, (n_units_b + 7) & ~7ll));
    const dim3 gq((unsigned)(n_sw + n_de));
    if (cfg.fast) hipExtLaunchKernelGGL((k_score_q<true, true>), gq, dim3(64 * kDenseWaves), ldsq, st, nullptr, ev_stop, 0, a, cfg, scaleinv_guard2, 1, (unsigned)n_sw, one_kernel == 2 ? 1 : 0);
    else hipExtLaunchKernelGGL((k_score_q<false, true>), gq, dim3(64 * kDenseWaves), ldsq, st, nullptr, ev_stop, 0, a, cfg, scaleinv_guard2, 1, (unsigned)n_sw, one_kernel == 2 ? 1 : 0);
    if (ev_after && ev_markers) (void)hipEventRecord(ev_after, st);
    return;
  }
  if (split) {
    // static schedule: exactly the workgroups that are resident at once
    static int occ_split[3] = {0, 0, 0};
    const int var = perm_is_placement ? 0 : (sorted ? 1 : 2);
    if (occ_split[var] == 0) {
      int o = 0;
      const size_t ldsw = lds * kSweepWaves;
      hipError_t e = var == 0 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&o, k_score3<true, false, true, true>, 64 * kSweepWaves, ldsw)
                   : var == 1 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&o, k_score3<true, true, false, true>, 64 * kSweepWaves, ldsw)
                              : hipOccupancyMaxActiveBlocksPerMultiprocessor(&o, k_score3<true, false, false, true>, 64 * kSweepWaves, ldsw);
      occ_split[var] = (e == hipSuccess && o > 0) ? o : 8;
    }
    a.sp_wave_lds = (int)lds;
    const dim3 grid1((unsigned)std::max<long long>(1, std::min<long long>((n_tiles + kSweepWaves - 1) / kSweepWaves, (long long)occ_split[var] * n_cu)));
    const dim3 block1(64 * kSweepWaves);
    const size_t lds1 = lds * kSweepWaves;
    if (perm_is_placement) hipLaunchKernelGGL((k_score3<true, false, true, true>), grid1, block1, lds1, st, a, cfg, scaleinv_guard2);
    else if (sorted) hipLaunchKernelGGL((k_score3<true, true, false, true>), grid1, block1, lds1, st, a, cfg, scaleinv_guard2);
    else hipLaunchKernelGGL((k_score3<true, false, false, true>), grid1, block1, lds1, st, a, cfg, scaleinv_guard2);
    // k_dense8: kDensePerCU workgroups of four waves per CU (registers: four waves per SIMD; LDS: one 512 B x max_nb table per tile)
    const size_t lds2 = (size_t)kDenseHdrBytes + (size_t)a.sp_t_max * (size_t)max_nb * 512;
    const long long fit = std::max<long long>(1, std::min<long long>(kDensePerCU, (long long)(160 * 1024 / lds2)));
    const int by = perm_is_placement ? 1 : (sorted && a.spos ? 2 : 0);
    // depth-sorted tiles without class lists (the exhaustive mode): units of kChunkTiles tiles over row-compacted tables
    const int rows_cap = few_rows ? 64 : kDenseLdsBudget / (std::max(max_nb, 1) * 8);
    if (sorted && !use_pc && bucket_cnt == nullptr && rows_cap >= 64 && !dense_tables) {
      const size_t lds3 = (size_t)((kRowsHdrBytes + 15) & ~15) + (size_t)rows_cap * (size_t)max_nb * 8;
      const long long fit3 = std::max<long long>(1, std::min<long long>(kDensePerCU, (long long)(160 * 1024 / lds3)));
      const dim3 g3((unsigned)std::max<long long>(8, (fit3 * n_cu) & ~7ll));
      if (cfg.fast) hipExtLaunchKernelGGL(k_dense_rows<true>, g3, dim3(64 * kDenseWaves), lds3, st, nullptr, ev_stop, 0, a, cfg, by, rows_cap);
      else hipExtLaunchKernelGGL(k_dense_rows<false>, g3, dim3(64 * kDenseWaves), lds3, st, nullptr, ev_stop, 0, a, cfg, by, rows_cap);
      if (ev_after && ev_markers) (void)hipEventRecord(ev_after, st);
      return;
    }
    const dim3 g2((unsigned)std::max<long long>(8, (fit * n_cu) & ~7ll));  // a multiple of 8: see the unit queues
    if (cfg.fast) hipExtLaunchKernelGGL(k_dense8<true>, g2, dim3(64 * kDenseWaves), lds2, st, nullptr, ev_stop, 0, a, cfg, by);
    else hipExtLaunchKernelGGL(k_dense8<false>, g2, dim3(64 * kDenseWaves), lds2, st, nullptr, ev_stop, 0, a, cfg, by);
    if (ev_after && ev_markers) (void)hipEventRecord(ev_after, st);
    return;
  }
  if (perm_is_placement) {
    if (f32) hipExtLaunchKernelGGL((k_score3<true, false, true, false>), grid, block, lds, st, nullptr, ev_stop, 0, a, cfg, scaleinv_guard2);
    else hipExtLaunchKernelGGL((k_score3<false, false, true, false>), grid, block, lds, st, nullptr, ev_stop, 0, a, cfg, scaleinv_guard2);
  } else if (sorted) hipExtLaunchKernelGGL((k_score3<true, true, false, false>), grid, block, lds, st, nullptr, ev_stop, 0, a, cfg, scaleinv_guard2);
  else if (f32) hipExtLaunchKernelGGL((k_score3<true, false, false, false>), grid, block, lds, st, nullptr, ev_stop, 0, a, cfg, scaleinv_guard2);
  else hipExtLaunchKernelGGL((k_score3<false, false, false, false>), grid, block, lds, st, nullptr, ev_stop, 0, a, cfg, scaleinv_guard2);
  if (ev_after && ev_markers) (void)hipEventRecord(ev_after, st);
}

}  // namespace lt
