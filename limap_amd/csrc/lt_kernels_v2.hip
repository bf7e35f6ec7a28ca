// lt_kernels_v2.hip -- second-generation kernels of the matched-mode pipeline and of the scoring
// stage (see DESIGN.md section 3 for the map; lt_kernels.hip keeps the invariant builders, the
// generic radix-sort grouping, the exhaustive generation and the selection kernels).
//
//   k_rows_starts + k_rows_transpose (upload) -> k_gates_ln + k_tri_rounds -> k_node_prefix + k_place_rounds
//       HOT LOOP 1 in the LINE-SLOT form (round 5; the default for what limap's matchers write: sorted blocks with
//       contiguous lines and at most 32 rows per line).  k_gates_ln: one lane per line of the image, the view-1 side of
//       the gates once per lane, rows as 16-bit neighbour lines in (rank, line) order, the neighbour's gate table in
//       LDS by DMA -> the block's dense survivor list.  k_tri_rounds: rounds of 64 survivors, persistent workgroups ->
//       valid candidates per round.  k_place_rounds: the permutation into the reference's candidate order.
//   k_gates + k_tri_rows
//       HOT LOOP 1 in the ROW-SLOT form (blocks in any other shape; extra proposals).  k_gates: coalesced match rows, the
//       neighbour's gate table staged in LDS, cheap three-way gates on every row -> ordered survivor lists.  k_tri_rows:
//       triangulation etc. for the survivors on dense wave64s -> valid candidates appended in row
//       order to per-slot lists.
//   k_node_prefix + k_place (rows of every block sorted by line id -- what limap's matchers write)
//       Sort-free placement into the reference's candidate order (neighbour-ascending, then
//       match-row order -- base_line_triangulator.cc:71-103): per-(block, line) counts -> per-node
//       prefix over the neighbour blocks -> final position of every candidate.
//   k_pack_keys + radix sort + k_permute (generic rows): stable sort of the candidates by node.
// The scoring stage (k_cand_meta, k_score3, ...) lives in lt_kernels_score.hip.
// Compiled with -ffp-contract=off (see lt_geom.h).

#include "lt_devfn.h"

#include <algorithm>

namespace lt {

// ---------------------------------------------------------------------------------------------
// HOT LOOP 1, row order (triangulateOneNode, base_line_triangulator.cc:161-337)
// ---------------------------------------------------------------------------------------------
// The rows of neighbour block b = (image, neighbour) are cut into SLOTS of kRowsPerWave rows; one
// wave owns one slot in every kernel of this stage (slot s of block b: rows m_off[b] + s * kRowsPerWave
// ..., list index lin = b * n_slots + s).  grid.y = block, so the image pair, F, the baseline and
// both cameras are wave-uniform (scalar registers).
//   k_gates     stage A on every row with the three-way cheap gates (gate3): lean, high occupancy.
//               The neighbour's gate table (SegGate, 80 B per segment) is staged once per workgroup
//               in LDS.  Per slot an ordered list of surviving rows:
//               st_row[r0 + k] = (line | undecided << 31, neighbour line).
//   k_tri_rows  stage B (triangulation / cheirality / sensitivity / uncertainty / ranges) from those
//               lists on dense wave64s; rows flagged undecided first go through the exact gates
//               (gen_gates).  Valid candidates are appended IN ROW ORDER to the slot's list
//               st_*[r0 ...]; wave_count[] holds the list lengths.  In the fast path the number of
//               valid candidates per (block, line) run is counted in cnt_bl for the placement pass.
// Tuning knobs (compile-time; Makefile EXTRA=-D...)
#ifndef LT_GEN_CHUNKS
#define LT_GEN_CHUNKS 5
#endif
#ifndef LT_GATE_WAVES
#define LT_GATE_WAVES 8
#endif
#ifndef LT_GATE_WAVES_PER_EU
#define LT_GATE_WAVES_PER_EU 4  // two 8-wave workgroups per CU: the register budget is 128
#endif
#define LT_GATE_OCC __attribute__((amdgpu_waves_per_eu(LT_GATE_WAVES_PER_EU, LT_GATE_WAVES_PER_EU)))
constexpr int kGenChunks = LT_GEN_CHUNKS;  // 64-row chunks per slot
constexpr int kRowsPerWave = 64 * kGenChunks;
constexpr int kGateWaves = LT_GATE_WAVES;  // waves (= slots) per k_gates workgroup
// Stage B and the placement work on GROUPS of kTriSlots consecutive slots (one wave per group): the
// survivors of one slot (~10 % of its rows) would fill only a fraction of a wave64.
#ifndef LT_TRI_SLOTS
#define LT_TRI_SLOTS 4
#endif
constexpr int kTriSlots = LT_TRI_SLOTS;
#ifndef LT_TRI_WAVES
#define LT_TRI_WAVES 4
#endif
constexpr int kTriWaves = LT_TRI_WAVES;  // waves (= groups of one block) per k_tri_rows workgroup; they share nothing
                                         // (measured: 1 -> 68.5 us, 2 -> 65.0, 4 -> 60.6)
constexpr int kTriRows = kTriSlots * kRowsPerWave;
static_assert(kGateWaves % kTriSlots == 0, "a group must not straddle k_gates workgroups' slot ranges");

#ifdef LT_TRACE
// developer build: per-wave timestamps (100 MHz wall clock), read back by lt_debug_read_trace
__device__ unsigned long long g_trace[4 * 4 * 65536];
#define LT_TRACE_MARK(kern, id, slot) \
  if (lane_id() == 0 && (id) < 65536u) g_trace[(kern) * 4 * 65536 + 4 * (id) + (slot)] = wall_clock64()
#else
#define LT_TRACE_MARK(kern, id, slot)
#endif

struct GenArgs {
  const long long *m_off;
  const int *m_pairs;
  const int *blk_img, *blk_nb, *blk_slot;
  const long long *seg_off;
  const Cam *cams;
  const Seg *segs;
  const SegGate *gates;
  const PairRec *pairs;
  const long long *blk_line_base;
  uint2 *st_row;          // [P] surviving rows of stage A, per-slot lists at the slot's first row:
                          // (line | undecided << 31, neighbour line)
  unsigned *surv_count;   // [n_blk * n_slots]
  CRec *st_r;             // [P] valid candidates, per-slot lists at the slot's first row
  double *st_unc;         // their uncertainties
  unsigned *st_key;       // node id of every staged candidate
  unsigned *wave_count;   // [n_blk * n_slots]
  unsigned *cnt_bl;       // valid candidates per (block, line) or nullptr (generic path)
  const struct BlkRec *blk;  // [n_blk]
  int n_blk;
  int n_slots;            // slots per block (multiple of kGateWaves)
  int lds_segs;           // capacity (in segments) of the LDS table T2; 0 = read the gate records from HBM/L2
  int lds_segs1;          // capacity of T1 (the image's own segments); 0 = gather them from HBM/L2
  // VP-guided proposals (use_vp): per segment its vanishing point (image homogeneous coordinates) and a flag
  const double *seg_vp;          // [G][3] or nullptr
  const unsigned char *seg_has_vp;
  // point-guided proposals (SetBipartites2d): per segment its neighbouring points (CSR), SfM points or nullptr
  const long long *seg_pt_off;   // [G + 1] or nullptr
  const SegPoint *seg_pts;
  const double *sfm_xyz;         // [n_sfm][3] or nullptr (points are then triangulated from the two views)
  int *err_flag;
  // Extra proposals (kExtra): a connection yields a VARIABLE number of candidates (many-points, one per shared point,
  // vp(l1), vp(l2), algebraic -- the reference sets no limit, base_line_triangulator.cc:238-248), so stage B runs twice:
  // count_only = 1 counts the candidates of every (block, group) list into wave_count without storing anything, the host
  // scans the counts, and the second run writes the lists back to back at group_base[list] -- exact staging, no cap.
  // Without extras a row yields at most one candidate: group_base == nullptr, a group's list starts at its first row.
  const long long *group_base;
  int count_only;
  int many_on, one_on;    // which point-guided proposals run (seg_pts != null)
  // Line-slot form (round 5, k_gates_ln): slot s of a block = its lines [64 s, 64 s + 64), one lane per line.
  const unsigned short *tr;   // neighbour line of every row, per slot in (rank within the run, line) order: the rows
                              // the lanes of a wave read in iteration j are adjacent (k_rows_transpose)
  const unsigned *run_len;    // [(block, line)] rows of the line in the block
  const unsigned *slot_row0;  // [n_blk * n_slots] first row of the slot (global row index); nullptr: the row-slot
                              // form, slot s of block b starts at row rb + s * kRowsPerWave
};
// first row of slot s of block b (rb: the block's first row) in either slot form
static __device__ __forceinline__ long long slot_first_row(const unsigned *__restrict__ slot_row0, int n_slots, int b, int s,
                                                           long long rb) {
  return slot_row0 ? (long long)slot_row0[(size_t)b * n_slots + s] : rb + (long long)s * kRowsPerWave;
}


// Per-block record: everything the row kernels need to know about neighbour block b in ONE scalar
// load (instead of the chain m_off[b] / blk_img[b] / blk_nb[b] -> seg_off[...]).
struct BlkRec {
  long long rb, re;   // rows of the block
  long long g1, g2;   // first segment (= node id) of the image / of the neighbour
  long long lbase;    // first (block, line) counter
  int M2;             // segments of the neighbour
  int i1, i2, nbslot; // image, neighbour, position of the neighbour in the image's list
  int M1;             // segments of the image
  int pad_;
};
static_assert(sizeof(BlkRec) == 64, "BlkRec layout");

__global__ void k_build_blk(int n_blk, const long long *__restrict__ m_off, const int *__restrict__ blk_img,
                            const int *__restrict__ blk_nb, const int *__restrict__ blk_slot,
                            const long long *__restrict__ seg_off, const long long *__restrict__ blk_line_base,
                            BlkRec *__restrict__ out) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= n_blk) return;
  BlkRec r;
  r.rb = m_off[b]; r.re = m_off[b + 1];
  r.i1 = blk_img[b]; r.i2 = blk_nb[b]; r.nbslot = blk_slot[b];
  r.g1 = seg_off[r.i1]; r.g2 = seg_off[r.i2];
  r.M2 = (int)(seg_off[r.i2 + 1] - r.g2);
  r.lbase = blk_line_base[b];
  r.M1 = (int)(seg_off[r.i1 + 1] - r.g1);
  r.pad_ = 0;
  out[b] = r;
}

// The staged match rows arrive in the compressed block format of lt_rows.h (17 bits per row; blocks that cannot take it:
// plain words in the overflow array).  One wave per block rebuilds the plain row words line | neighbour line << 16 that
// k_gates reads, at the block's rows in DEVICE block order: line = first line + number of "new line" bits up to the row.
constexpr int kExpandWaves = 4;
__global__ void __launch_bounds__(64 * kExpandWaves)
k_expand_rows(int n_blk, const RowDesc *__restrict__ desc, const unsigned *__restrict__ stream,
              const unsigned *__restrict__ ovf, unsigned *__restrict__ rows) {
  const int b = blockIdx.x;
  if (b >= n_blk) return;
  const RowDesc d = desc[b];
  const int lane = lane_id();
  const int wave = threadIdx.x >> 6;
  unsigned *out = rows + d.row_off;
  if (d.ooff >= 0) {  // plain form
    for (int r = threadIdx.x; r < d.n; r += 64 * kExpandWaves) out[r] = ovf[d.ooff + r];
    return;
  }
  const long long nbw = ((((long long)d.n + 1) / 2) + 1) & ~1ll;  // lt_rows.h: cb_nb_words
  const unsigned short *nb = reinterpret_cast<const unsigned short *>(stream + d.coff);
  const unsigned long long *bits = reinterpret_cast<const unsigned long long *>(stream + d.coff + nbw);
  const int n_words = (d.n + 63) >> 6;
  unsigned base = (unsigned)d.line0;
  // 64 bit words (4096 rows) at a time: lane l holds word l and, after a scan over the wave, the line number at the
  // word's first row; every wave does that (a few instructions) and then takes every kExpandWaves-th word, so that the
  // row loads of different words are independent of each other (the serial form was bound by one load latency per word)
  for (int w0 = 0; w0 < n_words; w0 += 64) {
    const unsigned long long mine = (w0 + lane < n_words) ? bits[w0 + lane] : 0ull;
    const unsigned cnt = (unsigned)__popcll(mine);
    unsigned incl = cnt;
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) {
      const unsigned o = __shfl_up(incl, s, 64);
      if (lane >= s) incl += o;
    }
    const unsigned first = base + incl - cnt;
    const int k_end = min(64, n_words - w0);
#pragma unroll 4
    for (int k = wave; k < k_end; k += kExpandWaves) {
      const unsigned long long w = __shfl(mine, k, 64);
      const unsigned lb = __shfl(first, k, 64);
      const int r = ((w0 + k) << 6) + lane;
      const unsigned line = lb + (unsigned)__popcll(w & ((2ull << lane) - 1ull));
      if (r < d.n) out[r] = (line & 0xFFFFu) | ((unsigned)nb[r] << 16);
    }
    base += __shfl(incl, 63, 64);
  }
}

// Persistent workgroups: workgroup w takes a contiguous range of items, where item = (block, part) and
// a part is kGateWaves slots.  Both operand tables of an item live in LDS: T1 = the first 80 bytes
// (endpoints, start / end rays) of the image's own Seg records, reloaded only when the image changes
// (blocks of one image are consecutive), and T2 = the neighbour's SegGate records.  While an item is
// being processed the T2 units of the NEXT item are already in flight (held in registers until the
// table is free), so the chunk loop sees no global latency beyond the (prefetched) match rows.
// kLds1 / kLds2: compile-time choice of the operand source (LDS table or HBM/L2 gather) -- a run-time
// choice would merge the two pointers and turn every operand read into a FLAT load.
// blk_r / pairs_r: the same arrays as a.blk / a.pairs, as __restrict__ kernel parameters -- only then are
// the wave-uniform record loads inside the persistent loop scalar (s_load); through the struct they may
// alias the kernel's stores and are issued per lane.
template <bool kLds1, bool kLds2>
__global__ void __launch_bounds__(64 * kGateWaves) LT_GATE_OCC
k_gates(GenArgs a, GenCfg cfg, const BlkRec *__restrict__ blk_r, const PairRec *__restrict__ pairs_r) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  double2 *T2 = reinterpret_cast<double2 *>(smem_raw);   // [lds_segs][5] : SegGate records of the neighbour
  double2 *T1 = T2 + (size_t)a.lds_segs * 5;             // [lds_segs1][5]: own segments (x1 y1 x2 y2 rs re)
  const int wave = threadIdx.x >> 6;
  const int lane = lane_id();
  const int n_parts = a.n_slots / kGateWaves;
  const int n_items = a.n_blk * n_parts;
  constexpr int kRowsPerPart = kGateWaves * kRowsPerWave;
  constexpr int kTab = 5;  // 16-byte units of T2 per thread held in flight (5 * 512 threads * 16 B = 40 KB)
  constexpr int nth = 64 * kGateWaves;
  const int per_wg = (n_items + (int)gridDim.x - 1) / (int)gridDim.x;
  const int item_end = min(n_items, ((int)blockIdx.x + 1) * per_wg);

  // next item: record fields (wave-uniform) and table units (in flight)
  long long n_rb = 0, n_re = 0, n_g1 = 0, n_g2 = 0;
  int n_M2 = 0, n_i1 = -1;
  bool n_live = false;
  int n_blk2 = -1, fetched_b2 = -1, staged_b2 = -1;  // block of the next item / whose table is in flight / in T2
  double2 tv0, tv1, tv2, tv3, tv4;
  tv0 = tv1 = tv2 = tv3 = tv4 = double2{0.0, 0.0};
#define LT_GATES_FETCH(IT)                                                                          \
  {                                                                                                 \
    const int it_ = (IT);                                                                           \
    const int itc_ = it_ < item_end ? it_ : item_end - 1;                                           \
    const int bb_ = __builtin_amdgcn_readfirstlane(itc_ / n_parts), pp_ = itc_ - bb_ * n_parts;     \
    const BlkRec *rp_ = blk_r + bb_; /* wave-uniform index: scalar loads */                         \
    n_rb = rp_->rb; n_re = rp_->re; n_g1 = rp_->g1; n_g2 = rp_->g2; n_M2 = rp_->M2; n_i1 = rp_->i1; \
    n_live = it_ < item_end && n_rb + (long long)pp_ * kRowsPerPart < n_re;                         \
    n_blk2 = bb_;                                                                                   \
    if (kLds2 && n_live && bb_ != fetched_b2) { /* the parts of a block share the neighbour's table */ \
      fetched_b2 = bb_;                                                                             \
      const double2 *src_ = reinterpret_cast<const double2 *>(a.gates + n_g2);                      \
      const int units_ = n_M2 * 5;                                                                  \
      if ((int)threadIdx.x + 0 * nth < units_) tv0 = src_[threadIdx.x + 0 * nth];                   \
      if ((int)threadIdx.x + 1 * nth < units_) tv1 = src_[threadIdx.x + 1 * nth];                   \
      if ((int)threadIdx.x + 2 * nth < units_) tv2 = src_[threadIdx.x + 2 * nth];                   \
      if ((int)threadIdx.x + 3 * nth < units_) tv3 = src_[threadIdx.x + 3 * nth];                   \
      if ((int)threadIdx.x + 4 * nth < units_) tv4 = src_[threadIdx.x + 4 * nth];                   \
    }                                                                                               \
  }
  int item = (int)blockIdx.x * per_wg;
  if (item >= item_end) return;
  int cur_i1 = -1;       // image whose segments are in T1
  LT_GATES_FETCH(item);
  for (; item < item_end; ++item) {
    const long long rb = n_rb, re = n_re, g1 = n_g1, g2 = n_g2;
    const int M2 = n_M2, i1 = n_i1;
    const bool live = n_live;
    const int blk2 = n_blk2;
    // b is wave-uniform; telling the compiler so keeps the pair record (F) in scalar registers --
    // otherwise it is re-fetched with per-lane vector loads inside the chunk loop
    const int b = __builtin_amdgcn_readfirstlane(item / n_parts), part = item - b * n_parts;
    const int slot = part * kGateWaves + wave;
    const unsigned lin = (unsigned)b * (unsigned)a.n_slots + (unsigned)slot;
    LT_TRACE_MARK(0, lin, 0);
    if (live) {
      const bool new_img = kLds1 && i1 != cur_i1;
      const bool new_t2 = kLds2 && blk2 != staged_b2;
      if (new_t2 || new_img) __syncthreads();  // the previous item's readers are done with the tables
      if (new_t2) {
        staged_b2 = blk2;
        const int units = M2 * 5;
        if ((int)threadIdx.x + 0 * nth < units) T2[threadIdx.x + 0 * nth] = tv0;
        if ((int)threadIdx.x + 1 * nth < units) T2[threadIdx.x + 1 * nth] = tv1;
        if ((int)threadIdx.x + 2 * nth < units) T2[threadIdx.x + 2 * nth] = tv2;
        if ((int)threadIdx.x + 3 * nth < units) T2[threadIdx.x + 3 * nth] = tv3;
        if ((int)threadIdx.x + 4 * nth < units) T2[threadIdx.x + 4 * nth] = tv4;
        for (int u = threadIdx.x + kTab * nth; u < units; u += nth)  // tables beyond one pass (rare)
          T2[u] = reinterpret_cast<const double2 *>(a.gates + g2)[u];
      }
      if (new_img) {
        cur_i1 = i1;
        const int M1 = (int)(a.seg_off[i1 + 1] - g1);
        for (int u = threadIdx.x; u < M1 * 5; u += nth) {
          const int sidx = u / 5, k = u - sidx * 5;
          T1[u] = reinterpret_cast<const double2 *>(a.segs + g1 + sidx)[k];
        }
      }
      if (new_t2 || new_img) __syncthreads();
    }
    LT_GATES_FETCH(item + 1);
    LT_TRACE_MARK(0, lin, 1);
    const long long r0 = rb + (long long)slot * kRowsPerWave;
    unsigned wcount = 0;
    if (live && r0 < re) {
      double F[9];
#pragma unroll
      for (int k = 0; k < 9; ++k) F[k] = pairs_r[b].F[k];
      unsigned pass_bits = 0, und_bits = 0;
      // a match row is one packed word: line | neighbour line << 16 (lt_api.cpp packs while it validates; both ids are
      // below 65535 -- util/types.h:16 -- so 0xFFFFFFFF marks "no row")
      constexpr unsigned kNoRow = 0xFFFFFFFFu;
      unsigned row_n = kNoRow;
      {
        long long r = r0 + lane;
        if (r < re) row_n = (unsigned)a.m_pairs[r];
      }
      for (int c = 0; c < kGenChunks; ++c) {
        const unsigned row = row_n;
        row_n = kNoRow;
        {  // next chunk's rows
          long long r = r0 + 64ll * (c + 1) + lane;
          if (c + 1 < kGenChunks && r < re) row_n = (unsigned)a.m_pairs[r];
        }
        int res = 0;
        if (row != kNoRow) {
          // separate LDS / global code paths: a selected pointer would turn these into FLAT loads
          double2 e0, e1, e2, e3, e4, h0, h1, h2, h3, h4;
          if (kLds1) {
            // table offset = (low half of the row) x 80 bytes in ONE instruction (16-bit multiply-add with operand
            // select): unpacking the halves first cost two more VALU instructions per row in an issue-bound kernel
            unsigned off1;
            asm("v_mad_u32_u16 %0, %1, %2, 0" : "=v"(off1) : "v"(row), "s"(80u));
            const double2 *p1 = reinterpret_cast<const double2 *>(reinterpret_cast<const char *>(T1) + off1);
            e0 = p1[0]; e1 = p1[1]; e2 = p1[2]; e3 = p1[3]; e4 = p1[4];
          } else {
            const double2 *p1 = reinterpret_cast<const double2 *>(a.segs + g1 + (int)(row & 0xFFFFu));
            e0 = p1[0]; e1 = p1[1]; e2 = p1[2]; e3 = p1[3]; e4 = p1[4];
          }
          if (kLds2) {
            unsigned off2;  // (high half of the row) x 80
            asm("v_mad_u32_u16 %0, %1, %2, 0 op_sel:[1,0,0,0]" : "=v"(off2) : "v"(row), "s"(80u));
            const double2 *p2 = reinterpret_cast<const double2 *>(reinterpret_cast<const char *>(T2) + off2);
            h0 = p2[0]; h1 = p2[1]; h2 = p2[2]; h3 = p2[3]; h4 = p2[4];  // (a conflict-free access pattern was timed: no faster)
          } else {
            const int ng = (int)(row >> 16);
            const double2 *p2 = reinterpret_cast<const double2 *>(a.gates + g2 + ng);
            h0 = p2[0]; h1 = p2[1]; h2 = p2[2]; h3 = p2[3]; h4 = p2[4];
          }
          res = gate3(cfg, e0.x, e0.y, e1.x, e1.y, e2.x, e2.y, e3.x, e3.y, e4.x, e4.y,  // l1: endpoints, rs, re
                      h0.x, h0.y, h1.x, h1.y, h2.x, h2.y, h3.x, h3.y, h4.x, h4.y, F);   // l2: SegGate fields
        }
        // the outcome of chunk c is two bits per lane; the survivor list is written after the loop, so
        // that no global store (and no wait for one) sits between the chunks
        pass_bits |= (res != 0 ? 1u : 0u) << c;
        und_bits |= (res == 2 ? 1u : 0u) << c;
        if (r0 + 64ll * (c + 1) >= re) break;
      }
      static_assert(kGenChunks <= 32, "one outcome bit per chunk and lane");
#pragma unroll
      for (int c = 0; c < kGenChunks; ++c) {
        const bool pass = (pass_bits >> c) & 1u;
        const unsigned long long m = __ballot(pass);
        if (pass) {
          // the survivor entry carries the row itself (line | undecided << 31, neighbour line): stage B then
          // reads its rows as one contiguous list instead of one scattered 64-byte sector per survivor
          // (the chunk's rows are re-read here, coalesced and cache-hot, rather than held in registers)
          const unsigned v = (unsigned)a.m_pairs[r0 + 64ll * c + lane];
          a.st_row[r0 + wcount + __popcll(m & lanemask_lt())] =
              make_uint2((v & 0xFFFFu) | (((und_bits >> c) & 1u) ? 0x80000000u : 0u), v >> 16);
        }
        wcount += (unsigned)__popcll(m);
      }
    }
    LT_TRACE_MARK(0, lin, 2);
    if (lane == 0) a.surv_count[lin] = wcount;
  }
#undef LT_GATES_FETCH
}

// ---------------------------------------------------------------------------------------------
// Line-slot form of stage A (round 5).  What limap's matchers write -- every (image, neighbour) block sorted by line id,
// the lines of the block a contiguous range, a handful of rows (top-k) per line: exactly the blocks that take the
// compressed form of lt_rows.h -- lets ONE LANE OWN ONE LINE of the image:
//   * everything that depends on (l1, image pair) only -- the segment's endpoints and rays, the epipolar lines of its two
//     endpoints (gate3_epi: 2 x 15 of gate3's 178 lane instructions), its squared length -- is computed once per lane and
//     item instead of once per row, and the table T1 of the image's own segments (40 KB of LDS, a barrier pair per image)
//     disappears: the lane reads its own 80 bytes from the Seg array once per item;
//   * with T1 gone the neighbour's table T2 fits twice: the NEXT block's table is copied global -> LDS by the DMA path
//     (global_load_lds_dwordx4, no staging registers) into the other buffer while this block's rows are gated -- one
//     barrier per block instead of two around a register-staged copy that nothing overlapped;
//   * a row is 16 bits (the neighbour line; the line is the lane): k_rows_transpose stores the rows of a slot in
//     (rank within the run, line) order, so the rows a wave needs in iteration j are adjacent -- 128 bytes per load.
// Slots are line ranges here (slot s = lines [64 s, 64 s + 64) of the image); a slot's survivors, ordered lane-major = row
// order, are appended to the block's dense survivor list (stage B: k_tri_rounds).
// Requirements (lt_upload decides; otherwise the row-slot form k_gates runs): every block compressed, no run longer than
// kMaxRun rows (one outcome bit per row and lane), the neighbour tables within the LDS.
// ---------------------------------------------------------------------------------------------
constexpr int kMaxRun = 32;

// Pass 1 over a compressed block: first row of every run (= line) of the block -> rstart[(block, line)], number of runs.
__global__ void __launch_bounds__(256)
k_rows_starts(int n_blk, const RowDesc *__restrict__ desc, const unsigned *__restrict__ stream,
              const long long *__restrict__ blk_line_base, unsigned *__restrict__ rstart, int *__restrict__ blk_nruns,
              int *__restrict__ ln_flag) {
  const int b = blockIdx.x;
  if (b >= n_blk) return;
  const RowDesc d = desc[b];
  const int lane = lane_id();
  const int wave = threadIdx.x >> 6;
  if (d.ooff >= 0) {  // plain form: not a line-slot job
    if (threadIdx.x == 0) {
      atomicOr(ln_flag, 1);
      blk_nruns[b] = 0;
    }
    return;
  }
  const long long nbw = ((((long long)d.n + 1) / 2) + 1) & ~1ll;  // lt_rows.h: cb_nb_words
  const unsigned long long *bits = reinterpret_cast<const unsigned long long *>(stream + d.coff + nbw);
  const int n_words = (d.n + 63) >> 6;
  unsigned *out = rstart + blk_line_base[b] + d.line0;
  unsigned base = 0;  // runs that start before the current batch of words (row 0 carries no bit: run 0 starts there)
  for (int w0 = 0; w0 < n_words; w0 += 64) {
    const unsigned long long mine = (w0 + lane < n_words) ? bits[w0 + lane] : 0ull;
    const unsigned cnt = (unsigned)__popcll(mine);
    unsigned incl = cnt;
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) {
      const unsigned o = __shfl_up(incl, s, 64);
      if (lane >= s) incl += o;
    }
    const unsigned first = base + incl - cnt;
    const int k_end = min(64, n_words - w0);
    for (int k = wave; k < k_end; k += 4) {
      const unsigned long long w = __shfl(mine, k, 64);
      const unsigned lb = __shfl(first, k, 64);
      const int r = ((w0 + k) << 6) + lane;
      const bool head = r < d.n && (r == 0 || ((w >> lane) & 1ull));
      if (head) out[lb + (unsigned)__popcll(w & ((2ull << lane) - 1ull))] = (unsigned)r;
    }
    base += __shfl(incl, 63, 64);
  }
  if (threadIdx.x == 0) blk_nruns[b] = d.n > 0 ? (int)base + 1 : 0;
}

// Pass 2: one wave per (block, line slot).  Run lengths of the slot's lines, the slot's first row, and the neighbour
// lines of its rows in (rank within the run, line) order.
__global__ void __launch_bounds__(256)
k_rows_transpose(int n_blk, int n_slots, const RowDesc *__restrict__ desc, const unsigned *__restrict__ stream,
                 const long long *__restrict__ blk_line_base, const unsigned *__restrict__ rstart,
                 const int *__restrict__ blk_nruns, unsigned *__restrict__ run_len, unsigned *__restrict__ slot_row0,
                 unsigned short *__restrict__ tr, int *__restrict__ ln_flag) {
  const int b = blockIdx.y;
  const int s = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= n_blk || s >= n_slots) return;
  const RowDesc d = desc[b];
  const int lane = lane_id();
  const size_t lin = (size_t)b * n_slots + s;
  if (d.ooff >= 0) {
    if (lane == 0) slot_row0[lin] = (unsigned)d.row_off;
    return;
  }
  const long long lbase = blk_line_base[b];
  const int M1 = (int)(blk_line_base[b + 1] - lbase);
  const int nr = blk_nruns[b];
  const int line = 64 * s + lane;
  const int rel = line - d.line0;
  const bool in = rel >= 0 && rel < nr;
  const unsigned start = in ? rstart[lbase + line] : 0u;
  const unsigned end = in ? (rel == nr - 1 ? (unsigned)d.n : rstart[lbase + line + 1]) : 0u;
  const unsigned len = end - start;
  if (line < M1) run_len[lbase + line] = len;
  // the lines of a compressed block are a contiguous range and every one of them has rows: the slot's first row is the
  // start of its first line inside the range (0 in front of the range, n behind it)
  const int rel0 = 64 * s - d.line0;
  const unsigned row0 = rel0 <= 0 ? 0u : (rel0 < nr ? rstart[lbase + 64 * s] : (unsigned)d.n);
  if (lane == 0) slot_row0[lin] = (unsigned)d.row_off + row0;
  const unsigned maxlen = wave_max_u32(len);
  if (maxlen > (unsigned)kMaxRun && lane == 0) atomicOr(ln_flag, 2);
  const unsigned short *nb = reinterpret_cast<const unsigned short *>(stream + d.coff);
  unsigned short *out = tr + d.row_off;
  unsigned base = row0;
  for (unsigned j = 0; j < maxlen; ++j) {
    const bool act = j < len;
    const unsigned long long m = __ballot(act);
    if (act) out[base + (unsigned)__popcll(m & lanemask_lt())] = nb[start + j];
    base += (unsigned)__popcll(m);
  }
}

// s_setprio takes an immediate
static __device__ __forceinline__ void set_prio(int p) {  // p: wave-uniform, 0..3
  if (p >= 3) __builtin_amdgcn_s_setprio(3);
  else if (p == 2) __builtin_amdgcn_s_setprio(2);
  else if (p == 1) __builtin_amdgcn_s_setprio(1);
  else __builtin_amdgcn_s_setprio(0);
}
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_void_t;

// kW waves per workgroup, one table buffer per workgroup.  Workgroups of four waves while four tables fit a CU (tables up
// to 40 KB: 512 neighbour segments), of eight beyond (two tables of up to 80 KB): 16 waves per CU either way, and as many
// INDEPENDENT workgroups as the LDS allows -- the phases of an item in which a wave issues little (requests, survivor
// list, barrier) then idle one wave in four on a SIMD instead of two (measured at 100 x 500: eight waves with two table
// buffers -- the next block's table landing while this one is gated -- 65.7 us, eight waves with one buffer 65.7, four
// waves with one buffer 63.6: what the second buffer hides is hidden by the other workgroups anyway).
// Latency.  An item costs a wave ~2 us of arithmetic (8 us at four waves per SIMD), a global load 1-2 us, and the barrier
// of a new block puts the waves of a workgroup in the same phase.  So no load is waited for where it is issued:
//   * rows are fetched a GROUP of kRowAhead iterations ahead;
//   * the per-lane inputs of item k + 1 (first row group, the lane's own segment) and the row counts / slot start of
//     item k + 2 are requested when the gate loop of item k ends, and land while its survivors are written;
//   * the next block's table is requested as soon as the last wave has left the gate loop of this block's last part
//     (barrier), before the survivors are written;
//   * the survivor list is written WITHOUT re-reading the rows: a lane keeps the neighbour lines of its last kKeep
//     survivors in registers (a first version replayed the row positions and re-read the rows -- one dependent load per
//     iteration, ~10 us per item).  A wave in which some lane has more survivors replays.
// Two (or four) workgroups share a CU and the SIMD arbiter serves the OLDER waves first: without help the workgroup
// dispatched first runs its items at full speed, the other at half, and once the first has finished the second has the
// CU to itself at half the waves (trace, two workgroups of eight waves: items of 8 against 17 us, kernel end at 64 us for
// 40).  Wave priority alternating in time slices between the workgroups dispatched in the first and in the second half of
// the grid (LT_GATE_PRIO 3) evens that out; priority by the number of items done works only while a workgroup has at most
// four items (the counts wrap), priority by position inside the item puts all waves in lockstep (slower: 72 us).
#ifndef LT_GATE_ROW_AHEAD
#define LT_GATE_ROW_AHEAD 4
#endif
constexpr int kRowAhead = LT_GATE_ROW_AHEAD;
#ifndef LT_GATE_PRIO
#define LT_GATE_PRIO 3
#endif
#ifndef LT_GATE_SLICE_BIT
#define LT_GATE_SLICE_BIT 10  // 100 MHz clock: 10.24 us slices
#endif
constexpr int kKeep = 8;
template <int kW>
__global__ void __launch_bounds__(64 * kW) LT_GATE_OCC
k_gates_ln(GenArgs a, GenCfg cfg, const BlkRec *__restrict__ blk_r, const PairRec *__restrict__ pairs_r,
           const unsigned short *__restrict__ tr, const unsigned *__restrict__ run_len,
           const unsigned *__restrict__ slot_row0, uint2 *__restrict__ st_row, unsigned *__restrict__ blk_surv,
           const int *__restrict__ vorder) {
  // vorder (round 6): the blocks in the order (neighbour image, image) -- virtual block v is block vorder[v] -- and the
  // workgroups of one XCD (dispatched round-robin: blockIdx % 8) take one CONTIGUOUS eighth of that order.  A neighbour's
  // gate table (80 B x its segments) is what every block reads whole; in image-major order the ~20 blocks that share a
  // table were spread over the grid, every XCD's L2 saw every table and the kernel's traffic was 2.8 x its algorithmic
  // bytes.  Now the blocks of a neighbour run on one XCD (its table misses that L2 once), and a workgroup whose next
  // block has the same neighbour keeps the table in its LDS.  Results do not depend on the order: every block writes its
  // own survivor list.  nullptr: image-major as in round 5 (LT_TEST_GATES_IMAGE_MAJOR).
  extern __shared__ __align__(16) unsigned char smem_raw[];
  const int wave = threadIdx.x >> 6;
  const int lane = lane_id();
  const int n_parts = a.n_slots / kW;
  const int n_items = a.n_blk * n_parts;
  const int per_wg = (n_items + (int)gridDim.x - 1) / (int)gridDim.x;
  unsigned wq = blockIdx.x;
  if (vorder && (gridDim.x & 7u) == 0u) wq = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  auto blk_of = [&](int it) -> int {  // (uniform) the block of an item
    const int v = __builtin_amdgcn_readfirstlane(it / n_parts);
    return vorder ? __builtin_amdgcn_readfirstlane(vorder[v]) : v;
  };
  int item = (int)wq * per_wg;
  const int wg_parity = (int)(2u * blockIdx.x >= gridDim.x);  // dispatched in the first / second round over the CUs (a guess)
  (void)wg_parity;
  const int item_end = min(n_items, item + per_wg);
  if (item >= item_end) return;
  // the neighbour's SegGate records, 1 KB per instruction and wave: LDS address = wave-uniform base + lane x 16
  auto issue_table = [&](int bb) {
    const BlkRec *rp = blk_r + bb;
    const char *src = reinterpret_cast<const char *>(a.gates + rp->g2);
    const int bytes = rp->M2 * (int)sizeof(SegGate);
    for (int o = wave * 1024; o < bytes; o += kW * 1024)
      if (o + lane * 16 < bytes)
        __builtin_amdgcn_global_load_lds((gbl_void_t *)(src + o + lane * 16), (lds_void_t *)(smem_raw + o), 16, 0, 0);
  };
  // per-lane inputs of an item: rows of the lane's line in the block and first row of the wave's slot ...
  auto load_counts = [&](int it, unsigned &len_o, unsigned &row0_o) {
    len_o = 0;
    row0_o = 0;
    if (it >= item_end) return;
    const int bb = blk_of(it), pp = it % n_parts;
    const BlkRec *rp = blk_r + bb;
    const int M1 = rp->M1;
    const int sl = pp * kW + wave;
    const int ln = 64 * sl + lane;
    if (ln < M1) len_o = run_len[rp->lbase + ln];
    row0_o = slot_row0[(size_t)bb * a.n_slots + sl];
  };
  // ... and the lane's own segment (endpoints and rays: the first 80 bytes of the record)
  double2 e0, e1, e2, e3, e4;
  auto load_seg = [&](int it) {
    if (it >= item_end) return;
    const int bb = blk_of(it), pp = it % n_parts;
    const BlkRec *rp = blk_r + bb;
    const long long g1 = rp->g1;
    const int M1 = rp->M1;
    const int ln = 64 * (pp * kW + wave) + lane;
    const double2 *p1 = reinterpret_cast<const double2 *>(a.segs + g1 + (ln < M1 ? ln : 0));
    e0 = p1[0]; e1 = p1[1]; e2 = p1[2]; e3 = p1[3]; e4 = p1[4];
  };
  // rows: group g = iterations [g kRowAhead, (g + 1) kRowAhead); the rows of iteration j are adjacent in `tr`, the
  // lane's one is the (number of lower lanes that still have a row)-th of them.  `base`: first row of iteration j0.
  auto load_group = [&](unsigned ln_len, unsigned &base, unsigned j0, unsigned *dst) {
#pragma unroll
    for (int u = 0; u < kRowAhead; ++u) {
      const bool act = j0 + (unsigned)u < ln_len;
      const unsigned long long m = __ballot(act);
      dst[u] = 0;
      if (act) dst[u] = tr[base + (unsigned)__popcll(m & lanemask_lt())];
      base += (unsigned)__popcll(m);
    }
  };
  unsigned len, row0, len_n, row0_n;  // this item's / the next item's
  unsigned nbq[kRowAhead];
  unsigned base_first;
  issue_table(blk_of(item));
  load_counts(item, len, row0);
  load_seg(item);
  load_counts(item + 1, len_n, row0_n);
  base_first = row0;
  load_group(len, base_first, 0, nbq);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (; item < item_end; ++item) {
    const int b = blk_of(item), part = item % n_parts;
    // (uniform over the workgroup) the block of the next item, -1: the table stays -- the same block's next part, or a
    // block with the same neighbour (the order groups them)
    int nb_blk = (item + 1 < item_end && part == n_parts - 1) ? blk_of(item + 1) : -1;
    if (nb_blk >= 0 && blk_r[nb_blk].g2 == blk_r[b].g2 && blk_r[nb_blk].M2 == blk_r[b].M2) nb_blk = -1;
    const int slot = part * kW + wave;
    const unsigned lin = (unsigned)b * (unsigned)a.n_slots + (unsigned)slot;
#if LT_GATE_PRIO == 3
    set_prio((int)(((wall_clock64() >> LT_GATE_SLICE_BIT) ^ (unsigned long long)wg_parity) & 1ull) + 1);
#endif
    LT_TRACE_MARK(0, lin, 0);
    const int line = 64 * slot + lane;
    const unsigned my_len = len, my_row0 = row0;
    const unsigned maxlen = wave_max_u32(my_len);
    unsigned pass_bits = 0, und_bits = 0;
    unsigned k0 = 0, k1 = 0, k2 = 0, k3 = 0;  // neighbour lines of the lane's last kKeep survivors, 16 bits each
    if (maxlen > 0) {
      unsigned base = base_first;  // behind the first group (requested at the end of the previous item)
      unsigned nbn[kRowAhead];
      double F[9];
#pragma unroll
      for (int k = 0; k < 9; ++k) F[k] = pairs_r[b].F[k];
      const double d1x = e0.x - e1.x, d1y = e0.y - e1.y;
      const double q1 = __builtin_fma(d1x, d1x, d1y * d1y);
      const GateEpi ea = gate3_epi(F, e0.x, e0.y), eb = gate3_epi(F, e1.x, e1.y);
      for (unsigned j0 = 0; j0 < maxlen; j0 += kRowAhead) {
#if LT_GATE_PRIO == 3
        set_prio((int)(((wall_clock64() >> LT_GATE_SLICE_BIT) ^ (unsigned long long)wg_parity) & 1ull) + 1);
#endif
        if (j0 + kRowAhead < maxlen) load_group(my_len, base, j0 + kRowAhead, nbn);
#pragma unroll
        for (int u = 0; u < kRowAhead; ++u) {
          const unsigned j = j0 + (unsigned)u;
          if (j < maxlen) {  // (uniform)
            int res = 0;
            if (j < my_len) {
              const double2 *p2 = reinterpret_cast<const double2 *>(smem_raw + __umul24(nbq[u], (unsigned)sizeof(SegGate)));
              const double2 h0 = p2[0], h1 = p2[1], h2 = p2[2], h3 = p2[3], h4 = p2[4];
              res = gate3_core_fma(cfg, q1, e2.x, e2.y, e3.x, e3.y, e4.x, e4.y, h0.x, h0.y, h1.x, h1.y, h2.x, h2.y, h3.x,
                                   h3.y, h4.x, h4.y, ea, eb);
            }
            if (res != 0) {
              // 128-bit shift register of 16-bit entries: the newest survivor enters at the top
              k0 = __builtin_amdgcn_alignbit(k1, k0, 16);
              k1 = __builtin_amdgcn_alignbit(k2, k1, 16);
              k2 = __builtin_amdgcn_alignbit(k3, k2, 16);
              k3 = __builtin_amdgcn_alignbit(nbq[u], k3, 16);
              pass_bits |= 1u << j;
              und_bits |= (res == 2 ? 1u : 0u) << j;
            }
          }
        }
#pragma unroll
        for (int u = 0; u < kRowAhead; ++u) nbq[u] = nbn[u];
      }
    }
    LT_TRACE_MARK(0, lin, 1);
    if (nb_blk >= 0) {
      __syncthreads();  // nobody reads this block's table any more
      issue_table(nb_blk);
    }
    // requests for the next items (see the header): first rows and segment of item + 1, counts of item + 2
    base_first = row0_n;
    load_group(len_n, base_first, 0, nbq);
    load_seg(item + 1);
    len = len_n;
    row0 = row0_n;
    load_counts(item + 2, len_n, row0_n);
    // survivor list of the slot, lane-major (= row order): offsets by a wave scan of the per-lane counts
    unsigned total = 0;
    if (maxlen > 0) {
      const unsigned cnt = (unsigned)__popc(pass_bits);
      const unsigned incl = wave_incl_scan_u32(cnt);
      total = (unsigned)__builtin_amdgcn_readlane((int)incl, 63);
      // The block's survivors form ONE dense list from its first row on (stage B cuts it into rounds of 64): the slot's
      // piece goes where the block's cursor stands.  The pieces of a block land in the order the waves get there --
      // that is fine: a (block, line) run lies inside one piece, and nothing downstream depends on the order of runs.
      unsigned off = 0;
      if (total > 0 && lane == 0) off = atomicAdd(&blk_surv[b], total);
      off = (unsigned)__builtin_amdgcn_readfirstlane((int)off);
      const unsigned dst0 = (unsigned)blk_r[b].rb + off + incl - cnt;
      if (__ballot(cnt > (unsigned)kKeep) == 0ull) {
        // the register holds the survivors newest-first from the top: entry kKeep - 1 is the last one, kKeep - cnt the first
        unsigned bits = pass_bits;
#pragma unroll
        for (int k = kKeep - 1; k >= 0; --k) {
          const int back = kKeep - 1 - k;  // this entry is the (cnt - 1 - back)-th survivor
          if ((unsigned)back < cnt) {
            const unsigned j = 31u - (unsigned)__builtin_clz(bits);
            bits &= ~(1u << j);
            const unsigned w = k >= 6 ? k3 : (k >= 4 ? k2 : (k >= 2 ? k1 : k0));
            const unsigned v = (k & 1) ? (w >> 16) : (w & 0xFFFFu);
            st_row[dst0 + cnt - 1u - (unsigned)back] =
                make_uint2((unsigned)line | (((und_bits >> j) & 1u) ? 0x80000000u : 0u), v);
          }
        }
      } else if (total > 0) {
        // some lane has more survivors than it keeps: replay the row positions and read the rows again
        unsigned dst = dst0;
        unsigned bs = my_row0;
        for (unsigned j = 0; j < maxlen; ++j) {
          const bool act = j < my_len;
          const unsigned long long m = __ballot(act);
          if ((pass_bits >> j) & 1u) {
            const unsigned v = tr[bs + (unsigned)__popcll(m & lanemask_lt())];
            st_row[dst++] = make_uint2((unsigned)line | (((und_bits >> j) & 1u) ? 0x80000000u : 0u), v);
          }
          bs += (unsigned)__popcll(m);
        }
      }
    }
    LT_TRACE_MARK(0, lin, 2);
    if (nb_blk >= 0) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's share of the next table has landed
      __syncthreads();                                    // ... and everybody's
    }
    LT_TRACE_MARK(0, lin, 3);
  }
}

// One wave per (block, group).  (A persistent-wave variant with the next item's record prefetched was
// measured slower: the kernel is bound by gather / scatter throughput, not by latency.)
// kExtra: additionally the optional proposals of steps 1.1 and 2 (base_line_triangulator.cc:183-281) -- per
// connection the candidates in the reference's order many-points, one-point (one per shared point),
// vp(l1), vp(l2), algebraic.  Which
// of them are active is a run-time property (a.seg_pts / a.seg_vp may be null).
// kTS: slots per group
template <bool kExtra, int kTS>
__global__ void __launch_bounds__(64 * kTriWaves)
k_tri_rows(GenArgs a, GenCfg cfg, const Cam *__restrict__ cams_r, const PairRec *__restrict__ pairs_r,
           const BlkRec *__restrict__ blk_r, const unsigned *__restrict__ slot_row0) {
  extern __shared__ __align__(16) unsigned char smem_raw[];  // per wave: 64 x (CRec | unc | key)
  const int wave = threadIdx.x >> 6;
  const int lane = lane_id();
  const int b = blockIdx.y;
  const int n_groups = a.n_slots / kTS;
  const int g = blockIdx.x * kTriWaves + wave;
  if (g >= n_groups) return;
  const BlkRec *rec = blk_r + b;
  const long long rb = rec->rb, re = rec->re;
  const long long r0 = slot_first_row(slot_row0, a.n_slots, b, g * kTS, rb);
  const unsigned lin = (unsigned)b * (unsigned)n_groups + (unsigned)g;
  LT_TRACE_MARK(1, lin, 0);
  if (r0 >= re) {
    if (lane == 0) a.wave_count[lin] = 0;
    return;
  }
  const int i1 = rec->i1, i2 = rec->i2, nbslot = rec->nbslot;
  const long long g1 = rec->g1, g2 = rec->g2;
  const PairRec *pr = pairs_r + b;
  const long long lbase = a.cnt_bl ? rec->lbase : 0;
  const long long out0 = (kExtra && a.group_base) ? a.group_base[lin] : r0;  // first staging slot of the group's list
  const bool wr = !(kExtra && a.count_only);
  // survivor lists of the group's slots, walked as one concatenated list
  unsigned cs[kTS + 1];
  cs[0] = 0;
#pragma unroll
  for (int k = 0; k < kTS; ++k)
    cs[k + 1] = cs[k] + a.surv_count[(size_t)b * a.n_slots + (size_t)g * kTS + k];
  const unsigned n_s = cs[kTS];
  unsigned wcount = 0;
#ifdef LT_TRACE
  if (n_s > 0) LT_TRACE_MARK(1, lin, 1);
#endif
  for (unsigned e0 = 0; e0 < n_s; e0 += 64) {
    const unsigned e = e0 + lane;
    bool ok = false;
    bool okx[3] = {false, false, false};  // many-points, vp(l1), vp(l2)
    unsigned n_one = 0;                    // one-point: shared points that gave a candidate
    GenOut o;
    int line = 0, ng = 0;
    // the extra proposals are evaluated once for their validity and a second time when they are
    // written (in candidate order), instead of being kept in registers next to the algebraic one
    auto extra = [&](int which, GenOut *dst) -> bool {
      const Seg &s1 = a.segs[g1 + line];
      const Seg &s2 = a.segs[g2 + ng];
      if (which == 0) {
        const long long pa0 = a.seg_pt_off[g1 + line], pb0 = a.seg_pt_off[g2 + ng];
        bool missing = false;
        const bool r = points_candidate(cfg, cams_r[i1], cams_r[i2], s1, s2, a.seg_pts + pa0,
                                        (int)(a.seg_pt_off[g1 + line + 1] - pa0), a.seg_pts + pb0,
                                        (int)(a.seg_pt_off[g2 + ng + 1] - pb0), a.sfm_xyz, dst, &missing);
        if (missing) *a.err_flag = 2;  // a shared point3D_id without an SfM point
        return r;
      }
      return vp_candidate(cfg, cams_r[i1], cams_r[i2], s1, s2, pr->B,
                          a.seg_vp + 3 * (which == 1 ? g1 + line : g2 + ng), dst);
    };
    // one-point proposals (step 1.2): one candidate per shared point, in ascending point3D_id.  store ==
    // false: returns the NUMBER of shared points that give a candidate; store == true: evaluates them again (same
    // function, same result) and writes the candidates from staging slot p on.  No limit on the shared points of a
    // connection (round 4: the counting run sizes the staging exactly).
    auto one_points = [&](bool store, long long p) -> unsigned {
      const Seg &s1 = a.segs[g1 + line];
      const Seg &s2 = a.segs[g2 + ng];
      const long long pa0 = a.seg_pt_off[g1 + line], pb0 = a.seg_pt_off[g2 + ng];
      const SegPoint *pa = a.seg_pts + pa0, *pb = a.seg_pts + pb0;
      const int na = (int)(a.seg_pt_off[g1 + line + 1] - pa0), nb = (int)(a.seg_pt_off[g2 + ng + 1] - pb0);
      unsigned n_ok = 0;
      int i = 0, j = 0;
      while (i < na && j < nb) {
        const int ia = pa[i].p3d_id, ib = pb[j].p3d_id;
        if (ia < ib) { ++i; continue; }
        if (ib < ia) { ++j; continue; }
        d3 P = mk3(0, 0, 0);
        bool okp = true;
        if (a.sfm_xyz) {
          const int sidx = pa[i].sfm;
          if (sidx < 0) { *a.err_flag = 2; okp = false; }
          else P = mk3(a.sfm_xyz[3 * sidx], a.sfm_xyz[3 * sidx + 1], a.sfm_xyz[3 * sidx + 2]);
        } else {
          okp = tri_point(cams_r[i1], cam_ray(cams_r[i1], d2{pa[i].x, pa[i].y}), cams_r[i2],
                          cam_ray(cams_r[i2], d2{pb[j].x, pb[j].y}), &P);
        }
        if (okp) {
          GenOut ov;
          if (one_point_candidate(cfg, cams_r[i1], cams_r[i2], s1, s2, P, &ov)) {
            ++n_ok;
            if (store) {
              ov.r.nb_slot = lite_pack(nbslot, i2);
              ov.r.ng_line = ng;
              a.st_r[p] = ov.r;
              a.st_unc[p] = ov.unc;
              a.st_key[p] = (unsigned)(g1 + line);
              ++p;
            }
          }
        }
        ++i; ++j;
      }
      return n_ok;
    };
    if (e < n_s) {
      int k = 0;
      unsigned first = 0;
#pragma unroll
      for (int t = 1; t < kTS; ++t)
        if (e >= cs[t]) { k = t; first = cs[t]; }
      const long long rs0 = slot_first_row(slot_row0, a.n_slots, b, g * kTS + k, rb);
      const uint2 u = a.st_row[rs0 + (e - first)];
      line = (int)(u.x & 0x7FFFFFFFu);
      ng = (int)u.y;
      const Seg &s1 = a.segs[g1 + line];
      const Seg &s2 = a.segs[g2 + ng];
      ok = true;
      if (u.x >> 31) ok = gen_gates(cfg, s1, s2, pr->F);  // the cheap gates could not decide
      if (ok) ok = gen_finish(cfg, cams_r[i1], cams_r[i2], s1, s2, pr->B, &o);
      if (kExtra) {
        // both segments long enough (:166,177) -- with extra proposals stage A lets every row through
        L2 l1{mk2(s1.x1, s1.y1), mk2(s1.x2, s1.y2)};
        L2 l2{mk2(s2.x1, s2.y1), mk2(s2.x2, s2.y2)};
        const bool len_ok = !(len(l1) <= cfg.min_length_2d) && !(len(l2) <= cfg.min_length_2d);
        GenOut tmp;
        if (len_ok && a.seg_pts && a.many_on) okx[0] = extra(0, &tmp);
        if (len_ok && a.seg_pts && a.one_on) n_one = one_points(false, 0);
        if (len_ok && a.seg_vp && a.seg_has_vp[g1 + line]) okx[1] = extra(1, &tmp);
        if (len_ok && a.seg_vp && a.seg_has_vp[g2 + ng]) okx[2] = extra(2, &tmp);
      }
      o.r.nb_slot = lite_pack(nbslot, i2);
      o.r.ng_line = ng;
    }
    const unsigned long long m = __ballot(ok);
    unsigned below = (unsigned)__popcll(m & lanemask_lt());
    unsigned total = (unsigned)__popcll(m);
    if (kExtra) {
      // per-lane candidate counts vary (one per shared point): wave prefix by shuffles
      const unsigned mine = (okx[0] ? 1u : 0u) + n_one + (okx[1] ? 1u : 0u) + (okx[2] ? 1u : 0u);
      const unsigned cnt = mine + (ok ? 1u : 0u);
      unsigned incl = cnt;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const unsigned t = (unsigned)__shfl_up((int)incl, d);
        if (lane >= d) incl += t;
      }
      below = incl - cnt;
      total = (unsigned)__shfl((int)incl, 63);
      long long p = out0 + wcount + below;
      // the reference's order within a connection: many-points, one-point (ascending point3D_id), vp(l1),
      // vp(l2), algebraic
#pragma unroll
      for (int w = 0; w < 3 && wr; ++w) {
        if (w == 1 && n_one) {
          (void)one_points(true, p);
          p += n_one;
        }
        if (okx[w]) {
          GenOut ov;
          (void)extra(w, &ov);
          ov.r.nb_slot = lite_pack(nbslot, i2);
          ov.r.ng_line = ng;
          a.st_r[p] = ov.r;
          a.st_unc[p] = ov.unc;
          a.st_key[p] = (unsigned)(g1 + line);
          ++p;
        }
      }
      if (wr && a.cnt_bl && mine) atomicAdd(&a.cnt_bl[lbase + line], mine);
      if (ok && wr) {
        a.st_r[p] = o.r;
        a.st_unc[p] = o.unc;
        a.st_key[p] = (unsigned)(g1 + line);
        if (a.cnt_bl) atomicAdd(&a.cnt_bl[lbase + line], 1u);
      }
    } else {
      // The batch's valid candidates go to a contiguous piece of the group's list: compact them through
      // LDS and write the piece with consecutive lanes on consecutive 16-byte units (a record-per-lane
      // store touches 64 cache lines per instruction).
      static_assert(sizeof(CRec) == 8 * 16, "record size in 16-byte units");
      double2 *Lc = reinterpret_cast<double2 *>(smem_raw) + (size_t)wave * (64 * 9 + 16);
      double *Lu = reinterpret_cast<double *>(Lc + 64 * 8);
      unsigned *Lk = reinterpret_cast<unsigned *>(Lu + 64);
      if (ok) {
        const double2 *oc = reinterpret_cast<const double2 *>(&o.r);
#pragma unroll
        for (int k = 0; k < 8; ++k) Lc[below * 8 + k] = oc[k];
        Lu[below] = o.unc;
        Lk[below] = (unsigned)(g1 + line);
        // (one atomic per candidate costs this kernel 4 us -- 54.8 us without; one per RUN of equal keys in the compacted
        // batch, a plain store where the run touches neither end of the batch: 70.7 us, the scan over the LDS keys costs more)
        if (a.cnt_bl) atomicAdd(&a.cnt_bl[lbase + line], 1u);
      }
      wave_lds_sync();
      const long long p0 = out0 + wcount;
      double2 *dc = reinterpret_cast<double2 *>(a.st_r + p0);
      for (unsigned u = lane; u < total * 8u; u += 64) dc[u] = Lc[u];
      if ((unsigned)lane < total) {
        a.st_unc[p0 + lane] = Lu[lane];
        a.st_key[p0 + lane] = Lk[lane];
      }
      wave_lds_sync();
    }
    wcount += total;
#ifdef LT_TRACE
    if (e0 == 0) LT_TRACE_MARK(1, lin, 3);
#endif
  }
  LT_TRACE_MARK(1, lin, 2);
  if (lane == 0) a.wave_count[lin] = wcount;
}

// Stage B of the line-slot form: the survivors of block b are ONE dense list st_row[rb ...] of blk_surv[b] entries
// (k_gates_ln); round r = its entries [64 r, 64 r + 64), one wave per round, every round but a block's last one full
// (groups of slots ran three rounds for ~140 survivors: 73 % of the lanes).  The valid candidates of a round go, compacted
// and in list order, to st_*[rb + 64 r ...]; round_count[blk_rnd0[b] + r] holds their number (blk_rnd0: exclusive sum of
// ceil(rows / 64) over the blocks -- a slot for every round a block could have).
// Persistent workgroups (as many as are resident at once): a round is ~5 us of a wave, and one short-lived wave per round
// -- 18 000 of them, in 8 000 workgroups -- was bound by the rate at which workgroups can be launched (trace: 2 000-2 500
// of 4 096 wave slots occupied).  Workgroup g takes blocks g, g + gridDim.x, ...; its kTriWaves waves share the rounds
// of a block (round r goes to wave (r - b) mod kTriWaves, so that the first rounds do not all land on wave 0) and move on
// to the next block independently of each other.  (Time-sliced wave priorities between the workgroups of a CU, which
// help k_gates_ln, cost this kernel 10 us; a staggered start of the four workgroups of a CU 2 us: not kept.  Forcing
// four waves per SIMD -- 127 registers with six spilled, against 131 and three waves -- is worth 10 us.)
__global__ void __launch_bounds__(64 * kTriWaves) __attribute__((amdgpu_waves_per_eu(4, 4)))
k_tri_rounds(GenArgs a, GenCfg cfg, const Cam *__restrict__ cams_r, const PairRec *__restrict__ pairs_r,
             const BlkRec *__restrict__ blk_r, const unsigned *__restrict__ blk_surv, const unsigned *__restrict__ blk_rnd0,
             unsigned *__restrict__ round_count, const int *__restrict__ vorder, unsigned *__restrict__ unit_ctr) {
  extern __shared__ __align__(16) unsigned char smem_raw[];  // per wave: 64 x (CRec | unc | key)
  const int wave = threadIdx.x >> 6;
  const int lane = lane_id();
  // vorder (round 6, as in k_gates_ln): the blocks in (neighbour, image) order, the workgroups of XCD x = blockIdx % 8 take
  // the x-th eighth of it -- the neighbour-side segment records a block gathers (128 B each, half of this kernel's reads)
  // then stay in that XCD's L2 across the ~20 blocks that share the neighbour.  Measured on a streamed chunk (250 images x
  // 600 segments): L2 -> memory read requests 2.47 M -> 0.64 M (316 -> 82 MB), the kernel's time unchanged (0.176 ms): it
  // is bound by its dependent chains, not by these reads -- kept for the traffic (530 -> 244 MB per chunk, 1.7 x algorithmic)
  const bool xcd_major = vorder != nullptr && (gridDim.x & 7u) == 0u;
  const int n8 = xcd_major ? (a.n_blk + 7) / 8 : a.n_blk;
  const int v_begin = xcd_major ? (int)(blockIdx.x & 7u) * n8 : 0;
  const int v_end = xcd_major ? min(a.n_blk, v_begin + n8) : a.n_blk;
  const int v_step = xcd_major ? (int)(gridDim.x >> 3) : (int)gridDim.x;
  // UNITS (round 6, second half): a unit = the rounds x, x + 4, ... of one block = what one wave does in a block.  A wave's
  // FIRST unit is static as before (its workgroup's first block, rounds (wave + block) % 4), every later one is CLAIMED from
  // its XCD's counter over the queue's remaining units (unit u = block u / 4 of the XCD's eighth, rounds u % 4), the claim
  // issued at the start of the unit before.  With blocks dealt g, g + G, ... the waves had 2 blocks each whatever the blocks
  // held (a block has 200 to 1 200 survivors): trace at 100 x 500 -- 1 700 waves in rounds at 20 us, 857 at 30 us, 78 at 40 us
  // of 47.  Measured: BASELINE config 3 (20 000 blocks, 19 per workgroup) 0.855 -> 0.778 ms; at 100 x 500 (2 000 blocks, 2 per
  // workgroup) 50.5 -> 56 us -- the four units of a block then run on four CUs and each fetches the block's cameras, pair
  // record and neighbour segments for itself, which costs more than two blocks per workgroup can be out of balance.  So:
  // claims from four blocks per workgroup on; unit_ctr == nullptr (or a grid that is not a multiple of 8): the static deal.
  const bool claims = xcd_major && unit_ctr != nullptr && (long long)a.n_blk >= 4ll * (long long)gridDim.x;
  unsigned *ctr = claims ? unit_ctr + 32u * (blockIdx.x & 7u) : nullptr;
  const unsigned n_static = claims ? (gridDim.x >> 3) * (unsigned)kTriWaves : 0u;  // units dealt statically per queue
  unsigned c_raw = 0;
  bool first = true;
  int v = v_begin + (xcd_major ? (int)(blockIdx.x >> 3) : (int)blockIdx.x);
  int x_dyn = -1;
  for (;;) {
  if (claims) {
    if (!first) {
      const unsigned u = n_static + (unsigned)__builtin_amdgcn_readfirstlane((int)c_raw);
      v = v_begin + (int)(u / (unsigned)kTriWaves);
      x_dyn = (int)(u % (unsigned)kTriWaves);
    }
    if (v >= v_end) break;
    if (lane == 0) c_raw = atomicAdd(ctr, 1u);  // the unit after this one (read when this one is done)
  } else {
    if (!first) v += v_step;
    if (v >= v_end) break;
  }
  first = false;
  const int b = vorder ? vorder[v] : v;
  const int x = x_dyn >= 0 ? x_dyn : (wave + b) % kTriWaves;
  const unsigned n_s = blk_surv[b];
  const int n_rounds = (int)((n_s + 63u) >> 6);
  if (x >= n_rounds) continue;
  const BlkRec *rec = blk_r + b;
  const long long rb = rec->rb;
  const unsigned rnd0 = blk_rnd0[b];
  const int i1 = rec->i1, i2 = rec->i2, nbslot = rec->nbslot;
  const long long g1 = rec->g1, g2 = rec->g2;
  const PairRec *pr = pairs_r + b;
  const long long lbase = rec->lbase;
  for (int r = x; r < n_rounds; r += kTriWaves) {
    LT_TRACE_MARK(1, rnd0 + (unsigned)r, 0);
    const unsigned e = 64u * (unsigned)r + (unsigned)lane;
    bool ok = false;
    GenOut o;
    int line = 0;
    if (e < n_s) {
      const uint2 u = a.st_row[rb + e];
      line = (int)(u.x & 0x7FFFFFFFu);
      const int ng = (int)u.y;
#ifdef LT_TRACE
      if (line >= 0) LT_TRACE_MARK(1, rnd0 + (unsigned)r, 1);
#endif
      const Seg &s1 = a.segs[g1 + line];
      const Seg &s2 = a.segs[g2 + ng];
      ok = true;
      if (u.x >> 31) ok = gen_gates(cfg, s1, s2, pr->F);  // the cheap gates could not decide
      if (ok) ok = gen_finish(cfg, cams_r[i1], cams_r[i2], s1, s2, pr->B, &o);
      o.r.nb_slot = lite_pack(nbslot, i2);
      o.r.ng_line = ng;
    }
    LT_TRACE_MARK(1, rnd0 + (unsigned)r, 3);
    const unsigned long long m = __ballot(ok);
    const unsigned below = (unsigned)__popcll(m & lanemask_lt());
    const unsigned total = (unsigned)__popcll(m);
    // The round's valid candidates go to a contiguous piece: compact them through LDS and write the piece with
    // consecutive lanes on consecutive 16-byte units (a record-per-lane store touches 64 cache lines per instruction).
    static_assert(sizeof(CRec) == 8 * 16, "record size in 16-byte units");
    double2 *Lc = reinterpret_cast<double2 *>(smem_raw) + (size_t)wave * (64 * 9 + 16);
    double *Lu = reinterpret_cast<double *>(Lc + 64 * 8);
    unsigned *Lk = reinterpret_cast<unsigned *>(Lu + 64);
    if (ok) {
      const double2 *oc = reinterpret_cast<const double2 *>(&o.r);
#pragma unroll
      for (int k = 0; k < 8; ++k) Lc[below * 8 + k] = oc[k];
      Lu[below] = o.unc;
      Lk[below] = (unsigned)(g1 + line);
      atomicAdd(&a.cnt_bl[lbase + line], 1u);  // valid candidates per (block, line): the placement's prefixes
    }
    wave_lds_sync();
    const long long p0 = rb + 64ll * r;
    double2 *dc = reinterpret_cast<double2 *>(a.st_r + p0);
    for (unsigned u = lane; u < total * 8u; u += 64) dc[u] = Lc[u];
    if ((unsigned)lane < total) {
      a.st_unc[p0 + lane] = Lu[lane];
      a.st_key[p0 + lane] = Lk[lane];
    }
    if (lane == 0) round_count[rnd0 + (unsigned)r] = total;
    wave_lds_sync();
    LT_TRACE_MARK(1, rnd0 + (unsigned)r, 2);
  }
  }
}

#ifndef LT_PLACE_WG_PER_CU
#define LT_PLACE_WG_PER_CU 8
#endif
// Placement for the round lists of k_tri_rounds: as k_place, one wave per round; a (block, line) run that begins in an
// earlier round of the block is followed back through the previous rounds' lists.
__global__ void __launch_bounds__(256)
k_place_rounds(const long long *__restrict__ m_off, const int *__restrict__ blk_img, const long long *__restrict__ seg_off,
               const long long *__restrict__ blk_line_base, const unsigned *__restrict__ base_bl,
               const long long *__restrict__ tri_off, const CRec *__restrict__ st_r, const double *__restrict__ st_unc,
               const unsigned *__restrict__ st_key, CRec *__restrict__ cand, double *__restrict__ cand_unc,
               unsigned *__restrict__ cand_node, unsigned *__restrict__ perm, const unsigned *__restrict__ blk_surv,
               const unsigned *__restrict__ blk_rnd0, const unsigned *__restrict__ round_count, int n_blk) {
  const int wave = threadIdx.x >> 6;
  const int lane = lane_id();
  for (int b = (int)blockIdx.x; b < n_blk; b += (int)gridDim.x) {  // persistent, as k_tri_rounds
  const int x = (wave + b) & 3;
  const int n_rounds = (int)((blk_surv[b] + 63u) >> 6);
  if (x >= n_rounds) continue;
  const long long rb = m_off[b];
  const unsigned rnd0 = blk_rnd0[b];
  const long long g1 = seg_off[blk_img[b]];
  const long long lbase = blk_line_base[b];
  // (the count and the keys of the wave's next round are requested before this round's dependent loads: the kernel is a
  // chain of small loads -- count -> keys -> offsets -- and the next round's first two links cost nothing this way; the key
  // array has 64 entries of slack behind the last row)
  unsigned count_n = round_count[rnd0 + (unsigned)x];
  unsigned key_n = st_key[rb + 64ll * x + lane];
  for (int r = x; r < n_rounds; r += 4) {
    const unsigned count = count_n;
    const unsigned key_raw = key_n;
    if (r + 4 < n_rounds) {
      count_n = round_count[rnd0 + (unsigned)(r + 4)];
      key_n = st_key[rb + 64ll * (r + 4) + lane];
    }
    if (count == 0) continue;
    const long long s0 = rb + 64ll * r;
    const bool act = (unsigned)lane < count;
    // rank within the (block, line) run: inside the list the distance to the run's first lane (ballot of the run heads);
    // the run that reaches back beyond the list is followed through the earlier rounds (wave-uniform)
    const unsigned key = act ? key_raw : 0xFFFFFFFFu;
    const unsigned prev = (unsigned)__shfl_up((int)key, 1);
    const bool head = act && (lane == 0 || key != prev);
    const unsigned long long heads = __ballot(head);
    const int my_head = 63 - __builtin_clzll(heads & ((2ull << lane) - 1ull) | 1ull);
    const unsigned key0 = (unsigned)__builtin_amdgcn_readfirstlane((int)key);
    unsigned carry = 0;
    for (int pr_ = r - 1; pr_ >= 0; --pr_) {
      const unsigned pc = round_count[rnd0 + (unsigned)pr_];
      // an empty list cannot tell whether the run continues further back: keep walking (bounded by the block)
      if (pc == 0) continue;
      const long long j = (long long)pc - 64 + lane;  // the list's last 64 (all of its) entries, lane 63 = the last one
      const bool valid = j >= 0;
      const unsigned k = valid ? st_key[rb + 64ll * pr_ + j] : 0u;
      const unsigned long long m = __ballot(valid && k == key0);
      const unsigned lead = m == ~0ull ? 64u : (unsigned)__builtin_clzll(~m);
      carry += lead;
      if (lead < pc) break;  // a different key precedes: the run starts here
    }
    unsigned pos32 = 0;
    if (act) {
      const unsigned rank = (unsigned)(lane - my_head) + (my_head == 0 ? carry : 0u);
      const long long pos = tri_off[key] + base_bl[lbase + (long long)(key - g1)] + rank;
      pos32 = (unsigned)pos;  // candidate positions fit 32 bits (cand_node / tri counts are 32-bit)
      cand_node[pos] = key;
      if (perm) perm[pos] = (unsigned)(s0 + lane);
    }
    if (perm) continue;
    // (LT_TEST_PLACE_COPY) cooperative copy in 16-byte units, see k_place
    const double2 *src_c = reinterpret_cast<const double2 *>(st_r + s0);
    double2 *dst_c = reinterpret_cast<double2 *>(cand);
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const unsigned u = (unsigned)it * 64u + (unsigned)lane;
      const unsigned ci = u >> 3, piece = u & 7u;
      const unsigned p = (unsigned)__shfl((int)pos32, (int)ci);
      if (u < count * 8u) dst_c[(size_t)p * 8u + piece] = src_c[u];
    }
    if (act) cand_unc[pos32] = st_unc[s0 + lane];
  }
  }
}

// Fast path: exclusive prefix of cnt_bl over the neighbour blocks of every node (into base_bl) and the
// node's candidate count.
// The exclusive scan of the node counts (tri_off) is part of the same kernel: single-pass scan with decoupled
// look-back.  A workgroup takes a ticket (its tile = 256 consecutive nodes in ticket order, so every
// predecessor is already running), publishes its tile total, and its first wave looks back 64 predecessors at
// a time until it meets one whose inclusive prefix is known.  status[0] = ticket counter, status[1 + tile] =
// flag << 62 | value (flag 1: tile total, 2: inclusive prefix); zeroed by k_build_pairs of the same run.
// State words are read / written with device-scope atomic RMWs (the L2s of the XCDs are not coherent for
// plain loads).  (rocPRIM's device scan of the same 50 001 entries is two launches, ~10 us.)
__global__ void __launch_bounds__(256)
k_node_prefix(long long G, const int *__restrict__ node_img, const long long *__restrict__ seg_off,
              const long long *__restrict__ nb_off, const long long *__restrict__ blk_line_base,
              unsigned *__restrict__ cnt_bl, unsigned *__restrict__ base_bl, unsigned *__restrict__ n_tris,
              long long *__restrict__ tri_off, unsigned long long *__restrict__ status, int *__restrict__ err_flag,
              uint4 *__restrict__ node_rec) {
  __shared__ unsigned s_tile;
  __shared__ long long s_wave[4];
  __shared__ long long s_prefix;
  if (threadIdx.x == 0) s_tile = (unsigned)atomicAdd(&status[0], 1ull);
  __syncthreads();
  const long long tile = s_tile;
  const long long g = tile * 256 + threadIdx.x;
  const int lane = lane_id(), wave = threadIdx.x >> 6;
  unsigned run = 0;
  unsigned nbv = 0;  // (nb_off[img] << 8) | neighbours of the node's image: CandMeta::nb
  if (g < G) {
    int img = node_img[g];
    int line = (int)(g - seg_off[img]);
    // all loads of a chunk of 16 blocks first, then the stores: interleaved, every load would have to wait for
    // the store before it (same array as far as the compiler can tell) -- 20 serial round trips per thread
    const long long b0 = nb_off[img], b1 = nb_off[img + 1];
    nbv = ((unsigned)b0 << 8) | (unsigned)(b1 - b0);
    for (long long bb = b0; bb < b1; bb += 16) {
      long long e[16];
      unsigned c[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        e[k] = (bb + k < b1) ? blk_line_base[bb + k] + line : -1;
      }
#pragma unroll
      for (int k = 0; k < 16; ++k) c[k] = e[k] >= 0 ? cnt_bl[e[k]] : 0u;
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        if (e[k] >= 0) {
          cnt_bl[e[k]] = 0;  // every counter is read exactly once: leave the array clean for the next run
          base_bl[e[k]] = run;
        }
        run += c[k];
      }
    }
  }
  // tile-local inclusive scan
  long long incl = (long long)run;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const long long o = __shfl_up(incl, d);
    if (lane >= d) incl += o;
  }
  if (lane == 63) s_wave[wave] = incl;
  __syncthreads();
  long long wbase = 0;
  for (int w = 0; w < wave; ++w) wbase += s_wave[w];
  const long long total = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
  constexpr unsigned long long kMask = (1ull << 62) - 1ull;
  if (wave == 0) {
    long long prefix = 0;
    if (tile > 0) {
      if (lane == 0) atomicExch(&status[1 + tile], (1ull << 62) | (unsigned long long)total);
      long long hi = tile;  // predecessors [0, hi) still to be accounted for
      bool done = false;
      while (!done) {
        const long long k = hi - 1 - lane;  // lane 0: the nearest predecessor
        unsigned long long v = 2ull << 62;   // beyond tile 0: "inclusive prefix 0"
        if (k >= 0) {
          int budget = 1 << 22;
          do {
            v = atomicAdd(&status[1 + k], 0ull);
          } while ((v >> 62) == 0ull && --budget > 0);
          if ((v >> 62) == 0ull) {  // cannot happen (the predecessor holds an earlier ticket); never hang
            *err_flag = 4;
            v = 2ull << 62;
          }
        }
        const unsigned long long incl_lanes = __ballot((v >> 62) == 2ull);
        const int first = incl_lanes ? __builtin_ctzll(incl_lanes) : 64;
        long long part = (lane <= first) ? (long long)(v & kMask) : 0ll;
        for (int d = 32; d >= 1; d >>= 1) part += __shfl_xor(part, d);
        prefix += part;
        if (first < 64) done = true;
        else hi -= 64;
      }
    }
    if (lane == 0) {
      atomicExch(&status[1 + tile], (2ull << 62) | (unsigned long long)(prefix + total));
      s_prefix = prefix;
    }
  }
  __syncthreads();
  const long long off = s_prefix + wbase + incl - (long long)run;
  if (g <= G) {
    tri_off[g] = off;
    n_tris[g] = run;
  }
  // the scoring prologue record of the node's candidates (CandMeta, lt_kernels_score.hip), 16 bytes per node: k_cand_meta
  // then copies it with one gather per candidate instead of five (tri_off twice, node_img, nb_off twice)
  if (node_rec && g < G) node_rec[g] = uint4{(unsigned)(off & 0xFFFFFFFFll), (unsigned)(off >> 32), run, nbv};
}

// Fast path: move every staged candidate to its final, reference-ordered position
//   pos = tri_off[node] + (valid candidates of the node in earlier neighbour blocks) + rank in its run.
// One wave per slot.  Rows of a block are sorted by line id,
// so the candidates of one (block, line) run are adjacent in the row-ordered lists; the rank is
// found by looking back over equal keys (crossing into the previous wave's list if the run does).
template <int kTS>
__global__ void __launch_bounds__(256)
k_place(const long long *__restrict__ m_off, const int *__restrict__ blk_img,
        const long long *__restrict__ seg_off, const long long *__restrict__ blk_line_base,
        const unsigned *__restrict__ base_bl, const unsigned *__restrict__ wave_count,
        const long long *__restrict__ tri_off, const CRec *__restrict__ st_r,
        const double *__restrict__ st_unc, const unsigned *__restrict__ st_key, CRec *__restrict__ cand,
        double *__restrict__ cand_unc, unsigned *__restrict__ cand_node, int n_groups,
        const long long *__restrict__ group_base, unsigned *__restrict__ perm,
        const unsigned *__restrict__ slot_row0) {
  const int wave = threadIdx.x >> 6;
  const int lane = lane_id();
  const int b = blockIdx.y;
  const int g = blockIdx.x * 4 + wave;
  if (g >= n_groups) return;
  const int n_slots = n_groups * kTS;
  const long long rb = m_off[b], re = m_off[b + 1];
  const long long r0 = slot_first_row(slot_row0, n_slots, b, g * kTS, rb);
  if (r0 >= re) return;
  const unsigned lin = (unsigned)b * (unsigned)n_groups + (unsigned)g;
  const unsigned count = wave_count[lin];
  if (count == 0) return;
  const long long g1 = seg_off[blk_img[b]];
  const long long lbase = blk_line_base[b];
  // the group's first staging slot: its first row, or -- extra proposals -- the scanned list counts (GenArgs::group_base)
  const long long s0 = group_base ? group_base[lin] : r0;
  static_assert(sizeof(CRec) == 8 * 16, "record size in 16-byte units");
  for (unsigned e0 = 0; e0 < count; e0 += 64) {
    const unsigned e = e0 + lane;
    const bool act = e < count;
    unsigned pos32 = 0;
    // rank within the (block, line) run, without per-lane pointer chasing: the candidates of a run
    // are adjacent, so inside the batch the rank is the distance to the run's first lane (ballot of
    // the run heads); only the run that reaches back beyond the batch needs a look-back, and that
    // one is wave-uniform (64 earlier keys per step).
    const unsigned key = act ? st_key[s0 + e] : 0xFFFFFFFFu;
    const unsigned prev = (unsigned)__shfl_up((int)key, 1);
    const bool head = act && (lane == 0 || key != prev);
    const unsigned long long heads = __ballot(head);
    const int my_head = 63 - __builtin_clzll(heads & ((2ull << lane) - 1ull) | 1ull);
    const unsigned key0 = (unsigned)__builtin_amdgcn_readfirstlane((int)key);
    unsigned carry = 0;
    {
      long long cur_r0 = r0;
      int cur_g = g;
      long long idx_end = (long long)e0;  // entries [0, idx_end) of the current list precede the batch
      unsigned cur_lin = lin;
      while (true) {
        bool more = true;
        while (idx_end > 0) {
          const long long j = idx_end - 64 + lane;
          const bool valid = j >= 0;
          const unsigned k = valid ? st_key[(group_base ? group_base[cur_lin] : cur_r0) + j] : 0u;
          const unsigned long long m = __ballot(valid && k == key0);
          const unsigned lead = m == ~0ull ? 64u : (unsigned)__builtin_clzll(~m);
          carry += lead;
          const long long n_valid = idx_end < 64 ? idx_end : 64;
          if ((long long)lead < n_valid) { more = false; break; }  // a different key precedes: run starts here
          idx_end -= n_valid;
        }
        if (!more) break;
        if (cur_g == 0) break;  // first group of the block
        cur_g -= 1;
        cur_r0 = slot_first_row(slot_row0, n_slots, b, cur_g * kTS, rb);
        cur_lin -= 1;
        // an empty list cannot tell whether the run continues further back: keep walking (bounded by the block)
        idx_end = (long long)wave_count[cur_lin];
      }
    }
    if (act) {
      const unsigned rank = (unsigned)(lane - my_head) + (my_head == 0 ? carry : 0u);
      const long long toff = tri_off[key];
      const long long pos = toff + base_bl[lbase + (long long)(key - g1)] + rank;
      pos32 = (unsigned)pos;  // candidate positions fit 32 bits (cand_node / tri counts are 32-bit)
      cand_node[pos] = key;
      // perm != nullptr: the records stay where stage B staged them and the consumers read them through
      // perm[final position] = staging slot (4 bytes per candidate instead of moving 144)
      if (perm) perm[pos] = (unsigned)(s0 + e);
      // (writing the scoring kernel's CandMeta record here instead of running k_cand_meta was measured:
      // +9 us in this kernel against 5 us for the separate pass)
    }
    // Cooperative copy in 16-byte units: consecutive lanes read consecutive units of the (contiguous)
    // source list and write consecutive units of a destination record, so a wave touches ~1/8 of the
    // cache lines a record-per-lane copy would.
    if (perm) continue;
    const unsigned nb = min(64u, count - e0);
    const double2 *src_c = reinterpret_cast<const double2 *>(st_r + s0 + e0);
    double2 *dst_c = reinterpret_cast<double2 *>(cand);
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const unsigned u = (unsigned)it * 64u + (unsigned)lane;
      const unsigned ci = u >> 3, piece = u & 7u;
      const unsigned p = (unsigned)__shfl((int)pos32, (int)ci);
      if (u < nb * 8u) dst_c[(size_t)p * 8u + piece] = src_c[u];
    }
    if (act) cand_unc[pos32] = st_unc[s0 + e];
  }
}

// Generic path: pack the row-ordered wave lists into dense (key, source index) arrays for the sort
__global__ void __launch_bounds__(256)  // (generic rows: always the row-slot form)
k_pack_keys(const long long *__restrict__ m_off, const unsigned *__restrict__ wave_count,
            const long long *__restrict__ wave_pos, const unsigned *__restrict__ st_key,
            unsigned *__restrict__ keys_c, unsigned *__restrict__ src_c, int n_groups,
            const long long *__restrict__ group_base) {
  const int wave = threadIdx.x >> 6;
  const int lane = lane_id();
  const int b = blockIdx.y;
  const int g = blockIdx.x * 4 + wave;
  if (g >= n_groups) return;
  const long long rb = m_off[b], re = m_off[b + 1];
  const long long r0 = rb + (long long)g * kTriRows;
  if (r0 >= re) return;
  const unsigned lin = (unsigned)b * (unsigned)n_groups + (unsigned)g;
  const unsigned count = wave_count[lin];
  const long long base = wave_pos[lin];
  for (unsigned e = lane; e < count; e += 64) {
    const long long s0 = group_base ? group_base[lin] : r0;
    keys_c[base + e] = st_key[s0 + e];
    src_c[base + e] = (unsigned)(s0 + e);
  }
}

// Generic path: gather the candidates into sorted (node-major, stable) order
__global__ void k_permute(long long C, const unsigned *__restrict__ skeys, const unsigned *__restrict__ ssrc,
                          const CRec *__restrict__ st_r, const double *__restrict__ st_unc,
                          CRec *__restrict__ cand, double *__restrict__ cand_unc, unsigned *__restrict__ cand_node) {
  long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= C) return;
  unsigned src = ssrc[t];
  cand[t] = st_r[src];
  cand_unc[t] = st_unc[src];
  cand_node[t] = skeys[t];
}

// The split host-side view of the candidates (debug read-outs): Cand / CandLite at position t from record
// perm[t] (the staged records of the permutation-based store) or t (compact arrays).
__global__ void k_host_view(long long C, const unsigned *__restrict__ perm, const CRec *__restrict__ rec,
                            const double *__restrict__ unc, Cand *__restrict__ out_c, CandLite *__restrict__ out_l) {
  long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= C) return;
  const long long src = perm ? (long long)perm[t] : t;
  Cand c;
  CandLite l;
  crec_split(rec[src], unc[src], &c, &l);
  out_c[t] = c;
  out_l[t] = l;
}

static inline unsigned nblk2(long long n, int bs) { return (unsigned)((n + bs - 1) / bs); }

// slots per block: enough for the largest block, a multiple of the k_gates workgroup
int gen_slots(long long max_rows) {
  long long n = (max_rows + kRowsPerWave - 1) / kRowsPerWave;
  n = (n + kGateWaves - 1) / kGateWaves * kGateWaves;
  return (int)n;
}
// groups per block (lists of stage B / placement): wave_count[] has n_blk * gen_groups entries
int gen_groups(long long max_rows) { return gen_slots(max_rows) / kTriSlots; }
// the line-slot form: slots of 64 lines of the image
int gen_slots_ln(int max_own_segs) {
  int n = (max_own_segs + 63) / 64;
  return std::max((n + kGateWaves - 1) / kGateWaves * kGateWaves, kGateWaves);
}
int gen_max_run() { return kMaxRun; }
// the line-slot form of the staged rows (lt_upload): run starts, then run lengths / slot starts / transposed rows
void launch_rows_ln(hipStream_t st, int n_blk, int n_slots, const void *desc, const unsigned *stream,
                    const long long *blk_line_base, unsigned *rstart, int *blk_nruns, unsigned *run_len,
                    unsigned *slot_row0, unsigned short *tr, int *ln_flag) {
  if (n_blk <= 0) return;
  hipLaunchKernelGGL(k_rows_starts, dim3((unsigned)n_blk), dim3(256), 0, st, n_blk, reinterpret_cast<const RowDesc *>(desc),
                     stream, blk_line_base, rstart, blk_nruns, ln_flag);
  hipLaunchKernelGGL(k_rows_transpose, dim3(nblk2(n_slots, 4), (unsigned)n_blk), dim3(256), 0, st, n_blk, n_slots,
                     reinterpret_cast<const RowDesc *>(desc), stream, blk_line_base, rstart, blk_nruns, run_len, slot_row0, tr,
                     ln_flag);
}
#ifdef LT_TRACE
int score_read_trace(unsigned long long *host, size_t n);  // lt_kernels_score.hip: slices 2 and 3
extern "C" int lt_debug_read_trace(unsigned long long *host, size_t n) {
  const int rc = (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_trace), n * 8, 0, hipMemcpyDeviceToHost);
  return rc ? rc : score_read_trace(host, n);
}
#endif
size_t seg_gate_bytes() { return sizeof(SegGate); }
size_t seg_point_bytes() { return sizeof(SegPoint); }
size_t blk_rec_bytes() { return sizeof(BlkRec); }
void launch_expand_rows(hipStream_t st, int n_blk, const void *desc, const unsigned *stream, const unsigned *ovf, unsigned *rows) {
  if (n_blk > 0)
    hipLaunchKernelGGL(k_expand_rows, dim3((unsigned)n_blk), dim3(64 * kExpandWaves), 0, st, n_blk, reinterpret_cast<const RowDesc *>(desc), stream,
                       ovf, rows);
}
void launch_build_blk(hipStream_t st, int n_blk, const long long *m_off, const int *blk_img, const int *blk_nb,
                      const int *blk_slot, const long long *seg_off, const long long *blk_line_base, void *blkrec) {
  if (n_blk > 0)
    hipLaunchKernelGGL(k_build_blk, dim3(nblk2(n_blk, 128)), dim3(128), 0, st, n_blk, m_off, blk_img, blk_nb, blk_slot,
                       seg_off, blk_line_base, reinterpret_cast<BlkRec *>(blkrec));
}
// HOT LOOP 1: k_gates (survivor lists) + k_tri_rows (candidate lists)
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
                      unsigned *round_count, const int *blk_vorder, unsigned *tri_unit_ctr) {
  // tri_unit_ctr: eight counters, 128 B apart, zero at the start of the run: k_tri_rounds' unit claims (nullptr: static deal)
  // ln_slots > 0: the line-slot form (k_gates_ln; slots per block = ln_slots, tables tr / run_len / slot_row0) with
  // stage B in rounds of 64 survivors (k_tri_rounds; no extra proposals in this form)
  // phase 0: k_gates + k_tri_rows (no extra proposals).  Extra proposals: phase 1 = k_gates + the COUNTING run of
  // k_tri_rows (wave_count only), phase 2 = the storing run at group_base (the scanned counts), see GenArgs
  if (n_blk <= 0 || max_rows <= 0) return;
  static int n_cu = 0;
  if (n_cu == 0) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu <= 0)
      n_cu = 256;
  }
  // (the BlkRec table depends on the uploaded job only: launch_build_blk, once per lt_upload)
  // (the SegGate records are written together with the segment records: k_build_segs* in lt_kernels.hip)
  GenArgs a;
  a.m_off = m_off; a.m_pairs = m_pairs; a.blk_img = blk_img; a.blk_nb = blk_nb; a.blk_slot = blk_slot;
  a.seg_off = seg_off; a.cams = cams; a.segs = segs; a.gates = reinterpret_cast<const SegGate *>(gates);
  a.pairs = pairs; a.blk_line_base = blk_line_base; a.st_row = reinterpret_cast<uint2 *>(st_row); a.surv_count = surv_count;
  a.st_r = st_r; a.st_unc = st_unc; a.st_key = st_key; a.wave_count = wave_count; a.cnt_bl = cnt_bl;
  const bool ln = ln_slots > 0;
  a.n_slots = ln ? ln_slots : gen_slots(max_rows); a.lds_segs = lds_segs; a.lds_segs1 = lds_segs1;
  a.tr = ln ? tr : nullptr; a.run_len = ln ? run_len : nullptr; a.slot_row0 = ln ? slot_row0 : nullptr;
  a.seg_vp = seg_vp; a.seg_has_vp = seg_has_vp;
  a.seg_pt_off = seg_pt_off; a.seg_pts = reinterpret_cast<const SegPoint *>(seg_pts); a.sfm_xyz = sfm_xyz;
  a.err_flag = err_flag;
  const bool extra = seg_vp || seg_pts;
  a.group_base = extra ? group_base : nullptr;
  a.count_only = (extra && phase == 1) ? 1 : 0;
  a.many_on = many_on; a.one_on = one_on;
  a.blk = reinterpret_cast<const BlkRec *>(blkrec); a.n_blk = n_blk;
  // persistent grid: as many workgroups as fit at once (registers allow 16 waves per CU)
  // line-slot form: T2 only; workgroups of four waves while four tables fit a CU, of eight beyond
  const bool ln_w4 = ln && (size_t)lds_segs * sizeof(SegGate) <= 40 * 1024;
  const int gate_waves = ln ? (ln_w4 ? 4 : 8) : kGateWaves;
  const long long n_items = (long long)n_blk * (a.n_slots / gate_waves);
  const size_t lds = ln ? (size_t)lds_segs * sizeof(SegGate) : (size_t)(lds_segs + lds_segs1) * sizeof(SegGate);
  int per_cu = std::max(16 / gate_waves, 1);
  if (lds > 0) per_cu = (int)std::max<size_t>(std::min<size_t>(160 * 1024 / lds, (size_t)per_cu), 1);
  const unsigned n_wg = (unsigned)std::min<long long>(n_items, (long long)n_cu * per_cu);
  // the tables are sized by the largest image of the job, so "fits" is a per-launch property
  const dim3 grid(n_wg), block(64 * gate_waves);
  // (extra proposals run this in two phases: the k_gates events belong to phase 1, the one behind stage B to phase 2 --
  // the k_tri_rows figure then spans the counting run, the scan of the counts and the storing run)
  if (ev3 && phase != 2) (void)hipEventRecord(ev3[0], st);
  if (phase != 2 && ln) {
    if (ln_w4)
      hipLaunchKernelGGL((k_gates_ln<4>), grid, block, lds, st, a, cfg, a.blk, a.pairs, a.tr, a.run_len, a.slot_row0, a.st_row,
                         blk_surv, blk_vorder);
    else
      hipLaunchKernelGGL((k_gates_ln<8>), grid, block, lds, st, a, cfg, a.blk, a.pairs, a.tr, a.run_len, a.slot_row0, a.st_row,
                         blk_surv, blk_vorder);
  } else if (phase != 2) {
    if (lds_segs1 > 0 && lds_segs > 0) hipLaunchKernelGGL((k_gates<true, true>), grid, block, lds, st, a, cfg, a.blk, a.pairs);
    else if (lds_segs > 0) hipLaunchKernelGGL((k_gates<false, true>), grid, block, lds, st, a, cfg, a.blk, a.pairs);
    else if (lds_segs1 > 0) hipLaunchKernelGGL((k_gates<true, false>), grid, block, lds, st, a, cfg, a.blk, a.pairs);
    else hipLaunchKernelGGL((k_gates<false, false>), grid, block, lds, st, a, cfg, a.blk, a.pairs);
  }
  if (ev3 && phase != 2) (void)hipEventRecord(ev3[1], st);
  const size_t tri_lds = kTriWaves * (64 * 9 + 16) * sizeof(double2);
  if (ln) {
    // persistent: the workgroups that are resident at once (16 waves per CU: registers and LDS)
    const dim3 tg((unsigned)std::min<long long>(n_blk, (long long)n_cu * (16 / kTriWaves)));
    hipLaunchKernelGGL(k_tri_rounds, tg, dim3(64 * kTriWaves), tri_lds, st, a, cfg, a.cams, a.pairs, a.blk, blk_surv, blk_rnd0,
                       round_count, blk_vorder, tri_unit_ctr);
  } else {
    const dim3 tg(nblk2(a.n_slots / kTriSlots, kTriWaves), n_blk);
    if (extra)
      hipLaunchKernelGGL((k_tri_rows<true, kTriSlots>), tg, dim3(64 * kTriWaves), 0, st, a, cfg, a.cams, a.pairs, a.blk, a.slot_row0);
    else
      hipLaunchKernelGGL((k_tri_rows<false, kTriSlots>), tg, dim3(64 * kTriWaves), tri_lds, st, a, cfg, a.cams, a.pairs, a.blk,
                         a.slot_row0);
  }
  if (ev3 && phase != 1) (void)hipEventRecord(ev3[2], st);
}
void launch_node_prefix(hipStream_t st, long long G, const int *node_img, const long long *seg_off,
                        const long long *nb_off, const long long *blk_line_base, unsigned *cnt_bl,
                        unsigned *base_bl, unsigned *n_tris, long long *tri_off, unsigned long long *status,
                        int *err_flag, void *node_rec) {
  hipLaunchKernelGGL(k_node_prefix, dim3(nblk2(G + 1, 256)), dim3(256), 0, st, G, node_img, seg_off, nb_off,
                     blk_line_base, cnt_bl, base_bl, n_tris, tri_off, status, err_flag, reinterpret_cast<uint4 *>(node_rec));
}
void launch_place(hipStream_t st, int n_blk, long long max_rows, const long long *m_off, const int *blk_img,
                  const long long *seg_off, const long long *blk_line_base, const unsigned *base_bl,
                  const unsigned *wave_count, const long long *tri_off, const CRec *st_r, const double *st_unc,
                  const unsigned *st_key, CRec *cand, double *cand_unc, unsigned *cand_node, const long long *group_base,
                  unsigned *perm, const unsigned *blk_surv, const unsigned *blk_rnd0, const unsigned *round_count) {
  if (n_blk <= 0 || max_rows <= 0) return;
  if (round_count) {  // the line-slot form: candidate lists per round of 64 survivors
    static int n_cu = 0;
    if (n_cu == 0) {
      int dev = 0;
      if (hipGetDevice(&dev) != hipSuccess ||
          hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu <= 0)
        n_cu = 256;
    }
    hipLaunchKernelGGL(k_place_rounds, dim3((unsigned)std::min<long long>(n_blk, (long long)n_cu * LT_PLACE_WG_PER_CU)), dim3(256), 0, st, m_off,
                       blk_img, seg_off, blk_line_base, base_bl, tri_off, st_r, st_unc, st_key, cand, cand_unc, cand_node, perm,
                       blk_surv, blk_rnd0, round_count, n_blk);
    return;
  }
  const int n_groups = gen_groups(max_rows);
  hipLaunchKernelGGL(k_place<kTriSlots>, dim3(nblk2(n_groups, 4), n_blk), dim3(256), 0, st, m_off, blk_img, seg_off,
                     blk_line_base, base_bl, wave_count, tri_off, st_r, st_unc, st_key, cand, cand_unc, cand_node, n_groups,
                     group_base, perm, (const unsigned *)nullptr);
}
void launch_pack_keys(hipStream_t st, int n_blk, long long max_rows, const long long *m_off,
                      const unsigned *wave_count, const long long *wave_pos, const unsigned *st_key,
                      unsigned *keys_c, unsigned *src_c, const long long *group_base) {
  if (n_blk <= 0 || max_rows <= 0) return;
  const int n_groups = gen_groups(max_rows);
  hipLaunchKernelGGL(k_pack_keys, dim3(nblk2(n_groups, 4), n_blk), dim3(256), 0, st, m_off, wave_count, wave_pos,
                     st_key, keys_c, src_c, n_groups, group_base);
}
void launch_permute(hipStream_t st, long long C, const unsigned *skeys, const unsigned *ssrc, const CRec *st_r,
                    const double *st_unc, CRec *cand, double *cand_unc, unsigned *cand_node) {
  if (C > 0)
    hipLaunchKernelGGL(k_permute, dim3(nblk2(C, 256)), dim3(256), 0, st, C, skeys, ssrc, st_r, st_unc, cand, cand_unc,
                       cand_node);
}
void launch_host_view(hipStream_t st, long long C, const unsigned *perm, const CRec *rec, const double *unc, Cand *out_c,
                      CandLite *out_l) {
  if (C > 0) hipLaunchKernelGGL(k_host_view, dim3(nblk2(C, 256)), dim3(256), 0, st, C, perm, rec, unc, out_c, out_l);
}
}  // namespace lt
