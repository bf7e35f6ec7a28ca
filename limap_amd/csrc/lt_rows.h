// lt_rows.h -- the one host pass over a block of match rows (host side of TriangulateImage / TriangulateAll).
//
// A block is the (n, 2) int32 matrix the reference receives per (image, neighbour) (`matches[ng_img_id]`,
// base_line_triangulator.cc:82-98): column 0 = line id in the image, column 1 = line id in the neighbour.  The pass
//   * validates it as reductions: the largest id of either column taken as unsigned (a negative id wraps to a huge
//     value; the caller compares against the line counts and raises the reference's IndexError, :87-94) and whether
//     any line id is smaller than its predecessor (k_gates relies on runs of equal line ids only when sorted),
//   * writes the staged copy PACKED to one word per row, line | neighbour line << 16 (both below 65536 for every row
//     that passes the validation, util/types.h:16): half the bytes over PCIe and for k_gates.
// 80 MB of rows per 100 images make this pass the largest host cost of the call sequence; the baseline x86-64 code of
// the build (no -march) has no unsigned 32-bit max and packs one row at a time, so the pass is written for AVX-512 and
// AVX2 explicitly and dispatched once per process (4-5 GB/s per core before, memory speed now).
#pragma once

#include <cstdint>
#include <immintrin.h>

namespace lt {

struct RowStats {
  unsigned mx_line = 0, mx_ng = 0;
  int unsorted = 0;
  int irregular = 0;  // pack_rows_cb: a line step other than 0 / +1 (the block cannot take the compressed form)
};

inline void pack_rows_scalar(const int32_t *src, long long r0, long long n, unsigned *o, RowStats &s) {
  for (long long r = r0; r < n; ++r) {
    const unsigned line = (unsigned)src[2 * r], ng = (unsigned)src[2 * r + 1];
    s.mx_line = line > s.mx_line ? line : s.mx_line;
    s.mx_ng = ng > s.mx_ng ? ng : s.mx_ng;
    s.unsorted |= (r > 0 && src[2 * r] < src[2 * r - 2]) ? 1 : 0;
    o[r] = (line & 0xFFFFu) | (ng << 16);
  }
}

// A row is one 64-bit lane x = line | ng << 32.  Packed word = (x & 0xFFFF) | ((x >> 16) & 0xFFFF0000), taken from
// the low half of the lane.  The column maxima are one unsigned max over the interleaved 32-bit lanes (even lanes:
// lines, odd lanes: neighbour lines); sortedness compares every vector with the one loaded one row earlier (even
// lanes count).
__attribute__((target("avx2"))) inline void pack_rows_avx2(const int32_t *src, long long n, unsigned *o, RowStats &s) {
  long long r = 0;
  if (n > 0) {
    pack_rows_scalar(src, 0, 1, o, s);  // row 0 has no predecessor
    r = 1;
  }
  const __m256i lo16 = _mm256_set1_epi64x(0xFFFFll), hi16 = _mm256_set1_epi64x(0xFFFF0000ll);
  const __m256i even = _mm256_setr_epi32(0, 2, 4, 6, 0, 2, 4, 6);
  __m256i vmax = _mm256_setzero_si256(), vuns = _mm256_setzero_si256();
  for (; r + 8 <= n; r += 8) {
    const __m256i a = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(src + 2 * r));
    const __m256i b = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(src + 2 * r + 8));
    const __m256i pa = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(src + 2 * r - 2));
    const __m256i pb = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(src + 2 * r + 6));
    vmax = _mm256_max_epu32(vmax, _mm256_max_epu32(a, b));
    vuns = _mm256_or_si256(vuns, _mm256_or_si256(_mm256_cmpgt_epi32(pa, a), _mm256_cmpgt_epi32(pb, b)));
    const __m256i ta = _mm256_or_si256(_mm256_and_si256(a, lo16), _mm256_and_si256(_mm256_srli_epi64(a, 16), hi16));
    const __m256i tb = _mm256_or_si256(_mm256_and_si256(b, lo16), _mm256_and_si256(_mm256_srli_epi64(b, 16), hi16));
    const __m256i qa = _mm256_permutevar8x32_epi32(ta, even), qb = _mm256_permutevar8x32_epi32(tb, even);
    _mm256_storeu_si256(reinterpret_cast<__m256i *>(o + r), _mm256_permute2x128_si256(qa, qb, 0x20));
  }
  alignas(32) unsigned m[8];
  _mm256_store_si256(reinterpret_cast<__m256i *>(m), vmax);
  for (int k = 0; k < 8; k += 2) {
    s.mx_line = m[k] > s.mx_line ? m[k] : s.mx_line;
    s.mx_ng = m[k + 1] > s.mx_ng ? m[k + 1] : s.mx_ng;
  }
  if (_mm256_movemask_ps(_mm256_castsi256_ps(vuns)) & 0x55) s.unsorted = 1;
  pack_rows_scalar(src, r, n, o, s);
}

__attribute__((target("avx512f"))) inline void pack_rows_avx512(const int32_t *src, long long n, unsigned *o, RowStats &s) {
  long long r = 0;
  if (n > 0) {
    pack_rows_scalar(src, 0, 1, o, s);
    r = 1;
  }
  const __m512i lo16 = _mm512_set1_epi64(0xFFFFll), hi16 = _mm512_set1_epi64(0xFFFF0000ll);
  __m512i vmax = _mm512_setzero_si512();
  unsigned uns = 0;
  for (; r + 16 <= n; r += 16) {
    const __m512i a = _mm512_loadu_si512(src + 2 * r), b = _mm512_loadu_si512(src + 2 * r + 16);
    const __m512i pa = _mm512_loadu_si512(src + 2 * r - 2), pb = _mm512_loadu_si512(src + 2 * r + 14);
    vmax = _mm512_max_epu32(vmax, _mm512_max_epu32(a, b));
    uns |= (unsigned)_mm512_cmpgt_epi32_mask(pa, a) | (unsigned)_mm512_cmpgt_epi32_mask(pb, b);
    const __m512i ta = _mm512_or_si512(_mm512_and_si512(a, lo16), _mm512_and_si512(_mm512_srli_epi64(a, 16), hi16));
    const __m512i tb = _mm512_or_si512(_mm512_and_si512(b, lo16), _mm512_and_si512(_mm512_srli_epi64(b, 16), hi16));
    _mm256_storeu_si256(reinterpret_cast<__m256i *>(o + r), _mm512_cvtepi64_epi32(ta));
    _mm256_storeu_si256(reinterpret_cast<__m256i *>(o + r + 8), _mm512_cvtepi64_epi32(tb));
  }
  alignas(64) unsigned m[16];
  _mm512_store_si512(m, vmax);
  for (int k = 0; k < 16; k += 2) {
    s.mx_line = m[k] > s.mx_line ? m[k] : s.mx_line;
    s.mx_ng = m[k + 1] > s.mx_ng ? m[k + 1] : s.mx_ng;
  }
  if (uns & 0x5555u) s.unsorted = 1;
  pack_rows_scalar(src, r, n, o, s);
}

// ---------------------------------------------------------------------------------------------
// Compressed block format (round 4): 17 bits per row instead of 32 over PCIe.
// What limap's matchers write is sorted by line id with steps of 0 or 1 (every line of the image has its rows, top-k of
// them; base_line_triangulator.cc:82-98 iterates them in that order).  For such a block the line column is one BIT per
// row -- "a new line starts here" -- plus the first row's line id:
//   words [0, nbw), nbw = (n + 1) / 2 rounded up to even   the neighbour line of every row, 16 bits each (row 2 k in the low half)
//   words [nbw, nbw + 2 ceil(n / 64))                      bit r % 64 of 64-bit word r / 64 = line[r] != line[r - 1]  (row 0: 0)
// (an even number of words: blocks and bit words stay 8-byte aligned)
// cb_words(n) depends on n only, so the blocks of a call get their places before the pass runs.  A block that is not of
// that shape (a step other than 0 / 1: unsorted rows, a line without rows) is reported `irregular`; the caller stores it
// in the plain one-word-per-row form instead.  k_expand_rows (lt_kernels_v2.hip) rebuilds line | neighbour line << 16 on
// the device, where k_gates reads it.
// ---------------------------------------------------------------------------------------------
inline size_t cb_nb_words(long long n) { return (size_t)((((n + 1) / 2) + 1) & ~1ll); }
inline size_t cb_words(long long n) { return cb_nb_words(n) + (size_t)(2 * ((n + 63) / 64)); }

inline void pack_cb_scalar(const int32_t *src, long long r0, long long n, uint16_t *nb, uint64_t *bits, RowStats &s) {
  for (long long r = r0; r < n; ++r) {
    const unsigned line = (unsigned)src[2 * r], ng = (unsigned)src[2 * r + 1];
    s.mx_line = line > s.mx_line ? line : s.mx_line;
    s.mx_ng = ng > s.mx_ng ? ng : s.mx_ng;
    if (r > 0) {
      const unsigned d = line - (unsigned)src[2 * r - 2];
      s.unsorted |= (src[2 * r] < src[2 * r - 2]) ? 1 : 0;
      s.irregular |= d > 1u ? 1 : 0;
      if (d) bits[r >> 6] |= 1ull << (r & 63);
    }
    nb[r] = (uint16_t)ng;
  }
}

__attribute__((target("avx512f,bmi2"))) inline void pack_cb_avx512(const int32_t *src, long long n, uint16_t *nb, uint64_t *bits,
                                                                   RowStats &s) {
  long long r = 0;
  if (n > 0) {
    pack_cb_scalar(src, 0, 1, nb, bits, s);  // row 0 has no predecessor
    r = 1;
  }
  const __m512i one = _mm512_set1_epi32(1);
  __m512i vmax = _mm512_setzero_si512();
  unsigned uns = 0, irr = 0;
  for (; r + 16 <= n; r += 16) {
    const __m512i a = _mm512_loadu_si512(src + 2 * r), b = _mm512_loadu_si512(src + 2 * r + 16);
    const __m512i pa = _mm512_loadu_si512(src + 2 * r - 2), pb = _mm512_loadu_si512(src + 2 * r + 14);
    vmax = _mm512_max_epu32(vmax, _mm512_max_epu32(a, b));
    uns |= (unsigned)_mm512_cmpgt_epi32_mask(pa, a) | (unsigned)_mm512_cmpgt_epi32_mask(pb, b);
    irr |= (unsigned)_mm512_cmpgt_epu32_mask(_mm512_sub_epi32(a, pa), one) | (unsigned)_mm512_cmpgt_epu32_mask(_mm512_sub_epi32(b, pb), one);
    // even 32-bit lanes are the line ids: one bit per row out of the 16-lane inequality masks
    const unsigned na = _pext_u32((unsigned)_mm512_cmpneq_epi32_mask(a, pa), 0x5555u);
    const unsigned nb_ = _pext_u32((unsigned)_mm512_cmpneq_epi32_mask(b, pb), 0x5555u);
    const uint64_t w = (uint64_t)(na | (nb_ << 8));
    const unsigned sh = (unsigned)(r & 63);
    bits[r >> 6] |= w << sh;
    if (sh > 48) bits[(r >> 6) + 1] |= w >> (64 - sh);
    // neighbour lines: the high half of every 64-bit lane, narrowed to 16 bits
    _mm_storeu_si128(reinterpret_cast<__m128i *>(nb + r), _mm512_cvtepi64_epi16(_mm512_srli_epi64(a, 32)));
    _mm_storeu_si128(reinterpret_cast<__m128i *>(nb + r + 8), _mm512_cvtepi64_epi16(_mm512_srli_epi64(b, 32)));
  }
  alignas(64) unsigned m[16];
  _mm512_store_si512(m, vmax);
  for (int k = 0; k < 16; k += 2) {
    s.mx_line = m[k] > s.mx_line ? m[k] : s.mx_line;
    s.mx_ng = m[k + 1] > s.mx_ng ? m[k + 1] : s.mx_ng;
  }
  if (uns & 0x5555u) s.unsorted = 1;
  if (irr & 0x5555u) s.irregular = 1;
  pack_cb_scalar(src, r, n, nb, bits, s);
}

// out: cb_words(n) words (the bit words are cleared here).  level as for pack_rows (2 = AVX2 takes the scalar form).
inline RowStats pack_rows_cb(const int32_t *src, long long n, unsigned *out, int level = 0) {
  static const int best = (__builtin_cpu_supports("avx512f") && __builtin_cpu_supports("bmi2")) ? 3 : 1;
  if (level <= 0 || level > best) level = best;
  RowStats s;
  uint16_t *nb = reinterpret_cast<uint16_t *>(out);
  uint64_t *bits = reinterpret_cast<uint64_t *>(out + cb_nb_words(n));  // 8-byte aligned whenever `out` is
  const size_t nw = (size_t)((n + 63) / 64);
  for (size_t k = 0; k < nw; ++k) bits[k] = 0ull;
  for (size_t k = (size_t)n; k < 2 * cb_nb_words(n); ++k) nb[k] = 0;  // the unused halves behind the last row
  if (level == 3) pack_cb_avx512(src, n, nb, bits, s);
  else pack_cb_scalar(src, 0, n, nb, bits, s);
  return s;
}

// level: 0 = the best the CPU has, 1 = scalar, 2 = AVX2, 3 = AVX-512 (the explicit levels are for the tests)
inline RowStats pack_rows(const int32_t *src, long long n, unsigned *o, int level = 0) {
  static const int best = __builtin_cpu_supports("avx512f") ? 3 : (__builtin_cpu_supports("avx2") ? 2 : 1);
  if (level <= 0 || level > best) level = best;
  RowStats s;
  if (level == 3) pack_rows_avx512(src, n, o, s);
  else if (level == 2) pack_rows_avx2(src, n, o, s);
  else pack_rows_scalar(src, 0, n, o, s);
  return s;
}

}  // namespace lt
