// lt_devfn.h -- device functions shared by the kernel translation units (generation of one
// candidate, dense evaluation of one scoring pair, wave helpers).  See lt_kernels.hip for the map.
#pragma once

#include "lt_device.h"

namespace lt {

static __device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
static __device__ __forceinline__ unsigned long long lanemask_lt() {
  return (1ull << lane_id()) - 1ull;
}
// Wave-wide reductions on the VALU's DPP path (row shifts + row broadcasts, result read from lane 63 into a scalar
// register): a butterfly of __shfl_xor is six DEPENDENT ds_bpermute round trips through the LDS pipe (~100+ cycles
// each, more when other waves keep that pipe busy); this is a dozen VALU instructions.  Idempotent operators only
// (lanes without a DPP source combine with themselves).  All 64 lanes must be active.
template <class Op>
static __device__ __forceinline__ int wave_reduce_dpp(int v, Op op) {
  v = op(v, __builtin_amdgcn_update_dpp(v, v, 0x111, 0xF, 0xF, false));  // row_shr:1
  v = op(v, __builtin_amdgcn_update_dpp(v, v, 0x112, 0xF, 0xF, false));  // row_shr:2
  v = op(v, __builtin_amdgcn_update_dpp(v, v, 0x114, 0xF, 0xF, false));  // row_shr:4
  v = op(v, __builtin_amdgcn_update_dpp(v, v, 0x118, 0xF, 0xF, false));  // row_shr:8  -> lane 15 of every row
  v = op(v, __builtin_amdgcn_update_dpp(v, v, 0x142, 0xA, 0xF, false));  // row_bcast:15 into rows 1 and 3
  v = op(v, __builtin_amdgcn_update_dpp(v, v, 0x143, 0xC, 0xF, false));  // row_bcast:31 into rows 2 and 3
  return __builtin_amdgcn_readlane(v, 63);
}
// Inclusive prefix sum over the wave on the same path (no LDS round trips): shifts inside the rows of 16 lanes with zero
// fill, then the last lane of row 0 / 2 into rows 1 / 3 and lane 31 into rows 2 and 3.  All 64 lanes must be active.
static __device__ __forceinline__ unsigned wave_incl_scan_u32(unsigned x) {
  int v = (int)x;
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, true);  // row_shr:1
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, true);  // row_shr:2
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, true);  // row_shr:4
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, true);  // row_shr:8
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, true);  // row_bcast:15 -> rows 1 and 3
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, true);  // row_bcast:31 -> rows 2 and 3
  return (unsigned)v;
}
static __device__ __forceinline__ int wave_max_i32(int v) {
  return wave_reduce_dpp(v, [](int a, int b) { return a > b ? a : b; });
}
static __device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
  return (unsigned)wave_reduce_dpp((int)v, [](int a, int b) { return (unsigned)a > (unsigned)b ? a : b; });
}
static __device__ __forceinline__ unsigned wave_min_u32(unsigned v) {
  return (unsigned)wave_reduce_dpp((int)v, [](int a, int b) { return (unsigned)a < (unsigned)b ? a : b; });
}
// maximum of floats in which a NaN anywhere wins (the sweep's window radius: NaN -> no pruning)
static __device__ __forceinline__ float wave_max_f32_nan(float v) {
  return __int_as_float(wave_reduce_dpp(__float_as_int(v), [](int a, int b) {
    const float x = __int_as_float(a), y = __int_as_float(b);
    return (y > x || y != y) ? b : a;
  }));
}
static __device__ __forceinline__ double readlane_f64(double v, int src) {  // src: wave-uniform
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}

// Lanes of ONE wave exchanging data through LDS.  DS operations of a wave execute in order, so all
// that is needed is (a) the compiler must not move LDS accesses across this point (asm memory clobber)
// and (b) earlier DS results must have landed (lgkmcnt(0)).  Deliberately NOT a fence: an acq_rel
// fence also drains vmcnt, i.e. waits for every outstanding global store of the wave.
static __device__ __forceinline__ void wave_lds_sync() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}


struct GenOut {
  CRec r;      // nb_slot / ng_line are filled in by the caller
  double unc;  // min of the two views' uncertainties (side array of the candidate store)
};

// triangulate_point, functions.cc:100-117 (2x2 LDLT solve with diagonal pivoting)
static __device__ __forceinline__ bool tri_point(const Cam &c1, d3 r1, const Cam &c2, d3 r2, d3 *out) {
  d3 C1 = cam_center(c1), C2 = cam_center(c2);
  double a00 = dot(r1, r1), a10 = -dot(r2, r1), a11 = dot(r2, r2);
  double b0 = dot(r1, sub(C2, C1));
  double b1 = dot(r2, sub(C1, C2));
  bool sw = fabs(a11) > fabs(a00);
  double dd0 = sw ? a11 : a00, dd1 = sw ? a00 : a11;
  double q0 = sw ? b1 : b0, q1 = sw ? b0 : b1;
  double l10 = a10 / dd0;
  double s1d = dd1 - l10 * (dd0 * l10);
  double y1 = q1 - l10 * q0;
  double z0 = q0 / dd0, z1 = y1 / s1d;
  double s0 = z0 - l10 * z1;
  double x0 = sw ? z1 : s0, x1 = sw ? s0 : z1;
  d3 p = add(add(add(scale(r1, x0), C1), scale(r2, x1)), C2);
  p = d3{0.5 * p.x, 0.5 * p.y, 0.5 * p.z};
  if (cam_depth(c1, p) < kEps || cam_depth(c2, p) < kEps) return false;
  *out = p;
  return true;
}

// Line3d::sensitivity, linebase.cc:100-107
static __device__ __forceinline__ double sensitivity(const Cam &c, d3 s, d3 e, d3 dir3) {
  d2 ps = cam_project(c, s), pe = cam_project(c, e);
  d2 mid = d2{0.5 * (ps.x + pe.x), 0.5 * (ps.y + pe.y)};
  d3 ray = cam_ray(c, mid);
  double cv = fabs(dot(dir3, ray));
  return 90 - acos(cv) * 180.0 / kPi;
}

// `line.sensitivity(view) > sens_th` decided in the cosine domain where that is safe (cfg.sens_lo / sens_hi,
// see make_gen), by the exact expression otherwise: saves the two acos per triangulated connection
static __device__ __forceinline__ bool sensitivity_gt(const GenCfg &cfg, const Cam &c, d3 s, d3 e, d3 dir3) {
  d2 ps = cam_project(c, s), pe = cam_project(c, e);
  d2 mid = d2{0.5 * (ps.x + pe.x), 0.5 * (ps.y + pe.y)};
  d3 ray = cam_ray(c, mid);
  double cv = fabs(dot(dir3, ray));
  if (cv > cfg.sens_hi) return true;
  if (cv < cfg.sens_lo) return false;
  return 90 - acos(cv) * 180.0 / kPi > cfg.sens_th;
}

// compute_epipolar_IoU (functions.cc:76-98) with the fundamental matrix hoisted per image pair
static __device__ __forceinline__ double epipolar_iou(const Seg &s1, const Seg &s2, const double *F) {
  L2 l2{mk2(s2.x1, s2.y1), mk2(s2.x2, s2.y2)};
  double ln2 = len(l2);
  d3 lc2 = mk3(s2.lc[0], s2.lc[1], s2.lc[2]);
  d3 eps = unit(mv(F, mk3(s1.x1, s1.y1, 1.0)));
  d3 hs = cross(lc2, eps);
  double zs = hs.z + kEps;
  d2 cs = d2{hs.x / zs, hs.y / zs};
  d3 epe = unit(mv(F, mk3(s1.x2, s1.y2, 1.0)));
  d3 he = cross(lc2, epe);
  double ze = he.z + kEps;
  d2 ce = d2{he.x / ze, he.y / ze};
  d2 dv = dir(l2);
  double c1v = dot(sub(cs, l2.s), dv) / ln2;
  double c2v = dot(sub(ce, l2.s), dv) / ln2;
  if (c1v > c2v) {
    double t = c1v; c1v = c2v; c2v = t;
  }
  return (dmin(c2v, 1.0) - dmax(c1v, 0.0)) / (dmax(c2v, 1.0) - dmin(c1v, 0.0));
}

// line_triangulation (functions.cc:194-233): x = (A^-1 B)[0], A = [c1 | -c2s | -c2e]
static __device__ __forceinline__ bool tri_line(const Cam &c1, const Cam &c2, const Seg &s1, const Seg &s2,
                                                const double *Bv, d3 *ps_o, d3 *pe_o, double *z_start,
                                                double *z_end, double *d21, double *d22) {
  d3 r1s = mk3(s1.rs[0], s1.rs[1], s1.rs[2]), r1e = mk3(s1.re[0], s1.re[1], s1.re[2]);
  d3 c2s = mk3(s2.rs[0], s2.rs[1], s2.rs[2]), c2e = mk3(s2.re[0], s2.re[1], s2.re[2]);
  d3 u = mk3(-c2s.x, -c2s.y, -c2s.z), v = mk3(-c2e.x, -c2e.y, -c2e.z);
  // cofactors (j,0) of A do not involve column 0: shared by the start and the end solve
  double k0 = u.y * v.z - v.y * u.z;
  double k1 = u.z * v.x - v.z * u.x;
  double k2 = u.x * v.y - v.x * u.y;
  d3 B = mk3(Bv[0], Bv[1], Bv[2]);
  d3 C1 = cam_center(c1);
  d3 ps, pe;
  {
    double det = (k0 * r1s.x + k1 * r1s.y) + k2 * r1s.z;
    double id = 1.0 / det;
    double x0 = ((k0 * id) * B.x + (k1 * id) * B.y) + (k2 * id) * B.z;
    ps = add(scale(r1s, x0), C1);
    *z_start = cam_depth(c1, ps);
  }
  {
    double det = (k0 * r1e.x + k1 * r1e.y) + k2 * r1e.z;
    double id = 1.0 / det;
    double x0 = ((k0 * id) * B.x + (k1 * id) * B.y) + (k2 * id) * B.z;
    pe = add(scale(r1e, x0), C1);
    *z_end = cam_depth(c1, pe);
  }
  *ps_o = ps;
  *pe_o = pe;
  if (*z_start < kEps || *z_end < kEps) return false;
  *d21 = cam_depth(c2, ps);
  *d22 = cam_depth(c2, pe);
  if (*d21 < kEps || *d22 < kEps) return false;
  if (isnan(ps.x) || isnan(pe.x)) return false;
  return true;
}

// Stage A of triangulateOneNode: the cheap gates that reject most connections
// (base_line_triangulator.cc:166,177,293-307).  Needs only the two segment records and F.
static __device__ __forceinline__ bool gen_gates(const GenCfg &cfg, const Seg &s1, const Seg &s2,
                                                 const double *F) {
  L2 l1{mk2(s1.x1, s1.y1), mk2(s1.x2, s1.y2)};
  L2 l2{mk2(s2.x1, s2.y1), mk2(s2.x2, s2.y2)};
  if (len(l1) <= cfg.min_length_2d) return false;  // base_line_triangulator.cc:166
  if (len(l2) <= cfg.min_length_2d) return false;  // :177
  if (cfg.disable_algebraic) return false;
  // degeneracy by ray-plane angles (:293-302).  angle = 90 - acos(a) 180/pi < th  <=>  a < sin(th)
  // up to libm rounding: outside the [sin_lo, sin_hi] band the cosine alone decides.
  d3 n2 = mk3(s2.n[0], s2.n[1], s2.n[2]);
  d3 r1s = mk3(s1.rs[0], s1.rs[1], s1.rs[2]), r1e = mk3(s1.re[0], s1.re[1], s1.re[2]);
  double as = fabs(dot(n2, r1s));
  if (as < cfg.sin_lo) return false;
  if (!(as > cfg.sin_hi)) {
    double ang = 90 - acos(as) * 180.0 / kPi;
    if (ang < cfg.angle_th) return false;
  }
  double ae = fabs(dot(n2, r1e));
  if (ae < cfg.sin_lo) return false;
  if (!(ae > cfg.sin_hi)) {
    double ang = 90 - acos(ae) * 180.0 / kPi;
    if (ang < cfg.angle_th) return false;
  }
  // weak epipolar constraint (:305-307)
  double iou = epipolar_iou(s1, s2, F);
  if (iou < cfg.iou_th) return false;
  return true;
}

// 1 / x to full double precision without the IEEE division sequence (v_rcp_f64 + two Newton steps).
// Only used inside conservative decisions that carry their own error margin.
static __device__ __forceinline__ double fast_rcp(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
  r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
  return r;
}

// Three-way form of stage A for the split pipeline (k_gates + k_tri_rows): 0 = the reference
// certainly skips this connection, 1 = it certainly passes every gate, 2 = too close to a threshold
// (or badly conditioned) for the cheap arithmetic -- the exact gates (gen_gates) then decide in the
// triangulation kernel.  The reference computes IoU = (min(c2,1) - max(c1,0)) / (max(c2,1) - min(c1,0))
// from two normalised epipolar lines and rejects IoU < th; only the DECISION is needed here and the
// denominator is >= 1, so the test is num - th * den < 0.  Nothing transcendental or exact is
// evaluated: no acos, no normalisations, sqrt only in float (it enters D at the 1e-12 level), a
// reciprocal instead of the divisions, FMAs, no branches.  This is NOT reference arithmetic -- it
// only ever returns 0/1 when the reference's outcome is certain, with these error budgets:
//   * lengths: q vs min_length^2 (1 +- 1e-12)                  [exact test: sqrt(q) <= min_length]
//   * ray/plane angles: |cos| vs sin(th) (1 +- 1e-7)           [exact: 90 - acos(.) 180/pi < th]
//   * IoU: delta = num - th den vs a margin of 1e-7 (1 + |c1| + |c2|)(1 + |th|) plus 1e-12 x the
//     cancellation bound of the numerators, and only if D has no cancellation beyond 1e4 (`well`).
// The neighbour-side operands come from the per-segment SegGate record (segment-only invariants of
// the algebra c = (N.xy . v - D s2 . v) / (D |v|^2), N = lc2 x (F x~), D = N.z + eps |F x~|):
//   N.xy . v = az w1 + ax P + ay Q   with  w1 = lcy vx - lcx vy, P = lcz vy, Q = -lcz vx, v = e2 - s2.
struct SegGate {
  double n[3];      // plane normal of the segment's back-projection (Seg::n)
  double lcx, lcy;  // first two line coordinates (Seg::lc)
  double P, Q, w1;  // see above
  double sv;        // s2 . v
  double q2;        // |v|^2
};
static_assert(sizeof(SegGate) == 80, "SegGate layout");

static __device__ __forceinline__ void seg_gate_build(const Seg &s, SegGate *g) {
  const double vx = s.x2 - s.x1, vy = s.y2 - s.y1;
  g->n[0] = s.n[0]; g->n[1] = s.n[1]; g->n[2] = s.n[2];
  g->lcx = s.lc[0]; g->lcy = s.lc[1];
  g->P = s.lc[2] * vy;
  g->Q = -(s.lc[2] * vx);
  g->w1 = s.lc[1] * vx - s.lc[0] * vy;
  g->sv = s.x1 * vx + s.y1 * vy;
  g->q2 = vx * vx + vy * vy;  // == (x1-x2)^2 + (y1-y2)^2 bit for bit (negation is exact)
}

// The view-1 side of the IoU algebra: epipolar line of one endpoint x of l1 (a = F x~), |a|^2 and |a| (float sqrt,
// see above).  Depends on (l1, image pair) only -- the exhaustive kernel evaluates it once per node and neighbour
// image instead of once per connection.
struct GateEpi {
  double ax, ay, az, n2a, na;
};
static __device__ __forceinline__ GateEpi gate3_epi(const double *F, double px, double py) {
  GateEpi e;
  e.ax = __builtin_fma(F[0], px, __builtin_fma(F[1], py, F[2]));
  e.ay = __builtin_fma(F[3], px, __builtin_fma(F[4], py, F[5]));
  e.az = __builtin_fma(F[6], px, __builtin_fma(F[7], py, F[8]));
  e.n2a = __builtin_fma(e.ax, e.ax, __builtin_fma(e.ay, e.ay, e.az * e.az));
  e.na = (double)__builtin_amdgcn_sqrtf((float)e.n2a);
  return e;
}

// q1 = squared length of l1
static __device__ __forceinline__ int gate3_core(const GenCfg &cfg, double q1, double rs1x, double rs1y, double rs1z,
                                                 double re1x, double re1y, double re1z, double n2x, double n2y,
                                                 double n2z, double lcx, double lcy, double P, double Q, double w1,
                                                 double sv, double q2, const GateEpi &ea, const GateEpi &eb) {
  bool rej = (q1 <= cfg.len_lo2) | (q2 <= cfg.len_lo2) | (cfg.disable_algebraic != 0);
  bool und = !(q1 > cfg.len_hi2) | !(q2 > cfg.len_hi2);
  const double as = fabs(__builtin_fma(n2x, rs1x, __builtin_fma(n2y, rs1y, n2z * rs1z)));
  const double ae = fabs(__builtin_fma(n2x, re1x, __builtin_fma(n2y, re1y, n2z * re1z)));
  rej |= (as < cfg.sin_lo) | (ae < cfg.sin_lo);
  und |= !(as > cfg.sin_hi) | !(ae > cfg.sin_hi);
  double cv[2], ce[2];
  bool well = q2 > 0.0;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const GateEpi &e = k == 0 ? ea : eb;
    const double ax = e.ax, ay = e.ay, az = e.az, n2a = e.n2a, na = e.na;
    const double t1 = lcx * ay, t2 = lcy * ax;
    const double D = __builtin_fma(kEps, na, t1 - t2);
    const double m0 = az * w1, m1 = ax * P, m2 = ay * Q, m3 = D * sv;
    const double numer = ((m0 + m1) + m2) - m3;
    const double Dq = D * q2;
    const double r = fast_rcp(Dq);
    // conditioning: no cancellation in D beyond 1e4, the float sqrt stays far below the margin, no
    // overflow / underflow games
    well = well & (fabs(D) > 1e-4 * (fabs(t1) + fabs(t2))) & (fabs(D) > 1e-9 * na) & (n2a > 1e-30) &
           (n2a < 1e30) & (fabs(Dq) > 1e-280) & (fabs(Dq) < 1e280);
    cv[k] = numer * r;
    ce[k] = (((fabs(m0) + fabs(m1)) + fabs(m2)) + fabs(m3)) * fabs(r);
  }
  const double cerr = ce[0] + ce[1];
  // v_min/v_max drop NaN operands: harmless here, a NaN can only come from an infinite term of
  // `numer`, which makes cerr (hence the margin) infinite or NaN and the outcome "undecided"
  const double c1v = __builtin_fmin(cv[0], cv[1]), c2v = __builtin_fmax(cv[0], cv[1]);
  const double num = __builtin_fmin(c2v, 1.0) - __builtin_fmax(c1v, 0.0);
  const double den = __builtin_fmax(c2v, 1.0) - __builtin_fmin(c1v, 0.0);
  const double delta = num - cfg.iou_th * den;
  const double margin = __builtin_fma(1e-12, cerr, 1e-7 * (1.0 + fabs(c1v) + fabs(c2v)) * (1.0 + fabs(cfg.iou_th)));
  rej |= well & (delta < -margin);
  und |= !(well & (delta > margin));
  if (cfg.force_undecided) return 2;  // test switch (LT_TEST_NO_FAST_GATES)
  return rej ? 0 : (und ? 2 : 1);
}

// The same decision with fewer instructions (k_gates_ln): the numerator and its cancellation bound as FMA chains (this is
// not reference arithmetic -- see gate3_core -- and a fused product only tightens the roundings the margin covers).
static __device__ __forceinline__ int gate3_core_fma(const GenCfg &cfg, double q1, double rs1x, double rs1y, double rs1z,
                                                     double re1x, double re1y, double re1z, double n2x, double n2y,
                                                     double n2z, double lcx, double lcy, double P, double Q, double w1,
                                                     double sv, double q2, const GateEpi &ea, const GateEpi &eb) {
  bool rej = (q1 <= cfg.len_lo2) | (q2 <= cfg.len_lo2) | (cfg.disable_algebraic != 0);
  bool und = !(q1 > cfg.len_hi2) | !(q2 > cfg.len_hi2);
  const double as = fabs(__builtin_fma(n2x, rs1x, __builtin_fma(n2y, rs1y, n2z * rs1z)));
  const double ae = fabs(__builtin_fma(n2x, re1x, __builtin_fma(n2y, re1y, n2z * re1z)));
  rej |= (as < cfg.sin_lo) | (ae < cfg.sin_lo);
  und |= !(as > cfg.sin_hi) | !(ae > cfg.sin_hi);
  double cv[2], ce[2];
  bool well = q2 > 0.0;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const GateEpi &e = k == 0 ? ea : eb;
    const double ax = e.ax, ay = e.ay, az = e.az, n2a = e.n2a, na = e.na;
    const double t2 = lcy * ax;
    const double D = __builtin_fma(kEps, na, __builtin_fma(lcx, ay, -t2));
    const double tmag = __builtin_fma(fabs(lcx), fabs(ay), fabs(t2));  // |t1| + |t2|
    const double m3 = D * sv;
    const double numer = __builtin_fma(az, w1, __builtin_fma(ax, P, __builtin_fma(ay, Q, -m3)));
    const double mag = __builtin_fma(fabs(az), fabs(w1), __builtin_fma(fabs(ax), fabs(P), __builtin_fma(fabs(ay), fabs(Q), fabs(m3))));
    const double Dq = D * q2;
    // v_rcp_f64 is good to ~2^-23; one Newton step (2^-46) is far below the 1e-7 margin of the decision
    double r = __builtin_amdgcn_rcp(Dq);
    r = __builtin_fma(__builtin_fma(-Dq, r, 1.0), r, r);
    well = well & (fabs(D) > 1e-4 * tmag) & (fabs(D) > 1e-9 * na) & (n2a > 1e-30) & (n2a < 1e30) & (fabs(Dq) > 1e-280) &
           (fabs(Dq) < 1e280);
    cv[k] = numer * r;
    ce[k] = mag * fabs(r);
  }
  const double cerr = ce[0] + ce[1];
  const double c1v = __builtin_fmin(cv[0], cv[1]), c2v = __builtin_fmax(cv[0], cv[1]);
  const double num = __builtin_fmin(c2v, 1.0) - __builtin_fmax(c1v, 0.0);
  const double den = __builtin_fmax(c2v, 1.0) - __builtin_fmin(c1v, 0.0);
  const double delta = num - cfg.iou_th * den;
  const double margin = __builtin_fma(1e-12, cerr, 1e-7 * (1.0 + fabs(c1v) + fabs(c2v)) * (1.0 + fabs(cfg.iou_th)));
  rej |= well & (delta < -margin);
  und |= !(well & (delta > margin));
  if (cfg.force_undecided) return 2;  // test switch (LT_TEST_NO_FAST_GATES)
  return rej ? 0 : (und ? 2 : 1);
}

static __device__ __forceinline__ int gate3(const GenCfg &cfg, double a1x, double a1y, double b1x, double b1y,
                                            double rs1x, double rs1y, double rs1z, double re1x, double re1y,
                                            double re1z, double n2x, double n2y, double n2z, double lcx,
                                            double lcy, double P, double Q, double w1, double sv, double q2,
                                            const double *F) {
  const double d1x = a1x - b1x, d1y = a1y - b1y;
  const double q1 = __builtin_fma(d1x, d1x, d1y * d1y);
  const GateEpi ea = gate3_epi(F, a1x, a1y), eb = gate3_epi(F, b1x, b1y);
  return gate3_core_fma(cfg, q1, rs1x, rs1y, rs1z, re1x, re1y, re1z, n2x, n2y, n2z, lcx, lcy, P, Q, w1, sv, q2, ea, eb);
}

// Three-way form of `line.sensitivity(view) > sens_th` without a division or a square root: with b = Minv (mid, 1) the
// reference's |dir . unit(b)| is sqrt((dir . b)^2 / |b|^2), so the cosine-domain band of sensitivity_gt becomes
// (dir . b)^2 against sens_{lo,hi}^2 |b|^2.  The two perspective divisions of the projected endpoints use a refined
// reciprocal (not IEEE: ~1e-16 relative, against a band of 1e-7 + 1e-9).  1: certainly greater, 0: certainly not,
// 2: inside the band (or not finite) -- sensitivity_gt, the reference's arithmetic, decides.  The exact form costs
// 7 IEEE divisions and a square root per view, a third of gen_finish's instructions.
static __device__ __forceinline__ int sensitivity3(const GenCfg &cfg, const Cam &c, d3 s, d3 e, d3 dir3) {
  d3 vs = mv(c.R, s), ve = mv(c.R, e);
  vs.x += c.t[0]; vs.y += c.t[1]; vs.z += c.t[2];
  ve.x += c.t[0]; ve.y += c.t[1]; ve.z += c.t[2];
  const double rs = fast_rcp(vs.z + kEps), re = fast_rcp(ve.z + kEps);
  const double mx = 0.5 * (__builtin_fma(c.fx, vs.x, c.cx * vs.z) * rs + __builtin_fma(c.fx, ve.x, c.cx * ve.z) * re);
  const double my = 0.5 * (__builtin_fma(c.fy, vs.y, c.cy * vs.z) * rs + __builtin_fma(c.fy, ve.y, c.cy * ve.z) * re);
  const d3 b = cam_backproject(c, d2{mx, my});
  const double db = dot(dir3, b);
  const double dd = db * db, bb = dot(b, b);
  if (dd > cfg.sens_hi2 * bb) return 1;
  if (dd < cfg.sens_lo2 * bb) return 0;
  return 2;
}

// Stage B: triangulation, cheirality, sensitivity gate, uncertainty, ranges (:309-333).
static __device__ __forceinline__ bool gen_finish(const GenCfg &cfg, const Cam &c1, const Cam &c2,
                                                  const Seg &s1, const Seg &s2, const double *Bv,
                                                  GenOut *out) {
  d3 r1s = mk3(s1.rs[0], s1.rs[1], s1.rs[2]), r1e = mk3(s1.re[0], s1.re[1], s1.re[2]);
  d3 ps, pe;
  double z_start, z_end, d21, d22;
  if (!cfg.use_endpoints) {
    if (!tri_line(c1, c2, s1, s2, Bv, &ps, &pe, &z_start, &z_end, &d21, &d22)) return false;
  } else {
    // triangulate_line_by_endpoints (functions.cc:172-190)
    d3 c2s = mk3(s2.rs[0], s2.rs[1], s2.rs[2]), c2e = mk3(s2.re[0], s2.re[1], s2.re[2]);
    if (!tri_point(c1, r1s, c2, c2s, &ps)) return false;
    if (!tri_point(c1, r1e, c2, c2e, &pe)) return false;
    z_start = cam_depth(c1, ps);
    z_end = cam_depth(c1, pe);
    d21 = cam_depth(c2, ps);
    d22 = cam_depth(c2, pe);
  }
  d3 dir3 = unit(sub(pe, ps));
  // sensitivity gate (:315-317): rejected only if too sensitive in BOTH views
  {
    const int g1 = sensitivity3(cfg, c1, ps, pe, dir3);
    if (g1 != 0) {
      const int g2 = sensitivity3(cfg, c2, ps, pe, dir3);
      if (g2 != 0) {
        if (g1 == 1 && g2 == 1) return false;
        // inside a band: the reference's arithmetic for whichever view could not decide
        const bool t1 = g1 == 1 || sensitivity_gt(cfg, c1, ps, pe, dir3);
        if (t1 && (g2 == 1 || sensitivity_gt(cfg, c2, ps, pe, dir3))) return false;
      }
    }
  }
  // uncertainty (:319-321; linebase.cc:109-116; camera.cc:228-242)
  double u1 = cfg.var2d * ((z_start + z_end) / 2.0) / c1.f;
  double u2 = cfg.var2d * ((d21 + d22) / 2.0) / c2.f;
  // ranges (:330-333; functions.cc:8-26)
  if (cfg.use_ranges) {
    if (ps.x < cfg.lo[0] || ps.x > cfg.hi[0]) return false;
    if (ps.y < cfg.lo[1] || ps.y > cfg.hi[1]) return false;
    if (ps.z < cfg.lo[2] || ps.z > cfg.hi[2]) return false;
    if (pe.x < cfg.lo[0] || pe.x > cfg.hi[0]) return false;
    if (pe.y < cfg.lo[1] || pe.y > cfg.hi[1]) return false;
    if (pe.z < cfg.lo[2] || pe.z > cfg.hi[2]) return false;
  }
  out->r.s[0] = ps.x; out->r.s[1] = ps.y; out->r.s[2] = ps.z;
  out->r.e[0] = pe.x; out->r.e[1] = pe.y; out->r.e[2] = pe.z;
  out->r.depth[0] = z_start; out->r.depth[1] = z_end;
  out->unc = dmin(u1, u2);
  out->r.seg[0] = s2.x1; out->r.seg[1] = s2.y1; out->r.seg[2] = s2.x2; out->r.seg[3] = s2.y2;
  out->r.dir[0] = dir3.x; out->r.dir[1] = dir3.y; out->r.dir[2] = dir3.z;
  return true;
}

// The cheap part of gen_finish -- the two ray/plane intersections, cheirality in both views, the ranges -- as a
// pre-test for the list kernels: false means gen_finish (same functions, same operations) certainly returns false;
// true decides nothing.  About 100 instructions against ~1900 for the rest (direction, two sensitivities with their
// projections and normalisations, uncertainty).  Not applicable with use_endpoints (returns true).
static __device__ __forceinline__ bool gen_pretest(const GenCfg &cfg, const Cam &c1, const Cam &c2, const Seg &s1,
                                                   const Seg &s2, const double *Bv) {
  if (cfg.use_endpoints) return true;
  d3 ps, pe;
  double z_start, z_end, d21, d22;
  if (!tri_line(c1, c2, s1, s2, Bv, &ps, &pe, &z_start, &z_end, &d21, &d22)) return false;
  if (cfg.use_ranges) {
    if (ps.x < cfg.lo[0] || ps.x > cfg.hi[0]) return false;
    if (ps.y < cfg.lo[1] || ps.y > cfg.hi[1]) return false;
    if (ps.z < cfg.lo[2] || ps.z > cfg.hi[2]) return false;
    if (pe.x < cfg.lo[0] || pe.x > cfg.hi[0]) return false;
    if (pe.y < cfg.lo[1] || pe.y > cfg.hi[1]) return false;
    if (pe.z < cfg.lo[2] || pe.z > cfg.hi[2]) return false;
  }
  return true;
}

// Step 2 of triangulateOneNode (base_line_triangulator.cc:250-281): triangulate_line_with_direction
// (functions.cc:385-442) with the direction of a vanishing point mapped into the world through VIEW 1
// (getDirectionFromVP, functions.cc:37-42 -- the reference uses view1 for both lines' VPs), then
// uncertainty and ranges like every other proposal.  No sensitivity gate on this branch.
static __device__ __forceinline__ bool dir_candidate(const GenCfg &cfg, const Cam &c1, const Cam &c2, const Seg &s1,
                                                     const Seg &s2, const double *Bv, d3 direction, GenOut *out) {
  const d3 n1 = mk3(s1.n[0], s1.n[1], s1.n[2]);
  const double nd = dot(n1, direction);
  d3 direc = sub(direction, scale(n1, nd));
  if (sqrt(dot(direc, direc)) < kEps) return false;
  direc = unit(direc);
  const d3 perp = cross(n1, direc);
  const d3 v1s = mk3(s1.rs[0], s1.rs[1], s1.rs[2]), v1e = mk3(s1.re[0], s1.re[1], s1.re[2]);
  double a1s = dot(v1s, perp);
  double a1e = dot(v1e, perp);
  if (a1s < 0) {
    a1s *= -1;
    a1e *= -1;
  }
  if (a1s < 0.001 || a1e < 0.001) return false;
  const d3 C1 = cam_center(c1);
  const d3 n2 = mk3(s2.n[0], s2.n[1], s2.n[2]);
  const double c1s = dot(n2, v1s);
  const double c1e = dot(n2, v1e);
  const double b = dot(n2, mk3(Bv[0], Bv[1], Bv[2]));
  const double cc1 = c1s;
  const double cc2 = c1e * a1s / a1e;
  const double d1s_num = (cc1 + cc2) * b;
  const double d1s_denom = (cc1 * cc1 + cc2 * cc2);
  const double d1s = d1s_num / d1s_denom;
  const double d1e = d1s * a1s / a1e;
  const d3 ps = add(scale(v1s, d1s), C1);
  const d3 pe = add(scale(v1e, d1e), C1);
  const double z_start = cam_depth(c1, ps), z_end = cam_depth(c1, pe);
  if (z_start < kEps || z_end < kEps) return false;
  const double d21 = cam_depth(c2, ps), d22 = cam_depth(c2, pe);
  if (d21 < kEps || d22 < kEps) return false;
  if (isnan(ps.x) || isnan(pe.x)) return false;
  const double u1 = cfg.var2d * ((z_start + z_end) / 2.0) / c1.f;
  const double u2 = cfg.var2d * ((d21 + d22) / 2.0) / c2.f;
  if (cfg.use_ranges) {
    if (ps.x < cfg.lo[0] || ps.x > cfg.hi[0]) return false;
    if (ps.y < cfg.lo[1] || ps.y > cfg.hi[1]) return false;
    if (ps.z < cfg.lo[2] || ps.z > cfg.hi[2]) return false;
    if (pe.x < cfg.lo[0] || pe.x > cfg.hi[0]) return false;
    if (pe.y < cfg.lo[1] || pe.y > cfg.hi[1]) return false;
    if (pe.z < cfg.lo[2] || pe.z > cfg.hi[2]) return false;
  }
  const d3 dir3 = unit(sub(pe, ps));
  out->r.s[0] = ps.x; out->r.s[1] = ps.y; out->r.s[2] = ps.z;
  out->r.e[0] = pe.x; out->r.e[1] = pe.y; out->r.e[2] = pe.z;
  out->r.depth[0] = z_start; out->r.depth[1] = z_end;
  out->unc = dmin(u1, u2);
  out->r.seg[0] = s2.x1; out->r.seg[1] = s2.y1; out->r.seg[2] = s2.x2; out->r.seg[3] = s2.y2;
  out->r.dir[0] = dir3.x; out->r.dir[1] = dir3.y; out->r.dir[2] = dir3.z;
  return true;
}

// Per-segment point record of the point-guided proposals: the neighbouring 2D points of a line
// (structures::PL_Bipartite2d) de-duplicated by point3D_id and sorted by it, as the reference's
// std::map<int, ...> keyed by point3D_id sees them (base_line_triangulator.cc:186-204).
struct SegPoint {
  int p3d_id;   // point3D_id
  int sfm;      // index into the SfM point array, -1 = id not in SetSfMPoints (only read when SfM points are set)
  double x, y;  // the 2D point
};
static_assert(sizeof(SegPoint) == 24, "SegPoint layout");

// Principal axis of a symmetric 3x3 scatter matrix (cyclic Jacobi, fully unrolled: no indexed arrays).
// Stands in for Eigen::JacobiSVD(points - center, ComputeThinV).matrixV().col(0) of the many-points line
// fit (base_line_triangulator.cc:218-226); the sign of the axis does not matter for the Pluecker
// projection that follows.
static __device__ __forceinline__ d3 scatter_axis(double a00, double a01, double a02, double a11, double a12,
                                                  double a22) {
  double v00 = 1, v01 = 0, v02 = 0, v10 = 0, v11 = 1, v12 = 0, v20 = 0, v21 = 0, v22 = 1;
  for (int sweep = 0; sweep < 32; ++sweep) {
    const double off = fabs(a01) + fabs(a02) + fabs(a12), diag = fabs(a00) + fabs(a11) + fabs(a22);
    if (off <= 1e-18 * diag || off == 0.0) break;
#define LT_JACOBI(app, aqq, apq, apr, aqr, vp0, vp1, vp2, vq0, vq1, vq2)                     \
  if (apq != 0.0) {                                                                          \
    const double th = (aqq - app) / (2.0 * apq);                                             \
    const double t = (th >= 0 ? 1.0 : -1.0) / (fabs(th) + sqrt(th * th + 1.0));              \
    const double c = 1.0 / sqrt(t * t + 1.0), sn = t * c;                                    \
    const double npp = app - t * apq, nqq = aqq + t * apq;                                   \
    const double npr = c * apr - sn * aqr, nqr = sn * apr + c * aqr;                         \
    app = npp; aqq = nqq; apq = 0.0; apr = npr; aqr = nqr;                                   \
    double u;                                                                                \
    u = c * vp0 - sn * vq0; vq0 = sn * vp0 + c * vq0; vp0 = u;                               \
    u = c * vp1 - sn * vq1; vq1 = sn * vp1 + c * vq1; vp1 = u;                               \
    u = c * vp2 - sn * vq2; vq2 = sn * vp2 + c * vq2; vp2 = u;                               \
  }
    // column p of V is (v0p, v1p, v2p); rotations in the (0,1), (0,2), (1,2) planes
    LT_JACOBI(a00, a11, a01, a02, a12, v00, v10, v20, v01, v11, v21)
    LT_JACOBI(a00, a22, a02, a01, a12, v00, v10, v20, v02, v12, v22)
    LT_JACOBI(a11, a22, a12, a01, a02, v01, v11, v21, v02, v12, v22)
#undef LT_JACOBI
  }
  d3 ax = mk3(v00, v10, v20);
  double best = a00;
  if (a11 > best) { best = a11; ax = mk3(v01, v11, v21); }
  if (a22 > best) { ax = mk3(v02, v12, v22); }
  return ax;
}

// Step 1.1 of triangulateOneNode (base_line_triangulator.cc:183-236): the 3D points shared by l1 and l2
// (SfM points, or triangulated from the two 2D observations), a total-least-squares line through them,
// and l1's endpoint rays projected onto that infinite line with Pluecker coordinates
// (triangulate_line_with_infinite_line, functions.cc:306-321; InfiniteLine3d::project_from_infinite_line,
// infinite_line.cc:151-163).  Cheirality is tested in view 1 only, like the reference.
// *missing: a shared id without an SfM point (std::map::at would throw in the reference).
static __device__ __forceinline__ bool points_candidate(const GenCfg &cfg, const Cam &c1, const Cam &c2,
                                                        const Seg &s1, const Seg &s2, const SegPoint *pa, int na,
                                                        const SegPoint *pb, int nb, const double *sfm_xyz,
                                                        GenOut *out, bool *missing) {
  d3 center = mk3(0, 0, 0);
  double a00 = 0, a01 = 0, a02 = 0, a11 = 0, a12 = 0, a22 = 0;
  int n = 0;
  for (int pass = 0; pass < 2; ++pass) {
    int i = 0, j = 0;
    while (i < na && j < nb) {
      const int ia = pa[i].p3d_id, ib = pb[j].p3d_id;
      if (ia < ib) { ++i; continue; }
      if (ib < ia) { ++j; continue; }
      d3 P;
      bool ok = true;
      if (sfm_xyz) {
        const int idx = pa[i].sfm;
        if (idx < 0) { *missing = true; ok = false; }
        else P = mk3(sfm_xyz[3 * idx], sfm_xyz[3 * idx + 1], sfm_xyz[3 * idx + 2]);
      } else {
        ok = tri_point(c1, cam_ray(c1, d2{pa[i].x, pa[i].y}), c2, cam_ray(c2, d2{pb[j].x, pb[j].y}), &P);
      }
      if (ok) {
        if (pass == 0) {
          center = add(center, P);
          ++n;
        } else {
          const d3 e = sub(P, center);
          a00 += e.x * e.x; a01 += e.x * e.y; a02 += e.x * e.z;
          a11 += e.y * e.y; a12 += e.y * e.z; a22 += e.z * e.z;
        }
      }
      ++i; ++j;
    }
    if (pass == 0) {
      if (n < 2) return false;
      const double dn = (double)n;
      center = mk3(center.x / dn, center.y / dn, center.z / dn);
    }
  }
  const d3 direc = unit(scatter_axis(a00, a01, a02, a11, a12, a22));
  const d3 m2 = cross(center, direc);
  const d3 C1 = cam_center(c1);
  d3 pend[2];
  for (int k = 0; k < 2; ++k) {
    const d3 v = k == 0 ? mk3(s1.rs[0], s1.rs[1], s1.rs[2]) : mk3(s1.re[0], s1.re[1], s1.re[2]);
    const d3 m1 = cross(C1, v);
    const d3 cr = cross(v, direc);
    const d3 t1 = cross(m1, cross(direc, cr));
    const double w = dot(m2, cr);
    d3 pt = add(scale(t1, -1.0), scale(v, w));
    const double q = dot(cr, cr);
    pend[k] = mk3(pt.x / q, pt.y / q, pt.z / q);
  }
  const d3 ps = pend[0], pe = pend[1];
  const double z_start = cam_depth(c1, ps), z_end = cam_depth(c1, pe);
  if (z_start < kEps || z_end < kEps) return false;
  const double d21 = cam_depth(c2, ps), d22 = cam_depth(c2, pe);
  const double u1 = cfg.var2d * ((z_start + z_end) / 2.0) / c1.f;
  const double u2 = cfg.var2d * ((d21 + d22) / 2.0) / c2.f;
  if (cfg.use_ranges) {
    if (ps.x < cfg.lo[0] || ps.x > cfg.hi[0]) return false;
    if (ps.y < cfg.lo[1] || ps.y > cfg.hi[1]) return false;
    if (ps.z < cfg.lo[2] || ps.z > cfg.hi[2]) return false;
    if (pe.x < cfg.lo[0] || pe.x > cfg.hi[0]) return false;
    if (pe.y < cfg.lo[1] || pe.y > cfg.hi[1]) return false;
    if (pe.z < cfg.lo[2] || pe.z > cfg.hi[2]) return false;
  }
  const d3 dir3 = unit(sub(pe, ps));
  out->r.s[0] = ps.x; out->r.s[1] = ps.y; out->r.s[2] = ps.z;
  out->r.e[0] = pe.x; out->r.e[1] = pe.y; out->r.e[2] = pe.z;
  out->r.depth[0] = z_start; out->r.depth[1] = z_end;
  out->unc = dmin(u1, u2);
  out->r.seg[0] = s2.x1; out->r.seg[1] = s2.y1; out->r.seg[2] = s2.x2; out->r.seg[3] = s2.y2;
  out->r.dir[0] = dir3.x; out->r.dir[1] = dir3.y; out->r.dir[2] = dir3.z;
  return true;
}

// ---- one-point proposal (triangulate_line_with_one_point, functions.cc:325-383) -------------------------
// The reference's solver (solvers/triangulation: generated coefficients of a quartic in a Lagrange
// multiplier + PoseLib's root finder) is not copied; the optimisation problem it solves is restated:
// in plane-1 coordinates, p1 / p2 unit directions of l1's endpoint rays, p the known point, (lx, ly, lz)
// the trace of plane 2:  minimise (l.x1)^2 + (l.x2)^2, x1 = lambda1 p1, x2 = lambda2 p2 collinear with p,
// lambda > 0.  With c = p1 x p2, a = p1 x p, b = p x p2 collinearity gives lambda2 = a u / (c u - b),
// u = lambda1, and the stationary points are the real roots of
//     alpha1 (alpha1 u + lz) w^3 - alpha2 a b (alpha2 a u + lz w) = 0,   w = c u - b.
static __device__ __forceinline__ int roots_cubic_monic(double A, double B, double C, double *out) {
  const double sh = A / 3.0;
  const double P = B - A * A / 3.0;
  const double Q = 2.0 * A * A * A / 27.0 - A * B / 3.0 + C;
  const double disc = Q * Q / 4.0 + P * P * P / 27.0;
  if (disc > 0) {
    const double sq = sqrt(disc);
    out[0] = cbrt(-Q / 2.0 + sq) + cbrt(-Q / 2.0 - sq) - sh;
    return 1;
  }
  if (P == 0.0) {
    out[0] = -sh;
    return 1;
  }
  const double m = 2.0 * sqrt(-P / 3.0);
  double arg = 3.0 * Q / (P * m);
  arg = arg < -1.0 ? -1.0 : (arg > 1.0 ? 1.0 : arg);
  const double th = acos(arg) / 3.0;
  out[0] = m * cos(th) - sh;
  out[1] = m * cos(th - 2.0 * kPi / 3.0) - sh;
  out[2] = m * cos(th - 4.0 * kPi / 3.0) - sh;
  return 3;
}
static __device__ __forceinline__ int roots_quadratic(double a, double b, double c, double *out) {
  const double d = b * b - 4.0 * a * c;
  if (d < 0) return 0;
  const double sq = sqrt(d);
  const double q = -0.5 * (b + (b >= 0 ? sq : -sq));
  out[0] = q / a;
  out[1] = q != 0.0 ? c / q : 0.0;
  return 2;
}
static __device__ int roots_poly4(const double *Ain, double *out) {
  double A[5];
  double mx = 0;
  for (int k = 0; k < 5; ++k) mx = fmax(mx, fabs(Ain[k]));
  if (!(mx > 0) || !isfinite(mx)) return 0;
  for (int k = 0; k < 5; ++k) A[k] = Ain[k] / mx;
  int deg = 4;
  while (deg > 0 && fabs(A[deg]) < 1e-13) --deg;
  int n = 0;
  if (deg == 0) return 0;
  if (deg == 1) {
    out[n++] = -A[0] / A[1];
  } else if (deg == 2) {
    n = roots_quadratic(A[2], A[1], A[0], out);
  } else if (deg == 3) {
    n = roots_cubic_monic(A[2] / A[3], A[1] / A[3], A[0] / A[3], out);
  } else {
    const double a = A[3] / A[4], b = A[2] / A[4], c = A[1] / A[4], d = A[0] / A[4];
    const double p = b - 3.0 * a * a / 8.0;
    const double q = c - a * b / 2.0 + a * a * a / 8.0;
    const double r = d - a * c / 4.0 + a * a * b / 16.0 - 3.0 * a * a * a * a / 256.0;
    double ys[4];
    int ny = 0;
    if (fabs(q) < 1e-14 * (1.0 + fabs(p) + fabs(r))) {
      double t[2];
      const int nt = roots_quadratic(1.0, p, r, t);
      for (int k = 0; k < nt; ++k)
        if (t[k] >= 0) {
          const double sq = sqrt(t[k]);
          ys[ny++] = sq;
          ys[ny++] = -sq;
        }
    } else {
      double z[3];
      const int nz = roots_cubic_monic(2.0 * p, p * p - 4.0 * r, -q * q, z);
      double z0 = z[0];
      for (int k = 1; k < nz; ++k) z0 = fmax(z0, z[k]);
      if (z0 > 0) {
        const double sgm = sqrt(z0);
        double t[2];
        int nt = roots_quadratic(1.0, sgm, (p + z0 - q / sgm) / 2.0, t);
        for (int k = 0; k < nt; ++k) ys[ny++] = t[k];
        nt = roots_quadratic(1.0, -sgm, (p + z0 + q / sgm) / 2.0, t);
        for (int k = 0; k < nt; ++k) ys[ny++] = t[k];
      }
    }
    for (int k = 0; k < ny; ++k) out[n++] = ys[k] - a / 4.0;
  }
  for (int k = 0; k < n; ++k) {
    double x = out[k];
    for (int it = 0; it < 3; ++it) {
      const double f = (((A[4] * x + A[3]) * x + A[2]) * x + A[1]) * x + A[0];
      const double df = ((4.0 * A[4] * x + 3.0 * A[3]) * x + 2.0 * A[2]) * x + A[1];
      if (df == 0.0 || !isfinite(f / df)) break;
      x = x - f / df;
    }
    out[k] = x;
  }
  return n;
}
static __device__ bool solve_one_point(double lx, double ly, double lz, d2 p, d2 p1, d2 p2, double *lam1,
                                       double *lam2) {
  const double al1 = lx * p1.x + ly * p1.y, al2 = lx * p2.x + ly * p2.y;
  const double c = p1.x * p2.y - p1.y * p2.x, a = p1.x * p.y - p1.y * p.x, b = p.x * p2.y - p.y * p2.x;
  double A[5];
  A[4] = al1 * al1 * c * c * c;
  A[3] = al1 * (lz * c * c * c - 3.0 * al1 * c * c * b);
  A[2] = al1 * (3.0 * al1 * c * b * b - 3.0 * lz * c * c * b);
  A[1] = al1 * (3.0 * lz * c * b * b - al1 * b * b * b) - al2 * a * b * (al2 * a + lz * c);
  A[0] = lz * b * b * (al2 * a - al1 * b);
  double roots[4];
  const int n = roots_poly4(A, roots);
  bool found = false;
  double best_err = 1.7976931348623157e308;
  for (int k = 0; k < n; ++k) {
    double u = roots[k];
    // Newton on the factored form: the expanded coefficients carry the cancellation of (c u - b)^3
    for (int it = 0; it < 4; ++it) {
      const double w = c * u - b;
      const double f = al1 * (al1 * u + lz) * (w * w * w) - al2 * a * b * (al2 * a * u + lz * w);
      const double df = al1 * al1 * (w * w * w) + 3.0 * al1 * (al1 * u + lz) * (w * w) * c - al2 * a * b * (al2 * a + lz * c);
      if (df == 0.0 || !isfinite(f / df)) break;
      u = u - f / df;
    }
    const double w = c * u - b;
    if (w == 0.0) continue;
    const double l2v = a * u / w;
    if (!(u > 0) || !(l2v > 0)) continue;
    const double e1 = al1 * u + lz, e2 = al2 * l2v + lz;
    const double err = e1 * e1 + e2 * e2;
    if (err < best_err) {
      best_err = err;
      *lam1 = u;
      *lam2 = l2v;
      found = true;
    }
  }
  return found;
}

static __device__ bool one_point_candidate(const GenCfg &cfg, const Cam &c1, const Cam &c2, const Seg &s1,
                                           const Seg &s2, d3 point, GenOut *out) {
  const d3 n1 = mk3(s1.n[0], s1.n[1], s1.n[2]);
  const d3 C1 = cam_center(c1), C2 = cam_center(c2);
  const d3 pp = sub(point, scale(n1, dot(n1, sub(point, C1))));
  const d3 v1s = mk3(s1.rs[0], s1.rs[1], s1.rs[2]), v1e = mk3(s1.re[0], s1.re[1], s1.re[2]);
  const d3 n2 = mk3(s2.n[0], s2.n[1], s2.n[2]);
  const double alpha = (-1) * dot(n2, C2);
  const d3 r0 = v1s;
  const d3 r1 = unit(sub(v1e, scale(v1s, dot(v1s, v1e))));
  const d3 r2 = unit(cross(r0, r1));
  const d3 v2_t = mk3(dot(r0, v1e), dot(r1, v1e), dot(r2, v1e));
  const d3 pc = sub(pp, C1);
  const d3 p_t = mk3(dot(r0, pc), dot(r1, pc), dot(r2, pc));
  const d3 n2_t = mk3(dot(r0, n2), dot(r1, n2), dot(r2, n2));
  const double alpha_t = alpha + dot(n2, C1);
  const d2 ip = d2{p_t.x, p_t.y};
  const d2 iv1 = d2{1.0, 0.0};  // (1, 0).normalized()
  double nv = sqrt(v2_t.x * v2_t.x + v2_t.y * v2_t.y);
  const d2 iv2 = nv > 0 ? d2{v2_t.x / nv, v2_t.y / nv} : d2{v2_t.x, v2_t.y};
  double lam1 = -1, lam2 = -1;
  if (!solve_one_point(n2_t.x, n2_t.y, alpha_t, ip, iv1, iv2, &lam1, &lam2)) return false;
  const d2 ls2 = d2{iv1.x * lam1, iv1.y * lam1}, le2 = d2{iv2.x * lam2, iv2.y * lam2};
  const d3 ps = add(add(scale(r0, ls2.x), scale(r1, ls2.y)), C1);
  const d3 pe = add(add(scale(r0, le2.x), scale(r1, le2.y)), C1);
  const double z_start = cam_depth(c1, ps), z_end = cam_depth(c1, pe);
  if (z_start < kEps || z_end < kEps) return false;
  const double d21 = cam_depth(c2, ps), d22 = cam_depth(c2, pe);
  if (d21 < kEps || d22 < kEps) return false;
  const double u1 = cfg.var2d * ((z_start + z_end) / 2.0) / c1.f;
  const double u2 = cfg.var2d * ((d21 + d22) / 2.0) / c2.f;
  if (cfg.use_ranges) {
    if (ps.x < cfg.lo[0] || ps.x > cfg.hi[0]) return false;
    if (ps.y < cfg.lo[1] || ps.y > cfg.hi[1]) return false;
    if (ps.z < cfg.lo[2] || ps.z > cfg.hi[2]) return false;
    if (pe.x < cfg.lo[0] || pe.x > cfg.hi[0]) return false;
    if (pe.y < cfg.lo[1] || pe.y > cfg.hi[1]) return false;
    if (pe.z < cfg.lo[2] || pe.z > cfg.hi[2]) return false;
  }
  const d3 dir3 = unit(sub(pe, ps));
  out->r.s[0] = ps.x; out->r.s[1] = ps.y; out->r.s[2] = ps.z;
  out->r.e[0] = pe.x; out->r.e[1] = pe.y; out->r.e[2] = pe.z;
  out->r.depth[0] = z_start; out->r.depth[1] = z_end;
  out->unc = dmin(u1, u2);
  out->r.seg[0] = s2.x1; out->r.seg[1] = s2.y1; out->r.seg[2] = s2.x2; out->r.seg[3] = s2.y2;
  out->r.dir[0] = dir3.x; out->r.dir[1] = dir3.y; out->r.dir[2] = dir3.z;
  return true;
}

static __device__ __forceinline__ bool vp_candidate(const GenCfg &cfg, const Cam &c1, const Cam &c2, const Seg &s1,
                                                    const Seg &s2, const double *Bv, const double *vp,
                                                    GenOut *out) {
  // getDirectionFromVP(vp, view1), functions.cc:37-42
  return dir_candidate(cfg, c1, c2, s1, s2, Bv, unit(mv(c1.Minv, mk3(vp[0], vp[1], vp[2]))), out);
}

// (Callers run gen_gates where the cheap gates left a pair undecided and then gen_finish for EVERY lane from one call
// site: `undecided ? gates-then-finish : finish` as two inlined copies made a wave with mixed lanes run gen_finish twice.)

// Dense evaluation of one (i, j) pair (global_line_triangulator.cc:97-104): LineLinker3d score in
// shared-parent mode (3D angle + one-way scale-invariant endpoint distance with l_i's depths,
// line_linker.cc:306-331 with the flags of line_linker.h:115-121), then LineLinker2d score of l_i
// projected into the view of j against the 2D segment that generated j.  The unit directions are
// the ones stored with the candidates: Line3d::direction() is a pure function of the endpoints, so
// the stored value is bit-identical to the reference's recomputation.
static __device__ __forceinline__ double pair_score_terms(const ScoreCfg &cfg, d3 si, d3 ei, d3 diri, double dep0,
                                                    double dep1, d3 sj, d3 ej, d3 dirj, const double *segj,
                                                    const Cam &camj) {
  const LinkCfg3 &c3 = cfg.l3;
  double s3 = 1.0;
  {  // score_angle
    double ang = angle_deg_from_cos(fabs(dot(diri, dirj)));
    s3 = dmin(s3, gate(expscore(ang, c3.th_angle * c3.mult), c3.score_th));
  }
  if (s3 < c3.score_th) return 0.0;
  {  // score_scaleinv
    double ds = sqrt(sqn(sub(si, sj)));
    double de = sqrt(sqn(sub(ei, ej)));
    double d = dmax(ds / (dep0 + kEps), de / (dep1 + kEps));
    s3 = dmin(s3, gate(expscore(d, c3.th_scaleinv * c3.mult), c3.score_th));
  }
  if (s3 == 0) return 0.0;
  L2 pi{cam_project(camj, si), cam_project(camj, ei)};
  L2 sg{mk2(segj[0], segj[1]), mk2(segj[2], segj[3])};
  double s2 = score2d(cfg.l2, pi, sg);
  if (s2 == 0) return 0.0;
  return dmin(s3, s2);
}


// The same function with every shared sub-expression evaluated ONCE and ONE exponential (round 6).  pair_score_terms is the
// reference's text term by term: dir(l) and len(l) of the two 2D lines are recomputed by angle_between, both overlap_oneway
// and both perp_oneway (the compiler does not merge them across the early returns: 30 IEEE divisions, 14 square roots and
// five exp per pair), although they are the same expressions on the same inputs --
//   * len(l) = sqrt(sqn(l.s - l.e)) and the norm inside dir(l) = unit(l.e - l.s) are the same double: (a - b) = -(b - a)
//     exactly, squares and their sum agree bit for bit;
//   * overlap_oneway's numerators dot(l1.s - l2.s, dir(l2)), dot(l1.e - l2.s, dir(l2)) are perp_oneway's pa, pb.
// 22 divisions, 8 square roots (+ the two inside acos), same bits.
// One exp: every term's score is exp(-(q * q) / 2) with its own q = value / sigma, gated `< score_th -> 0`; the result is 0 if
// a gate fails, else the minimum = the exponential of the LARGEST q (the exponential's argument -(q * q) / 2 is monotone in
// q, rounding included).  Whether a gate fails is decided on q against sqrt(-2 ln score_th) with a +-1e-9 band inside which the
// exponential itself is evaluated (ScoreCfg::q*_lo / q*_hi, lt_api.cpp: make_score).  NaN terms take no part in dmin(score,
// NaN) = score of the term-by-term form; here `q > qmax` and the band tests are false for them: the same.
// tests/test_gpu_guards.py::test_pair_score_fused_equals_term_by_term holds both forms to the same bits
// (LT_TEST_PAIR_SCORE_TERMS=1 runs the reference's text).
static __device__ __forceinline__ bool term_fails(double q, double q_lo, double q_hi, double score_th) {
  if (q >= q_hi) return true;
  if (q > q_lo) return exp(-(q * q) / 2.0) < score_th;  // inside the band: the reference's own comparison
  return false;
}
static __device__ __forceinline__ double pair_score_fused(const ScoreCfg &cfg, d3 si, d3 ei, d3 diri, double dep0,
                                                          double dep1, d3 sj, d3 ej, d3 dirj, const double *segj,
                                                          const Cam &camj) {
  const LinkCfg3 &c3 = cfg.l3;
  const LinkCfg2 &c2 = cfg.l2;
  double qmax = 0.0;  // exp(-0) = 1: the score every linker starts from
  {  // LineLinker3d, shared-parent mode: score_angle (line_linker.cc:185-192)
    const double ang = angle_deg_from_cos(fabs(dot(diri, dirj)));
    const double q = ang / (c3.th_angle * c3.mult);
    if (term_fails(q, cfg.q3_lo, cfg.q3_hi, c3.score_th)) return 0.0;
    qmax = q > qmax ? q : qmax;
  }
  {  // score_scaleinv (line_linker.cc:269-277, line_dists.cc:55-60)
    const double ds = sqrt(sqn(sub(si, sj)));
    const double de = sqrt(sqn(sub(ei, ej)));
    const double d = dmax(ds / (dep0 + kEps), de / (dep1 + kEps));
    const double q = d / (c3.th_scaleinv * c3.mult);
    if (term_fails(q, cfg.q3_lo, cfg.q3_hi, c3.score_th)) return 0.0;
    qmax = q > qmax ? q : qmax;
  }
  // LineLinker2d::compute_score (line_linker.cc:139-160) of l_i projected into the view of j against j's 2D segment
  const d2 ps = cam_project(camj, si), pe = cam_project(camj, ei);
  const d2 gs = mk2(segj[0], segj[1]), ge = mk2(segj[2], segj[3]);
  // dir / len of both lines, once
  const d2 v1 = sub(pe, ps), v2 = sub(ge, gs);
  const double z1 = sqn(v1), z2 = sqn(v2);
  const double n1 = sqrt(z1), n2 = sqrt(z2);  // = len(l1), len(l2)
  const d2 u1 = z1 > 0.0 ? d2{v1.x / n1, v1.y / n1} : v1;
  const d2 u2 = z2 > 0.0 ? d2{v2.x / n2, v2.y / n2} : v2;
  double ang2 = 0.0;
  const bool need_ang = c2.use_angle || (c2.use_overlap && c2.use_smartangle);
  if (need_ang) ang2 = angle_deg_from_cos(fabs(dot(u1, u2)));
  if (c2.use_angle) {
    const double q = ang2 / (c2.th_angle * c2.mult);
    if (term_fails(q, cfg.q2_lo, cfg.q2_hi, c2.score_th)) return 0.0;
    qmax = q > qmax ? q : qmax;
  }
  // offsets of either line's endpoints from the other line's start, along and across the other line
  const d2 a = sub(ps, gs), b = sub(pe, gs);    // l1 against l2
  const double pa = dot(a, u2), pb = dot(b, u2);
  const d2 a2 = sub(gs, ps), b2 = sub(ge, ps);  // l2 against l1
  const double pa2 = dot(a2, u1), pb2 = dot(b2, u1);
  if (c2.use_overlap) {  // bioverlap (line_dists.h:189-208)
    double p1 = pa / n2, p2 = pb / n2;
    if (p1 > p2) { const double t = p1; p1 = p2; p2 = t; }
    const double o1 = dmin(p2, 1.0) - dmax(p1, 0.0);
    double r1 = pa2 / n1, r2 = pb2 / n1;
    if (r1 > r2) { const double t = r1; r1 = r2; r2 = t; }
    const double o2 = dmin(r2, 1.0) - dmax(r1, 0.0);
    const double ov = dmax(o1, o2);
    if (!(ov > c2.th_overlap)) return 0.0;
    if (c2.use_angle && c2.use_smartangle) {  // line_linker.cc:49-65
      double th = c2.th_angle;
      if (ov < c2.th_smartoverlap) {
        double ratio = (c2.th_smartoverlap - ov) / (c2.th_smartoverlap - c2.th_overlap);
        ratio = dmin(ratio, 1.0);
        th = c2.th_angle - ratio * (c2.th_angle - c2.th_smartangle);
      }
      const double q = ang2 / (th * c2.mult);
      if (term_fails(q, cfg.q2_lo, cfg.q2_hi, c2.score_th)) return 0.0;
      qmax = q > qmax ? q : qmax;
    }
  }
  if (c2.use_perp) {  // perp_dist (line_dists.h:98-133): max of the four endpoint distances
    double m = sqrt(dmax(sqn(a) - pa * pa, 0.0));
    const double de = sqrt(dmax(sqn(b) - pb * pb, 0.0));
    const double dc = sqrt(dmax(sqn(a2) - pa2 * pa2, 0.0));
    const double dd = sqrt(dmax(sqn(b2) - pb2 * pb2, 0.0));
    if (m < de) m = de;
    if (m < dc) m = dc;
    if (m < dd) m = dd;
    const double q = m / (c2.th_perp * c2.mult);
    if (term_fails(q, cfg.q2_lo, cfg.q2_hi, c2.score_th)) return 0.0;
    qmax = q > qmax ? q : qmax;
  }
  return exp(-(qmax * qmax) / 2.0);
}
// run-time choice (the fused k_score3; k_dense8 is compiled once per form: both bodies in one kernel cost it 34 registers
// and the fourth wave per SIMD)
static __device__ __forceinline__ double pair_score(const ScoreCfg &cfg, d3 si, d3 ei, d3 diri, double dep0,
                                                    double dep1, d3 sj, d3 ej, d3 dirj, const double *segj,
                                                    const Cam &camj) {
  if (!cfg.fast) return pair_score_terms(cfg, si, ei, diri, dep0, dep1, sj, ej, dirj, segj, camj);
  return pair_score_fused(cfg, si, ei, diri, dep0, dep1, sj, ej, dirj, segj, camj);
}

}  // namespace lt
