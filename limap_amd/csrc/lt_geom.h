// lt_geom.h -- FP64 geometry shared by the HIP kernels and the host tail of the MI355X backend.
//
// Everything here is plain IEEE double arithmetic in a FIXED evaluation order (the translation
// unit is compiled with -ffp-contract=off): sums of three products are (a0*b0 + a1*b1) + a2*b2,
// divisions are true divisions, sqrt is correctly rounded.  That makes every quantity that does
// not pass through acos/exp a pure function of its inputs, so per-camera / per-segment /
// per-image-pair invariants can be hoisted out of the hot loops (the reference recomputes
// R(), K_inv(), the fundamental matrix ... per connection) without changing a single bit of
// the results.  Reference semantics followed (paths relative to /root/reference/src/limap):
//   util/types.h:35-45, base/pose.cc:12-28, base/camera.h:72-110, base/camera.cc:228-279,
//   base/camera_view.cc:61-69, base/linebase.{h,cc}, base/line_dists.{h,cc},
//   base/line_linker.{h,cc}, triangulation/functions.cc.
#pragma once

#include <hip/hip_runtime.h>
#include <math.h>

#define LT_HD __host__ __device__ __forceinline__

namespace lt {

constexpr double kEps = 1e-12;           // util/types.h:35
constexpr double kPi = 3.14159265358979323846;
constexpr double kMaxDist = 1.7976931348623157e308;  // numeric_limits<double>::max()

struct d2 {
  double x, y;
};
struct d3 {
  double x, y, z;
};

LT_HD d3 mk3(double x, double y, double z) { return d3{x, y, z}; }
LT_HD d2 mk2(double x, double y) { return d2{x, y}; }
LT_HD d3 sub(d3 a, d3 b) { return d3{a.x - b.x, a.y - b.y, a.z - b.z}; }
LT_HD d3 add(d3 a, d3 b) { return d3{a.x + b.x, a.y + b.y, a.z + b.z}; }
LT_HD d3 scale(d3 a, double s) { return d3{a.x * s, a.y * s, a.z * s}; }
LT_HD d2 sub(d2 a, d2 b) { return d2{a.x - b.x, a.y - b.y}; }
LT_HD d2 add(d2 a, d2 b) { return d2{a.x + b.x, a.y + b.y}; }
LT_HD d2 scale(d2 a, double s) { return d2{a.x * s, a.y * s}; }
LT_HD double dot(d3 a, d3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
LT_HD double dot(d2 a, d2 b) { return a.x * b.x + a.y * b.y; }
LT_HD double sqn(d3 a) { return (a.x * a.x + a.y * a.y) + a.z * a.z; }
LT_HD double sqn(d2 a) { return a.x * a.x + a.y * a.y; }
LT_HD d3 cross(d3 a, d3 b) {
  return d3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
LT_HD d3 unit(d3 a) {  // v.normalized(): unchanged if the squared norm is not > 0
  double z = sqn(a);
  if (z > 0.0) {
    double n = sqrt(z);
    return d3{a.x / n, a.y / n, a.z / n};
  }
  return a;
}
LT_HD d2 unit(d2 a) {
  double z = sqn(a);
  if (z > 0.0) {
    double n = sqrt(z);
    return d2{a.x / n, a.y / n};
  }
  return a;
}
LT_HD double dmin(double a, double b) { return b < a ? b : a; }  // std::min(a,b)
LT_HD double dmax(double a, double b) { return a < b ? b : a; }  // std::max(a,b)
// row-major 3x3 * vector
LT_HD d3 mv(const double *m, d3 v) {
  return d3{(m[0] * v.x + m[1] * v.y) + m[2] * v.z, (m[3] * v.x + m[4] * v.y) + m[5] * v.z,
            (m[6] * v.x + m[7] * v.y) + m[8] * v.z};
}
LT_HD void mm(const double *a, const double *b, double *r) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j)
      r[3 * i + j] = (a[3 * i] * b[j] + a[3 * i + 1] * b[3 + j]) + a[3 * i + 2] * b[6 + j];
}
LT_HD void tr3(const double *a, double *r) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) r[3 * i + j] = a[3 * j + i];
}
LT_HD double cof(const double *a, int i, int j) {
  int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
  return a[3 * i1 + j1] * a[3 * i2 + j2] - a[3 * i1 + j2] * a[3 * i2 + j1];
}
// 3x3 inverse: cofactors times 1/det, det expanded along column 0
LT_HD void inv3(const double *a, double *r) {
  double c0 = cof(a, 0, 0), c1 = cof(a, 1, 0), c2 = cof(a, 2, 0);
  double det = (c0 * a[0] + c1 * a[3]) + c2 * a[6];
  double id = 1.0 / det;
  r[0] = c0 * id; r[1] = c1 * id; r[2] = c2 * id;
  r[3] = cof(a, 0, 1) * id; r[4] = cof(a, 1, 1) * id; r[5] = cof(a, 2, 1) * id;
  r[6] = cof(a, 0, 2) * id; r[7] = cof(a, 1, 2) * id; r[8] = cof(a, 2, 2) * id;
}

// ---------------------------------------------------------------------------------------------
// Per-image camera record, built once at Init (256 B, one per image, resident in HBM/L2).
// ---------------------------------------------------------------------------------------------
struct Cam {
  double fx, fy, cx, cy;  // K
  double R[9];            // world -> cam, from the re-normalised quaternion
  double t[3];
  double C[3];            // centre = (-R^T) t
  double Minv[9];         // R^T K^-1  (back-projection: ray = unit(Minv x~))
  double f;               // (fx + fy) / 2
  double Kinv[9];         // kept for the fundamental matrix
  double pad_[10];         // -> 48 doubles = 384 B
};
static_assert(sizeof(Cam) == 48 * 8, "Cam layout");

LT_HD void cam_build(const double *k4, const double *q4, const double *t3, Cam *c) {
  c->fx = k4[0]; c->fy = k4[1]; c->cx = k4[2]; c->cy = k4[3];
  // CameraPose ctor normalises once (camera.h:95), R() re-normalises (pose.cc:19-28)
  double q[4];
  {
    double n0 = sqrt((q4[0] * q4[0] + q4[2] * q4[2]) + (q4[1] * q4[1] + q4[3] * q4[3]));
    for (int i = 0; i < 4; ++i) q[i] = n0 > 0.0 ? q4[i] / n0 : q4[i];
  }
  double n = sqrt((q[0] * q[0] + q[2] * q[2]) + (q[1] * q[1] + q[3] * q[3]));
  double w, x, y, z;
  if (n == 0.0) {
    w = 1.0; x = q[1]; y = q[2]; z = q[3];
  } else {
    w = q[0] / n; x = q[1] / n; y = q[2] / n; z = q[3] / n;
  }
  double tx = 2.0 * x, ty = 2.0 * y, tz = 2.0 * z;
  double twx = tx * w, twy = ty * w, twz = tz * w;
  double txx = tx * x, txy = ty * x, txz = tz * x;
  double tyy = ty * y, tyz = tz * y, tzz = tz * z;
  double *R = c->R;
  R[0] = 1.0 - (tyy + tzz); R[1] = txy - twz;         R[2] = txz + twy;
  R[3] = txy + twz;         R[4] = 1.0 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy;         R[7] = tyz + twx;         R[8] = 1.0 - (txx + tyy);
  c->t[0] = t3[0]; c->t[1] = t3[1]; c->t[2] = t3[2];
  double Rt[9];
  tr3(R, Rt);
  d3 ctr = mv(Rt, mk3(-t3[0], -t3[1], -t3[2]));
  c->C[0] = ctr.x; c->C[1] = ctr.y; c->C[2] = ctr.z;
  double K[9] = {k4[0], 0.0, k4[2], 0.0, k4[1], k4[3], 0.0, 0.0, 1.0};
  inv3(K, c->Kinv);
  mm(Rt, c->Kinv, c->Minv);
  c->f = (k4[0] + k4[1]) / 2.0;
  for (int i = 0; i < 10; ++i) c->pad_[i] = 0.0;
}

LT_HD d3 cam_center(const Cam &c) { return mk3(c.C[0], c.C[1], c.C[2]); }
LT_HD double cam_depth(const Cam &c, d3 p) {  // CameraPose::projdepth
  return ((c.R[6] * p.x + c.R[7] * p.y) + c.R[8] * p.z) + c.t[2];
}
LT_HD d2 cam_project(const Cam &c, d3 p) {  // CameraView::projection: dehom(K (R p + t))
  d3 v = mv(c.R, p);
  v.x = v.x + c.t[0]; v.y = v.y + c.t[1]; v.z = v.z + c.t[2];
  // K rows are (fx,0,cx),(0,fy,cy),(0,0,1); adding the exact-zero products changes nothing
  double hx = c.fx * v.x + c.cx * v.z;
  double hy = c.fy * v.y + c.cy * v.z;
  double hz = v.z + kEps;
  return d2{hx / hz, hy / hz};
}
LT_HD d3 cam_backproject(const Cam &c, d2 p) {  // (R^T K^-1) x~, not normalised
  return mv(c.Minv, mk3(p.x, p.y, 1.0));
}
LT_HD d3 cam_ray(const Cam &c, d2 p) { return unit(cam_backproject(c, p)); }

// Where the staged form of a block of match rows lies and where its rows go (host: lt_upload, device: k_expand_rows)
struct RowDesc {
  long long coff;     // first word of the compressed block in the stream
  long long ooff;     // first word of the plain block in the overflow array, -1: compressed
  long long row_off;  // first row of the block in the device's row array
  int n, line0;       // rows, line id of the first row
};
static_assert(sizeof(RowDesc) == 32, "RowDesc layout");

// ---------------------------------------------------------------------------------------------
// Per-segment record (128 B): endpoints + view-only invariants of the generation stage.
// ---------------------------------------------------------------------------------------------
struct Seg {
  double x1, y1, x2, y2;
  double rs[3];  // ray_direction(start)
  double re[3];  // ray_direction(end)
  double n[3];   // getNormalDirection: unit((Minv s~) x (Minv e~))   functions.cc:28-35
  double lc[3];  // Line2d::coords(): unit(s~ x e~)                    linebase.cc:35-39
};
static_assert(sizeof(Seg) == 128, "Seg layout");

LT_HD void seg_build(const Cam &c, double x1, double y1, double x2, double y2, Seg *s) {
  s->x1 = x1; s->y1 = y1; s->x2 = x2; s->y2 = y2;
  d3 bs = cam_backproject(c, mk2(x1, y1));
  d3 be = cam_backproject(c, mk2(x2, y2));
  d3 rs = unit(bs), re = unit(be);
  d3 n = unit(cross(bs, be));
  d3 lc = unit(cross(mk3(x1, y1, 1.0), mk3(x2, y2, 1.0)));
  s->rs[0] = rs.x; s->rs[1] = rs.y; s->rs[2] = rs.z;
  s->re[0] = re.x; s->re[1] = re.y; s->re[2] = re.z;
  s->n[0] = n.x; s->n[1] = n.y; s->n[2] = n.z;
  s->lc[0] = lc.x; s->lc[1] = lc.y; s->lc[2] = lc.z;
}

// ---------------------------------------------------------------------------------------------
// Per-(image, neighbour) record (96 B): fundamental matrix + baseline.  functions.cc:44-74
// ---------------------------------------------------------------------------------------------
struct PairRec {
  double F[9];
  double B[3];  // C2 - C1
};

LT_HD void pair_build(const Cam &c1, const Cam &c2, PairRec *p) {
  double R1t[9], relR[9];
  tr3(c1.R, R1t);
  mm(c2.R, R1t, relR);
  d3 rt = mv(relR, mk3(c1.t[0], c1.t[1], c1.t[2]));
  d3 relT = mk3(c2.t[0] - rt.x, c2.t[1] - rt.y, c2.t[2] - rt.z);
  double sk[9] = {0.0, -relT.z, relT.y, relT.z, 0.0, -relT.x, -relT.y, relT.x, 0.0};
  double E[9], K2it[9], tmp[9];
  mm(sk, relR, E);
  tr3(c2.Kinv, K2it);
  mm(K2it, E, tmp);
  mm(tmp, c1.Kinv, p->F);
  p->B[0] = c2.C[0] - c1.C[0];
  p->B[1] = c2.C[1] - c1.C[1];
  p->B[2] = c2.C[2] - c1.C[2];
}

// ---------------------------------------------------------------------------------------------
// 3D candidate ("TriTuple", base_line_triangulator.h:17-18).
// CRec (128 B = one cache line, 128-byte aligned arrays) is THE record of the device pipeline: everything the
// scoring, selection and edge kernels read of a candidate comes with one line.  The uncertainty -- needed only
// for the per-node best candidate and the debug read-outs -- lives in a side array (8 B per record).
// Cand / CandLite are the split host-side view (per-node best record, getters, import / export): Line3d::score
// is 1.0 for every generated proposal, so it is not stored per candidate on the device.
// ---------------------------------------------------------------------------------------------
struct CRec {  // 128 B
  double s[3], e[3];
  double depth[2];  // depths in the source view (view1)
  double seg[4];    // the neighbour's 2D segment that generated the candidate (scoring compares
                    // reprojections against exactly this segment, global_line_triangulator.cc:100-101)
  double dir[3];    // Line3d::direction()
  int nb_slot;      // (neighbour image index << 8) | index of that image in neighbors_[img]
  int ng_line;      // line id in that neighbour image
};
static_assert(sizeof(CRec) == 128, "candidate record = one cache line");
struct Cand {  // 112 B
  double s[3], e[3];
  double depth[2];  // depths in the source view (view1)
  double unc;
  double score3;    // Line3d::score (1.0 for a valid proposal)
  double seg[4];
};
struct CandLite {  // 32 B
  double dir[3];   // Line3d::direction()
  int nb_slot;     // (neighbour image index << 8) | index of that image in neighbors_[img]
  int ng_line;     // line id in that neighbour image
};
static_assert(sizeof(Cand) == 112 && sizeof(CandLite) == 32, "candidate layout");
LT_HD void crec_split(const CRec &r, double unc, Cand *c, CandLite *l) {
  for (int k = 0; k < 3; ++k) { c->s[k] = r.s[k]; c->e[k] = r.e[k]; l->dir[k] = r.dir[k]; }
  c->depth[0] = r.depth[0]; c->depth[1] = r.depth[1];
  c->unc = unc; c->score3 = 1.0;
  for (int k = 0; k < 4; ++k) c->seg[k] = r.seg[k];
  l->nb_slot = r.nb_slot; l->ng_line = r.ng_line;
}
LT_HD int crec_slot(const CRec &r) { return r.nb_slot & 0xFF; }
LT_HD int crec_img(const CRec &r) { return (int)((unsigned)r.nb_slot >> 8); }
LT_HD int lite_pack(int slot, int img) { return (img << 8) | (slot & 0xFF); }
LT_HD int lite_slot(const CandLite &l) { return l.nb_slot & 0xFF; }
LT_HD int lite_img(const CandLite &l) { return (int)((unsigned)l.nb_slot >> 8); }

// ---------------------------------------------------------------------------------------------
// Configuration in device-friendly form (constants folded on the host with glibc, like the
// reference folds them with its libm: multiplier() = 1/sqrt(-2 log score_th)).
// ---------------------------------------------------------------------------------------------
struct LinkCfg2 {
  double score_th, th_angle, th_overlap, th_smartoverlap, th_smartangle, th_perp, th_innerseg;
  double mult;  // multiplier()
  int use_angle, use_overlap, use_smartangle, use_perp, use_innerseg, pad_;
};
struct LinkCfg3 {
  double score_th, th_angle, th_overlap, th_smartoverlap, th_smartangle, th_perp, th_innerseg,
      th_scaleinv;
  double mult;
  int use_angle, use_overlap, use_smartangle, use_perp, use_innerseg, use_scaleinv;
};
struct GenCfg {
  double min_length_2d, angle_th, iou_th, sens_th, var2d;
  double lo[3], hi[3];
  int use_ranges, use_endpoints, disable_algebraic;
  int force_undecided;  // test switch: stage A never decides (every row goes through the exact gates)
  // conservative cosine-domain guards for the 1-degree ray/plane gate (see gen kernel)
  double sin_lo, sin_hi;
  // squared-length guards of the min_length_2d test: q <= len_lo2 certainly fails `sqrt(q) > min_length`,
  // q > len_hi2 certainly passes it, in between the exact expression decides
  double len_lo2, len_hi2;
  // cosine-domain guards of the sensitivity gate: sensitivity > sens_th  <=>  |dir . ray| > sin(sens_th) up to
  // libm rounding; outside [sens_lo, sens_hi] the cosine decides, inside the exact expression does
  double sens_lo, sens_hi;
  // the same band in the squared-cosine domain with a further 1e-9 on either side, for the division-free form
  // sensitivity3 (lt_devfn.h): (dir . b)^2 against these x |b|^2
  double sens_lo2, sens_hi2;
};
struct ScoreCfg {
  LinkCfg2 l2;
  LinkCfg3 l3;       // already switched to shared-parent scoring
  double cos_guard;  // |cos| below this can never pass the 3D angle gate
  double fullscore_th;
  // pair_score's single-exp form (lt_devfn.h): a term exp(-q^2 / 2) passes its gate `>= score_th` iff q <= sqrt(-2 ln score_th)
  // up to rounding -- below q*_lo it certainly passes, from q*_hi on it certainly fails, in between the exponential itself
  // decides (q3: 3D linker's score_th, q2: 2D linker's)
  double q3_lo, q3_hi, q2_lo, q2_hi;
  int max_valid_conns;
  int fast;  // 1: pair_score_fused may be used (thresholds in the range the bands are proven for, no 2D inner-segment term)
};

LT_HD double expscore(double val, double sigma) {  // line_linker.cc:15-17 (pow(x,2) == x*x)
  double q = val / sigma;
  return exp(-(q * q) / 2.0);
}
LT_HD double angle_deg_from_cos(double c) { return acos(c) * 180.0 / kPi; }

// ----- 2D segment helpers (Line2d) -----
struct L2 {
  d2 s, e;
};
LT_HD double len(const L2 &l) { return sqrt(sqn(sub(l.s, l.e))); }
LT_HD d2 dir(const L2 &l) { return unit(sub(l.e, l.s)); }
struct L3 {
  d3 s, e;
};
LT_HD double len(const L3 &l) { return sqrt(sqn(sub(l.s, l.e))); }
LT_HD d3 dir(const L3 &l) { return unit(sub(l.e, l.s)); }

template <class L>
LT_HD double overlap_oneway(const L &l1, const L &l2) {  // line_dists.h:189-200
  double ln = len(l2);
  auto v = dir(l2);
  double p1 = dot(sub(l1.s, l2.s), v) / ln;
  double p2 = dot(sub(l1.e, l2.s), v) / ln;
  if (p1 > p2) {
    double t = p1; p1 = p2; p2 = t;
  }
  return dmin(p2, 1.0) - dmax(p1, 0.0);
}
template <class L>
LT_HD double bioverlap(const L &l1, const L &l2) {  // line_dists.h:202-208
  double v1 = overlap_oneway(l1, l2);
  double v2 = overlap_oneway(l2, l1);
  return dmax(v1, v2);
}
template <class L>
LT_HD void perp_oneway(const L &l1, const L &l2, double *ds, double *de) {  // line_dists.h:98-111
  auto v2 = dir(l2);
  auto a = sub(l1.s, l2.s);
  double pa = dot(a, v2);
  *ds = sqrt(dmax(sqn(a) - pa * pa, 0.0));
  auto b = sub(l1.e, l2.s);
  double pb = dot(b, v2);
  *de = sqrt(dmax(sqn(b) - pb * pb, 0.0));
}
template <class L>
LT_HD double perp_dist(const L &l1, const L &l2) {  // line_dists.h:122-133 (max of the four)
  double a, b, c, d;
  perp_oneway(l1, l2, &a, &b);
  perp_oneway(l2, l1, &c, &d);
  double m = a;
  if (m < b) m = b;
  if (m < c) m = c;
  if (m < d) m = d;
  return m;
}
template <class L>
LT_HD bool innerseg(const L &l1, const L &l2, L *out) {  // line_dists.h:159-176
  auto d1 = dir(l1);
  auto l2v = sub(l2.e, l2.s);
  double denom = dot(l2v, d1);
  double t1 = dot(sub(l1.s, l2.s), d1) / (denom + kEps);
  double t2 = dot(sub(l1.e, l2.s), d1) / (denom + kEps);
  if (t1 > t2) {
    double t = t1; t1 = t2; t2 = t;
  }
  if (t1 >= 1.0 || t2 <= 0.0) return false;
  out->s = add(l2.s, scale(l2v, dmax(t1, 0.0)));
  out->e = add(l2.s, scale(l2v, dmin(t2, 1.0)));
  return true;
}
template <class L>
LT_HD double innerseg_dist(const L &l1, const L &l2) {  // line_dists.h:178-187
  L a, b;
  if (!innerseg(l2, l1, &a)) return kMaxDist;
  if (!innerseg(l1, l2, &b)) return kMaxDist;
  return perp_dist(a, b);
}
template <class L>
LT_HD double angle_between(const L &l1, const L &l2) {  // line_dists.h:52-66
  double c = fabs(dot(dir(l1), dir(l2)));
  return angle_deg_from_cos(c);
}
LT_HD double gate(double s, double th) { return s < th ? 0.0 : s; }

// LineLinker2d::compute_score, line_linker.cc:139-160
LT_HD double score2d(const LinkCfg2 &c, const L2 &l1, const L2 &l2) {
  double score = 1.0;
  double sig = c.th_angle * c.mult;
  double ang = 0.0;
  bool have_ang = false;
  if (c.use_angle) {
    ang = angle_between(l1, l2);
    have_ang = true;
    score = dmin(score, gate(expscore(ang, sig), c.score_th));
  }
  if (score < c.score_th) return score;
  double ov = 0.0;
  bool have_ov = false;
  if (c.use_overlap) {
    ov = bioverlap(l1, l2);
    have_ov = true;
    score = dmin(score, ov > c.th_overlap ? 1.0 : 0.0);
  }
  if (score < c.score_th) return score;
  if (c.use_angle && c.use_overlap && c.use_smartangle) {  // line_linker.cc:49-65
    if (!have_ang) ang = angle_between(l1, l2);
    if (!have_ov) ov = bioverlap(l1, l2);
    double th = c.th_angle;
    if (ov < c.th_smartoverlap) {
      double ratio = (c.th_smartoverlap - ov) / (c.th_smartoverlap - c.th_overlap);
      ratio = dmin(ratio, 1.0);
      th = c.th_angle - ratio * (c.th_angle - c.th_smartangle);
    }
    score = dmin(score, gate(expscore(ang, th * c.mult), c.score_th));
  }
  if (score < c.score_th) return score;
  if (c.use_perp)
    score = dmin(score, gate(expscore(perp_dist(l1, l2), c.th_perp * c.mult), c.score_th));
  if (score < c.score_th) return score;
  if (c.use_innerseg)
    score = dmin(score, gate(expscore(innerseg_dist(l1, l2), c.th_innerseg * c.mult), c.score_th));
  return score;
}

// LineLinker3d::compute_score, line_linker.cc:306-331 (general form; unc = line uncertainties,
// dep1 = depths of l1 for the one-way scale-invariant endpoint distance, line_dists.cc:55-60)
LT_HD double score3d(const LinkCfg3 &c, const L3 &l1, const L3 &l2, double unc1, double unc2,
                     const double *dep1) {
  double score = 1.0;
  double ang = 0.0;
  bool have_ang = false;
  if (c.use_angle) {
    ang = angle_between(l1, l2);
    have_ang = true;
    score = dmin(score, gate(expscore(ang, c.th_angle * c.mult), c.score_th));
  }
  if (score < c.score_th) return score;
  double ov = 0.0;
  bool have_ov = false;
  if (c.use_overlap) {
    ov = bioverlap(l1, l2);
    have_ov = true;
    score = dmin(score, ov > c.th_overlap ? 1.0 : 0.0);
  }
  if (score < c.score_th) return score;
  if (c.use_angle && c.use_overlap && c.use_smartangle) {  // line_linker.cc:194-210
    if (!have_ang) ang = angle_between(l1, l2);
    if (!have_ov) ov = bioverlap(l1, l2);
    double th = c.th_angle;
    if (ov < c.th_smartoverlap) {
      double ratio = (c.th_smartoverlap - ov) / (c.th_smartoverlap - c.th_overlap);
      ratio = dmin(ratio, 1.0);
      th = c.th_angle - ratio * (c.th_angle - c.th_smartangle);
    }
    score = dmin(score, gate(expscore(ang, th * c.mult), c.score_th));
  }
  if (score < c.score_th) return score;
  if (c.use_perp) {  // line_linker.cc:237-246
    double u = dmin(unc1, unc2);
    score = dmin(score, gate(expscore(perp_dist(l1, l2), c.th_perp * u * c.mult), c.score_th));
  }
  if (score < c.score_th) return score;
  if (c.use_innerseg) {  // line_linker.cc:253-262
    double u = dmin(unc1, unc2);
    score = dmin(score, gate(expscore(innerseg_dist(l1, l2), c.th_innerseg * u * c.mult), c.score_th));
  }
  if (score < c.score_th) return score;
  if (c.use_scaleinv) {  // line_linker.cc:269-277
    double ds = sqrt(sqn(sub(l1.s, l2.s)));
    double de = sqrt(sqn(sub(l1.e, l2.e)));
    double d = dmax(ds / (dep1[0] + kEps), de / (dep1[1] + kEps));
    score = dmin(score, gate(expscore(d, c.th_scaleinv * c.mult), c.score_th));
  }
  return score;
}

// LineLinker3d::check_connection, line_linker.cc:285-304 (boolean form used by RemergeLineTracks,
// merging/merging.cc:523-556): angle <= th_angle directly, the other gates through their scores.
LT_HD bool check3d(const LinkCfg3 &c, const L3 &l1, const L3 &l2, double unc1, double unc2,
                   const double *dep1) {
  double ang = 0.0;
  bool have_ang = false;
  if (c.use_angle) {  // :212-216
    ang = angle_between(l1, l2);
    have_ang = true;
    if (!(ang <= c.th_angle)) return false;
  }
  double ov = 0.0;
  bool have_ov = false;
  if (c.use_overlap) {  // :232-235
    ov = bioverlap(l1, l2);
    have_ov = true;
    if (!(ov > c.th_overlap)) return false;
  }
  if (c.use_angle && c.use_overlap && c.use_smartangle) {  // :218-221
    if (!have_ang) ang = angle_between(l1, l2);
    if (!have_ov) ov = bioverlap(l1, l2);
    double th = c.th_angle;
    if (ov < c.th_smartoverlap) {
      double ratio = (c.th_smartoverlap - ov) / (c.th_smartoverlap - c.th_overlap);
      ratio = dmin(ratio, 1.0);
      th = c.th_angle - ratio * (c.th_angle - c.th_smartangle);
    }
    if (!(gate(expscore(ang, th * c.mult), c.score_th) >= c.score_th)) return false;
  }
  if (c.use_perp) {  // :248-251
    double u = dmin(unc1, unc2);
    if (!(gate(expscore(perp_dist(l1, l2), c.th_perp * u * c.mult), c.score_th) >= c.score_th)) return false;
  }
  if (c.use_innerseg) {  // :264-267
    double u = dmin(unc1, unc2);
    if (!(gate(expscore(innerseg_dist(l1, l2), c.th_innerseg * u * c.mult), c.score_th) >= c.score_th)) return false;
  }
  if (c.use_scaleinv) {  // :279-282
    double ds = sqrt(sqn(sub(l1.s, l2.s)));
    double de = sqrt(sqn(sub(l1.e, l2.e)));
    double d = dmax(ds / (dep1[0] + kEps), de / (dep1[1] + kEps));
    if (!(gate(expscore(d, c.th_scaleinv * c.mult), c.score_th) >= c.score_th)) return false;
  }
  return true;
}

}  // namespace lt
