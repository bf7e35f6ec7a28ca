// lt_oracle.cpp -- CPU ORACLE (test infrastructure, see lt_oracle.h).
//
// A literal FP64 restatement of the reference hot path, one function per reference function,
// same operation order, same container iteration order (std::map = ascending key), OpenMP at
// the reference's two loop sites.  Every function cites the reference file:line it follows
// (paths relative to /root/reference/src/limap).  No Eigen is available here, so the Eigen 3.4
// semantics the reference relies on are spelled out by hand; they are recalled from the 3.4
// sources and are ASSUMPTIONS at the ulp level:
//   * 3-vector sums/dots reduce as (a0 + a1) + a2 (SSE2 packet of the first two, then the tail);
//   * Vector4d squaredNorm reduces as (q0^2 + q2^2) + (q1^2 + q3^2) (two SSE2 packets added);
//   * v.normalized() = v / sqrt(v.squaredNorm()) (true division) and returns v if the norm is 0;
//   * Matrix3d::inverse() = cofactor matrix times 1/det, det expanded along column 0;
//   * A*B*v evaluates (A*B) into a temporary first;  products are plain mul/add, no FMA
//     (the reference builds for baseline x86-64, no -march flag).
// PINNED (round 2) against the reference's own sources: oracle/_ref compiles the unmodified hot-path files of
// /root/reference/src/limap (oracle/Makefile `ref`) against stand-in Eigen / COLMAP / PoseLib headers
// (oracle/ref_shim/), and tests/test_oracle_vs_ref.py holds this file to it bit for bit -- candidate lists and
// order, geometry, scores, arg-max, ordered valid edges, graph sizes, tracks, the post-triangulation chain, free
// functions, on the golden fixtures and on randomised scenes / configurations in both matching modes.  What stays
// an assumption is ONLY the list above (Eigen's internal evaluation order, shared by the stand-in headers), the
// SVD's sign convention, and PoseLib's quartic root finder (one-point proposal: compared with a tolerance).
//
// Build: see oracle/Makefile (g++ -O2 -fopenmp -ffp-contract=off).

#include "eigen_svd_ref.h"
#include "lt_oracle.h"

#include <omp.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <limits>
#include <map>
#include <queue>
#include <set>
#include <stdexcept>
#include <string>
#include <tuple>
#include <vector>

namespace ora {

// ---------------------------------------------------------------------------------------------
// util/types.h
// ---------------------------------------------------------------------------------------------
static const double EPS = 1e-12;  // util/types.h:35

struct V2 {
  double x = 0, y = 0;
};
struct V3 {
  double x = 0, y = 0, z = 0;
};
struct M3 {
  double m[3][3];
};

static inline V2 operator+(V2 a, V2 b) { return {a.x + b.x, a.y + b.y}; }
static inline V2 operator-(V2 a, V2 b) { return {a.x - b.x, a.y - b.y}; }
static inline V2 operator*(V2 a, double s) { return {a.x * s, a.y * s}; }
static inline V2 operator*(double s, V2 a) { return {s * a.x, s * a.y}; }
static inline V2 operator/(V2 a, double s) { return {a.x / s, a.y / s}; }
static inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
static inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
static inline V3 operator-(V3 a) { return {-a.x, -a.y, -a.z}; }
static inline V3 operator*(V3 a, double s) { return {a.x * s, a.y * s, a.z * s}; }
static inline V3 operator*(double s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
static inline V3 operator/(V3 a, double s) { return {a.x / s, a.y / s, a.z / s}; }

static inline double dot(V2 a, V2 b) { return a.x * b.x + a.y * b.y; }
static inline double dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
static inline double sqnorm(V2 a) { return a.x * a.x + a.y * a.y; }
static inline double sqnorm(V3 a) { return (a.x * a.x + a.y * a.y) + a.z * a.z; }
static inline double norm(V2 a) { return std::sqrt(sqnorm(a)); }
static inline double norm(V3 a) { return std::sqrt(sqnorm(a)); }
static inline V3 cross(V3 a, V3 b) {
  return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
template <class V>
static inline V normalized(V a) {  // Eigen MatrixBase::normalized()
  double z = sqnorm(a);
  if (z > 0.0) return a / std::sqrt(z);
  return a;
}
static inline M3 transpose(const M3 &a) {
  M3 r;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) r.m[i][j] = a.m[j][i];
  return r;
}
static inline M3 matmul(const M3 &a, const M3 &b) {
  M3 r;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j)
      r.m[i][j] = (a.m[i][0] * b.m[0][j] + a.m[i][1] * b.m[1][j]) + a.m[i][2] * b.m[2][j];
  return r;
}
static inline V3 matvec(const M3 &a, V3 v) {
  return {(a.m[0][0] * v.x + a.m[0][1] * v.y) + a.m[0][2] * v.z,
          (a.m[1][0] * v.x + a.m[1][1] * v.y) + a.m[1][2] * v.z,
          (a.m[2][0] * v.x + a.m[2][1] * v.y) + a.m[2][2] * v.z};
}
static inline double cofactor(const M3 &a, int i, int j) {  // Eigen cofactor_3x3<i,j>
  int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
  return a.m[i1][j1] * a.m[i2][j2] - a.m[i1][j2] * a.m[i2][j1];
}
static inline M3 inverse(const M3 &a) {  // Eigen compute_inverse<Matrix3d>
  double c0 = cofactor(a, 0, 0), c1 = cofactor(a, 1, 0), c2 = cofactor(a, 2, 0);
  double det = (c0 * a.m[0][0] + c1 * a.m[1][0]) + c2 * a.m[2][0];
  double invdet = 1.0 / det;
  M3 r;
  r.m[0][0] = c0 * invdet;
  r.m[0][1] = c1 * invdet;
  r.m[0][2] = c2 * invdet;
  r.m[1][0] = cofactor(a, 0, 1) * invdet;
  r.m[1][1] = cofactor(a, 1, 1) * invdet;
  r.m[1][2] = cofactor(a, 2, 1) * invdet;
  r.m[2][0] = cofactor(a, 0, 2) * invdet;
  r.m[2][1] = cofactor(a, 1, 2) * invdet;
  r.m[2][2] = cofactor(a, 2, 2) * invdet;
  return r;
}
static inline V3 homogeneous(V2 v) { return {v.x, v.y, 1.0}; }                     // types.h:37
static inline V2 dehomogeneous(V3 v) { return V2{v.x, v.y} / (v.z + EPS); }       // types.h:41-43

// ---------------------------------------------------------------------------------------------
// base/pose.cc, base/camera.{h,cc}, base/camera_view.cc
// ---------------------------------------------------------------------------------------------
struct Camera {                 // limap::Camera (colmap::Camera of a pinhole model)
  std::vector<double> params;   // fx, fy, cx, cy  (heap storage like colmap::Camera::params)
  M3 K() const {                // camera.h:72  (CalibrationMatrix of PINHOLE/SIMPLE_PINHOLE)
    M3 k;
    k.m[0][0] = params[0]; k.m[0][1] = 0.0;       k.m[0][2] = params[2];
    k.m[1][0] = 0.0;       k.m[1][1] = params[1]; k.m[1][2] = params[3];
    k.m[2][0] = 0.0;       k.m[2][1] = 0.0;       k.m[2][2] = 1.0;
    return k;
  }
  M3 K_inv() const { return inverse(K()); }  // camera.h:73
  double uncertainty(double depth, double var2d) const {  // camera.cc:228-242
    double f = (params[0] + params[1]) / 2.0;  // == f exactly for SIMPLE_PINHOLE (fx == fy)
    return var2d * depth / f;
  }
};

struct Pose {  // limap::CameraPose
  double q[4] = {1, 0, 0, 0};
  V3 t;
  M3 R() const {  // pose.cc:12-28 : re-normalise, then Eigen Quaternion::toRotationMatrix
    double n = std::sqrt((q[0] * q[0] + q[2] * q[2]) + (q[1] * q[1] + q[3] * q[3]));
    double w, x, y, z;
    if (n == 0) {
      w = 1.0; x = q[1]; y = q[2]; z = q[3];
    } else {
      w = q[0] / n; x = q[1] / n; y = q[2] / n; z = q[3] / n;
    }
    double tx = 2.0 * x, ty = 2.0 * y, tz = 2.0 * z;
    double twx = tx * w, twy = ty * w, twz = tz * w;
    double txx = tx * x, txy = ty * x, txz = tz * x;
    double tyy = ty * y, tyz = tz * y, tzz = tz * z;
    M3 r;
    r.m[0][0] = 1.0 - (tyy + tzz); r.m[0][1] = txy - twz;         r.m[0][2] = txz + twy;
    r.m[1][0] = txy + twz;         r.m[1][1] = 1.0 - (txx + tzz); r.m[1][2] = tyz - twx;
    r.m[2][0] = txz - twy;         r.m[2][1] = tyz + twx;         r.m[2][2] = 1.0 - (txx + tyy);
    return r;
  }
  V3 T() const { return t; }
  V3 center() const { return matvec(transpose(R()), -T()) ; }  // camera.h:106  (-R^T) * T
  double projdepth(V3 p) const {                                // camera.cc:276-279
    V3 pc = matvec(R(), p) + T();
    return pc.z;
  }
};

struct CameraView {  // limap::CameraView
  Camera cam;
  Pose pose;
  std::string name;  // image_name_ (copied by value with the view, image_collection.cc:366-371)
  M3 K() const { return cam.K(); }
  M3 K_inv() const { return cam.K_inv(); }
  M3 R() const { return pose.R(); }
  V3 T() const { return pose.T(); }
  V2 projection(V3 p) const {  // camera_view.cc:61-65
    V3 ph = matvec(K(), matvec(R(), p) + T());
    return dehomogeneous(ph);
  }
  V3 ray_direction(V2 p) const {  // camera_view.cc:67-69 : ((R^T K^-1) x~).normalized()
    return normalized(matvec(matmul(transpose(R()), K_inv()), homogeneous(p)));
  }
};
// NOTE on center(): Eigen evaluates `-R().transpose() * T()` as (-(R^T)) * T; negating every
// coefficient before the products is bit-identical to negating T (sign flips are exact).

// ---------------------------------------------------------------------------------------------
// base/linebase.{h,cc}
// ---------------------------------------------------------------------------------------------
struct Line2d {
  V2 start, end;
  double score = -1;
  double length() const { return norm(start - end); }                  // linebase.h:25
  V2 midpoint() const { return 0.5 * (start + end); }                  // linebase.h:26
  V2 direction() const { return normalized(end - start); }             // linebase.h:27
  V3 coords() const {                                                  // linebase.cc:35-39
    return normalized(cross(homogeneous(start), homogeneous(end)));
  }
};

struct Line3d {
  V3 start, end;
  double score = -1;
  double uncertainty = -1.0;
  double depths[2] = {-1, -1};  // left uninitialised by the reference's default ctor
  Line3d() {}
  Line3d(V3 s, V3 e, double sc = -1, double d0 = -1, double d1 = -1, double unc = -1)
      : start(s), end(e), score(sc), uncertainty(unc) {
    depths[0] = d0;
    depths[1] = d1;
  }
  double length() const { return norm(start - end); }
  V3 midpoint() const { return 0.5 * (start + end); }
  V3 direction() const { return normalized(end - start); }
  Line2d projection(const CameraView &view) const {  // linebase.cc:93-98
    Line2d l;
    l.start = view.projection(start);
    l.end = view.projection(end);
    return l;
  }
  double sensitivity(const CameraView &view) const {  // linebase.cc:100-107
    Line2d l2d = projection(view);
    V3 dir3d = view.ray_direction(l2d.midpoint());
    double cos_val = std::abs(dot(direction(), dir3d));
    double angle = std::acos(cos_val) * 180.0 / M_PI;
    return 90 - angle;
  }
  double computeUncertainty(const CameraView &view, double var2d) const {  // linebase.cc:109-116
    double d1 = view.pose.projdepth(start);
    double d2 = view.pose.projdepth(end);
    double d = (d1 + d2) / 2.0;
    return view.cam.uncertainty(d, var2d);
  }
};

// ---------------------------------------------------------------------------------------------
// base/line_dists.{h,cc}  (only the distances the path reaches)
// ---------------------------------------------------------------------------------------------
template <class L>
static double cosine(const L &l1, const L &l2) {  // line_dists.h:52-55
  return std::abs(dot(l1.direction(), l2.direction()));
}
template <class L>
static double compute_angle(const L &l1, const L &l2) {  // line_dists.h:62-66
  double c = cosine(l1, l2);
  return std::acos(c) * 180.0 / M_PI;
}
template <class L>
static std::pair<double, double> dists_endpoints_perpendicular_oneway(const L &l1, const L &l2) {
  // line_dists.h:98-111
  auto v2 = l2.direction();
  auto disps = l1.start - l2.start;
  double ds = dot(disps, v2);
  double d12s_sq = sqnorm(disps) - ds * ds;
  double d12s = std::sqrt(std::max(d12s_sq, 0.0));
  auto dispe = l1.end - l2.start;
  double de = dot(dispe, v2);
  double d12e_sq = sqnorm(dispe) - de * de;
  double d12e = std::sqrt(std::max(d12e_sq, 0.0));
  return {d12s, d12e};
}
template <class L>
static double dist_endpoints_perpendicular(const L &l1, const L &l2) {  // line_dists.h:122-133
  auto a = dists_endpoints_perpendicular_oneway(l1, l2);
  auto b = dists_endpoints_perpendicular_oneway(l2, l1);
  double d[4] = {a.first, a.second, b.first, b.second};
  return *std::max_element(d, d + 4);
}
static double dist_endpoints_scaleinv_oneway(const Line3d &l1, const Line3d &l2) {
  // line_dists.cc:55-60
  double dist_start = norm(l1.start - l2.start);
  double dist_end = norm(l1.end - l2.end);
  return std::max(dist_start / (l1.depths[0] + EPS), dist_end / (l1.depths[1] + EPS));
}
template <class L>
static bool get_innerseg(const L &l1, const L &l2, L &innerseg) {  // line_dists.h:159-176
  auto l1_dir = l1.direction();
  double denom = dot(l2.end - l2.start, l1_dir);
  double nume_start = dot(l1.start - l2.start, l1_dir);
  double t1 = nume_start / (denom + EPS);
  double nume_end = dot(l1.end - l2.start, l1_dir);
  double t2 = nume_end / (denom + EPS);
  if (t1 > t2) std::swap(t1, t2);
  if (t1 >= 1.0 || t2 <= 0.0) return false;
  innerseg.start = l2.start + (l2.end - l2.start) * std::max(t1, 0.0);
  innerseg.end = l2.start + (l2.end - l2.start) * std::min(t2, 1.0);
  return true;
}
template <class L>
static double dist_innerseg(const L &l1, const L &l2) {  // line_dists.h:178-187
  double MAX_DIST = std::numeric_limits<double>::max();
  L l1_inner, l2_inner;
  if (!get_innerseg(l2, l1, l1_inner)) return MAX_DIST;
  if (!get_innerseg(l1, l2, l2_inner)) return MAX_DIST;
  return dist_endpoints_perpendicular(l1_inner, l2_inner);
}
template <class L>
static double compute_overlap(const L &l1, const L &l2) {  // line_dists.h:189-200
  double len = l2.length();
  auto v = l2.direction();
  double p1 = dot(l1.start - l2.start, v) / len;
  double p2 = dot(l1.end - l2.start, v) / len;
  if (p1 > p2) std::swap(p1, p2);
  return std::min(p2, 1.0) - std::max(p1, 0.0);
}
template <class L>
static double compute_bioverlap(const L &l1, const L &l2) {  // line_dists.h:202-208
  double v1 = compute_overlap(l1, l2);
  double v2 = compute_overlap(l2, l1);
  return std::max(v1, v2);
}

// ---------------------------------------------------------------------------------------------
// base/line_linker.{h,cc}
// ---------------------------------------------------------------------------------------------
static double get_multiplier(double score_th) {  // line_linker.cc:9-12
  return 1.0 / std::sqrt(-std::log(score_th) * 2.0);
}
static double expscore(double val, double sigma) {  // line_linker.cc:15-17 ; pow(x,2) == x*x
  double q = val / sigma;
  return std::exp(-(q * q) / 2.0);
}

struct Linker2dCfg {  // line_linker.h:18-52
  double score_th = 0.5, th_angle = 8.0, th_overlap = 0.1, th_smartoverlap = 0.2,
         th_smartangle = 1.0, th_perp = 5.0, th_innerseg = 5.0;
  bool use_angle = true, use_overlap = true, use_smartangle = true, use_perp = true,
       use_innerseg = false;
  double multiplier() const { return get_multiplier(score_th); }
};
struct Linker3dCfg {  // line_linker.h:88-151
  double score_th = 0.5, th_angle = 10.0, th_overlap = 0.01, th_smartoverlap = 0.1,
         th_smartangle = 1.0, th_perp = 0.02, th_innerseg = 0.02, th_scaleinv = 0.01;
  bool use_angle = true, use_overlap = true, use_smartangle = true, use_perp = false,
       use_innerseg = true, use_scaleinv = false;
  double multiplier() const { return get_multiplier(score_th); }
  void set_to_shared_parent_scoring() {  // line_linker.h:115-121
    use_angle = true; use_overlap = false; use_perp = false; use_innerseg = false;
    use_scaleinv = true;
  }
  void set_to_spatial_merging() {  // line_linker.h:123-129
    use_angle = true; use_overlap = true; use_perp = false; use_innerseg = true;
    use_scaleinv = false;
  }
  void set_to_avgtest_merging() {  // line_linker.h:131-137
    use_angle = true; use_overlap = false; use_perp = true; use_innerseg = false;
    use_scaleinv = false;
  }
};

struct Linker2d {
  Linker2dCfg config;
  double score_angle(const Line2d &l1, const Line2d &l2) const {  // line_linker.cc:40-47
    double angle = compute_angle(l1, l2);
    double s = expscore(angle, config.th_angle * config.multiplier());
    if (s < config.score_th) s = 0.0;
    return s;
  }
  double score_smartangle(const Line2d &l1, const Line2d &l2) const {  // line_linker.cc:49-65
    double angle = compute_angle(l1, l2);
    double th_angle = config.th_angle;
    double overlap = compute_bioverlap(l1, l2);
    if (overlap < config.th_smartoverlap) {
      double ratio =
          (config.th_smartoverlap - overlap) / (config.th_smartoverlap - config.th_overlap);
      ratio = std::min(ratio, 1.0);
      th_angle = config.th_angle - ratio * (config.th_angle - config.th_smartangle);
    }
    double s = expscore(angle, th_angle * config.multiplier());
    if (s < config.score_th) s = 0.0;
    return s;
  }
  double score_overlap(const Line2d &l1, const Line2d &l2) const {  // line_linker.cc:78-85
    double overlap = compute_bioverlap(l1, l2);
    return overlap > config.th_overlap ? 1.0 : 0.0;
  }
  double score_perp(const Line2d &l1, const Line2d &l2) const {  // line_linker.cc:92-99
    double dist = dist_endpoints_perpendicular(l1, l2);
    double s = expscore(dist, config.th_perp * config.multiplier());
    if (s < config.score_th) s = 0.0;
    return s;
  }
  double score_innerseg(const Line2d &l1, const Line2d &l2) const {  // line_linker.cc:106-113
    double dist = dist_innerseg(l1, l2);
    double s = expscore(dist, config.th_innerseg * config.multiplier());
    if (s < config.score_th) s = 0.0;
    return s;
  }
  double compute_score(const Line2d &l1, const Line2d &l2) const {  // line_linker.cc:139-160
    double score = 1.0;
    if (config.use_angle) score = std::min(score, score_angle(l1, l2));
    if (score < config.score_th) return score;
    if (config.use_overlap) score = std::min(score, score_overlap(l1, l2));
    if (score < config.score_th) return score;
    if (config.use_angle && config.use_overlap && config.use_smartangle)
      score = std::min(score, score_smartangle(l1, l2));
    if (score < config.score_th) return score;
    if (config.use_perp) score = std::min(score, score_perp(l1, l2));
    if (score < config.score_th) return score;
    if (config.use_innerseg) score = std::min(score, score_innerseg(l1, l2));
    return score;
  }
};

struct Linker3d {
  Linker3dCfg config;
  double score_angle(const Line3d &l1, const Line3d &l2) const {  // line_linker.cc:185-192
    double angle = compute_angle(l1, l2);
    double s = expscore(angle, config.th_angle * config.multiplier());
    if (s < config.score_th) s = 0.0;
    return s;
  }
  double score_smartangle(const Line3d &l1, const Line3d &l2) const {  // line_linker.cc:194-210
    double angle = compute_angle(l1, l2);
    double th_angle = config.th_angle;
    double overlap = compute_bioverlap(l1, l2);
    if (overlap < config.th_smartoverlap) {
      double ratio =
          (config.th_smartoverlap - overlap) / (config.th_smartoverlap - config.th_overlap);
      ratio = std::min(ratio, 1.0);
      th_angle = config.th_angle - ratio * (config.th_angle - config.th_smartangle);
    }
    double s = expscore(angle, th_angle * config.multiplier());
    if (s < config.score_th) s = 0.0;
    return s;
  }
  double score_overlap(const Line3d &l1, const Line3d &l2) const {  // line_linker.cc:223-230
    double overlap = compute_bioverlap(l1, l2);
    return overlap > config.th_overlap ? 1.0 : 0.0;
  }
  double score_perp(const Line3d &l1, const Line3d &l2) const {  // line_linker.cc:237-246
    double dist = dist_endpoints_perpendicular(l1, l2);
    double unc = std::min(l1.uncertainty, l2.uncertainty);
    double s = expscore(dist, config.th_perp * unc * config.multiplier());
    if (s < config.score_th) s = 0.0;
    return s;
  }
  double score_innerseg(const Line3d &l1, const Line3d &l2) const {  // line_linker.cc:253-262
    double dist = dist_innerseg(l1, l2);
    double unc = std::min(l1.uncertainty, l2.uncertainty);
    double s = expscore(dist, config.th_innerseg * unc * config.multiplier());
    if (s < config.score_th) s = 0.0;
    return s;
  }
  double score_scaleinv(const Line3d &l1, const Line3d &l2) const {  // line_linker.cc:269-277
    double dist = dist_endpoints_scaleinv_oneway(l1, l2);
    double s = expscore(dist, config.th_scaleinv * config.multiplier());
    if (s < config.score_th) s = 0.0;
    return s;
  }
  bool check_connection(const Line3d &l1, const Line3d &l2) const {  // line_linker.cc:285-304
    if (config.use_angle)  // check_connection_angle, :212-216: plain angle <= th_angle
      if (!(compute_angle(l1, l2) <= config.th_angle)) return false;
    if (config.use_overlap)  // :232-235
      if (!(score_overlap(l1, l2) == 1.0)) return false;
    if (config.use_angle && config.use_overlap && config.use_smartangle)  // :218-221
      if (!(score_smartangle(l1, l2) >= config.score_th)) return false;
    if (config.use_perp)  // :248-251
      if (!(score_perp(l1, l2) >= config.score_th)) return false;
    if (config.use_innerseg)  // :264-267
      if (!(score_innerseg(l1, l2) >= config.score_th)) return false;
    if (config.use_scaleinv)  // :279-282
      if (!(score_scaleinv(l1, l2) >= config.score_th)) return false;
    return true;
  }
  double compute_score(const Line3d &l1, const Line3d &l2) const {  // line_linker.cc:306-331
    double score = 1.0;
    if (config.use_angle) score = std::min(score, score_angle(l1, l2));
    if (score < config.score_th) return score;
    if (config.use_overlap) score = std::min(score, score_overlap(l1, l2));
    if (score < config.score_th) return score;
    if (config.use_angle && config.use_overlap && config.use_smartangle)
      score = std::min(score, score_smartangle(l1, l2));
    if (score < config.score_th) return score;
    if (config.use_perp) score = std::min(score, score_perp(l1, l2));
    if (score < config.score_th) return score;
    if (config.use_innerseg) score = std::min(score, score_innerseg(l1, l2));
    if (score < config.score_th) return score;
    if (config.use_scaleinv) score = std::min(score, score_scaleinv(l1, l2));
    return score;
  }
};

// ---------------------------------------------------------------------------------------------
// triangulation/functions.cc
// ---------------------------------------------------------------------------------------------
static bool test_line_inside_ranges(const Line3d &line, const std::pair<V3, V3> &r) {  // :8-26
  if (line.start.x < r.first.x || line.start.x > r.second.x) return false;
  if (line.start.y < r.first.y || line.start.y > r.second.y) return false;
  if (line.start.z < r.first.z || line.start.z > r.second.z) return false;
  if (line.end.x < r.first.x || line.end.x > r.second.x) return false;
  if (line.end.y < r.first.y || line.end.y > r.second.y) return false;
  if (line.end.z < r.first.z || line.end.z > r.second.z) return false;
  return true;
}

static V3 getNormalDirection(const Line2d &l, const CameraView &view) {  // functions.cc:28-35
  const M3 K_inv = view.K_inv();
  const M3 R = view.R();
  const M3 RtKinv = matmul(transpose(R), K_inv);
  V3 c_start = matvec(RtKinv, V3{l.start.x, l.start.y, 1});
  V3 c_end = matvec(RtKinv, V3{l.end.x, l.end.y, 1});
  V3 n = cross(c_start, c_end);
  return normalized(n);
}

static M3 compute_essential_matrix(const CameraView &view1, const CameraView &view2) {  // :44-67
  const M3 R1 = view1.R();
  const V3 T1 = view1.T();
  const M3 R2 = view2.R();
  const V3 T2 = view2.T();
  M3 relR = matmul(R2, transpose(R1));
  V3 relT = T2 - matvec(relR, T1);
  M3 tskew;
  tskew.m[0][0] = 0.0;     tskew.m[0][1] = -relT.z; tskew.m[0][2] = relT.y;
  tskew.m[1][0] = relT.z;  tskew.m[1][1] = 0.0;     tskew.m[1][2] = -relT.x;
  tskew.m[2][0] = -relT.y; tskew.m[2][1] = relT.x;  tskew.m[2][2] = 0.0;
  return matmul(tskew, relR);
}

static M3 compute_fundamental_matrix(const CameraView &view1, const CameraView &view2) {  // :69-74
  M3 E = compute_essential_matrix(view1, view2);
  return matmul(matmul(transpose(view2.K_inv()), E), view1.K_inv());
}

static double compute_epipolar_IoU(const Line2d &l1, const CameraView &view1, const Line2d &l2,
                                   const CameraView &view2) {  // functions.cc:76-98
  M3 F = compute_fundamental_matrix(view1, view2);
  V3 coor_l2 = l2.coords();
  V3 coor_epline_start = normalized(matvec(F, V3{l1.start.x, l1.start.y, 1}));
  V3 homo_c_start = cross(coor_l2, coor_epline_start);
  V2 c_start = dehomogeneous(homo_c_start);
  V3 coor_epline_end = normalized(matvec(F, V3{l1.end.x, l1.end.y, 1}));
  V3 homo_c_end = cross(coor_l2, coor_epline_end);
  V2 c_end = dehomogeneous(homo_c_end);
  double c1 = dot(c_start - l2.start, l2.direction()) / l2.length();
  double c2 = dot(c_end - l2.start, l2.direction()) / l2.length();
  if (c1 > c2) std::swap(c1, c2);
  double IoU = (std::min(c2, 1.0) - std::max(c1, 0.0)) / (std::max(c2, 1.0) - std::min(c1, 0.0));
  return IoU;
}

static std::pair<V3, bool> triangulate_point(const V2 &p1, const CameraView &view1, const V2 &p2,
                                             const CameraView &view2) {  // functions.cc:100-117
  V3 C1 = view1.pose.center();
  V3 C2 = view2.pose.center();
  V3 n1e = view1.ray_direction(p1);
  V3 n2e = view2.ray_direction(p2);
  double a00 = dot(n1e, n1e), a01 = -dot(n1e, n2e), a10 = -dot(n2e, n1e), a11 = dot(n2e, n2e);
  double b0 = dot(n1e, C2 - C1);
  double b1 = dot(n2e, C1 - C2);
  // Eigen A.ldlt().solve(b) on a 2x2 (lower triangle, pivoting on the larger diagonal entry).
  // Both diagonal entries are 1 up to rounding; restated without pivot swap when a00 >= a11.
  double x0, x1;
  {
    bool swap = std::abs(a11) > std::abs(a00);  // LDLT pivots on the biggest |diagonal|
    double d0 = swap ? a11 : a00, d1 = swap ? a00 : a11, off = a10;
    double r0 = swap ? b1 : b0, r1 = swap ? b0 : b1;
    double l10 = off / d0;
    double dd1 = d1 - l10 * (d0 * l10);  // Eigen ldlt_inplace: mat(1,1) -= A10 * (D * A10^T)
    double y0 = r0;
    double y1 = r1 - l10 * y0;
    double z0 = y0 / d0, z1 = y1 / dd1;
    double s1 = z1;
    double s0 = z0 - l10 * s1;
    x0 = swap ? s1 : s0;
    x1 = swap ? s0 : s1;
    (void)a01;
  }
  V3 point = 0.5 * (((n1e * x0 + C1) + n2e * x1) + C2);
  if (view1.pose.projdepth(point) < EPS || view2.pose.projdepth(point) < EPS)
    return {V3{0, 0, 0}, false};
  return {point, true};
}

static const Line3d kSentinel() { return Line3d(V3{0, 0, 0}, V3{1, 1, 1}, -1.0); }

static Line3d triangulate_line_by_endpoints(const Line2d &l1, const CameraView &view1,
                                            const Line2d &l2,
                                            const CameraView &view2) {  // functions.cc:172-190
  auto rs = triangulate_point(l1.start, view1, l2.start, view2);
  if (!rs.second) return kSentinel();
  V3 pstart = rs.first;
  auto re = triangulate_point(l1.end, view1, l2.end, view2);
  if (!re.second) return kSentinel();
  V3 pend = re.first;
  double z_start = view1.pose.projdepth(pstart);
  double z_end = view1.pose.projdepth(pend);
  return Line3d(pstart, pend, 1.0, z_start, z_end);
}

static std::pair<Line3d, bool> line_triangulation(const Line2d &l1, const CameraView &view1,
                                                  const Line2d &l2,
                                                  const CameraView &view2) {  // functions.cc:194-233
  V3 c1_start = view1.ray_direction(l1.start);
  V3 c1_end = view1.ray_direction(l1.end);
  V3 c2_start = view2.ray_direction(l2.start);
  V3 c2_end = view2.ray_direction(l2.end);
  V3 B = view2.pose.center() - view1.pose.center();

  auto solve0 = [&](V3 c1) {  // (A.inverse() * B)[0], A = [c1, -c2_start, -c2_end] (columns)
    M3 A;
    A.m[0][0] = c1.x; A.m[0][1] = -c2_start.x; A.m[0][2] = -c2_end.x;
    A.m[1][0] = c1.y; A.m[1][1] = -c2_start.y; A.m[1][2] = -c2_end.y;
    A.m[2][0] = c1.z; A.m[2][1] = -c2_start.z; A.m[2][2] = -c2_end.z;
    M3 Ai = inverse(A);
    return (Ai.m[0][0] * B.x + Ai.m[0][1] * B.y) + Ai.m[0][2] * B.z;
  };
  double res_start0 = solve0(c1_start);
  V3 l3d_start = c1_start * res_start0 + view1.pose.center();
  double z_start = view1.pose.projdepth(l3d_start);
  double res_end0 = solve0(c1_end);
  V3 l3d_end = c1_end * res_end0 + view1.pose.center();
  double z_end = view1.pose.projdepth(l3d_end);

  if (z_start < EPS || z_end < EPS) return {Line3d(), false};
  double d21 = view2.pose.projdepth(l3d_start);
  double d22 = view2.pose.projdepth(l3d_end);
  if (d21 < EPS || d22 < EPS) return {Line3d(), false};
  if (std::isnan(l3d_start.x) || std::isnan(l3d_end.x)) return {Line3d(), false};
  return {Line3d(l3d_start, l3d_end, 1.0, z_start, z_end), true};
}

static V3 getDirectionFromVP(const V3 &vp, const CameraView &view) {  // functions.cc:37-42
  const M3 K_inv = view.K_inv();
  const M3 R = view.R();
  V3 direc = matvec(matmul(transpose(R), K_inv), vp);
  return normalized(direc);
}

// base/infinite_line.h: Pluecker line (direction d, moment m = p x d)
struct InfiniteLine3d {
  V3 d, m;
  InfiniteLine3d(const V3 &p, const V3 &direc) : d(direc), m(cross(p, direc)) {}  // infinite_line.cc:55-59
  // infinite_line.cc:151-163: the point of THIS line closest to `line`
  V3 project_from_infinite_line(const InfiniteLine3d &line) const {
    V3 l1 = d, m1 = m, l2 = line.d, m2 = line.m;
    V3 cr = cross(l1, l2);
    V3 p = cross(m1, cross(l2, cr)) * (-1.0) + l1 * dot(m2, cr);
    p = p / sqnorm(cr);
    return p;
  }
  V3 project_to_infinite_line(const InfiniteLine3d &line) const { return line.project_from_infinite_line(*this); }
};

// unproject endpoints with known infinite line -- functions.cc:306-321
static Line3d triangulate_line_with_infinite_line(const Line2d &l1, const CameraView &view1,
                                                  const InfiniteLine3d &inf_line) {
  InfiniteLine3d ray1_start(view1.pose.center(), view1.ray_direction(l1.start));
  V3 pstart = inf_line.project_to_infinite_line(ray1_start);
  double z_start = view1.pose.projdepth(pstart);
  InfiniteLine3d ray1_end(view1.pose.center(), view1.ray_direction(l1.end));
  V3 pend = inf_line.project_to_infinite_line(ray1_end);
  double z_end = view1.pose.projdepth(pend);
  if (z_start < EPS || z_end < EPS) return kSentinel();
  return Line3d(pstart, pend, 1.0, z_start, z_end);
}

// ---------------------------------------------------------------------------------------------
// triangulate_line_with_one_point (functions.cc:325-383 + solvers/triangulation).
// The reference's solver is ~600 lines of machine-generated coefficients of a quartic in a Lagrange
// multiplier, solved by PoseLib's univariate::solve_quartic_real (a dependency that is not in the tree).
// TWO forms live here.  (1) solve_line_with_one_point_generated below: the generated polynomials as data, evaluated
// term by term in the reference's order -- the default, bit-identical to oracle/_ref.  (2) The same optimisation
// problem restated from its definition (the comments at the end of that file), which is what the device code follows
// (ora_set_one_point_solver(0); tests/test_oracle_kat.py ties the two together to the conditioning of form 1):
// in the 2D coordinates of plane 1, with p1, p2 the unit directions of l1's
// endpoint rays, p the known point and (lx, ly, lz) the trace of plane 2,
//     minimise (l . x1)^2 + (l . x2)^2   over x1 = lambda1 p1, x2 = lambda2 p2,
//     subject to x1, x2, p collinear, lambda1 > 0, lambda2 > 0.
// Collinearity reads lambda1 lambda2 c - lambda1 a - lambda2 b = 0 with c = p1 x p2, a = p1 x p, b = p x p2,
// so lambda2 = a u / (c u - b), u = lambda1; the stationary points of the objective in u are the real roots
// of the quartic  alpha1 (alpha1 u + lz) w^3 - alpha2 a b (alpha2 a u + lz w) = 0,  w = c u - b,
// alpha_k = lx p_kx + ly p_ky, and the solution is the admissible one with the smallest objective -- the
// same stationary points and the same selection rule as the reference, up to rounding (parity for this
// proposal is therefore a tolerance, like the SVD stand-in of the line fit).
// ---------------------------------------------------------------------------------------------
static int real_roots_cubic_monic(double A, double B, double C, double out[3]) {  // z^3 + A z^2 + B z + C
  const double sh = A / 3.0;
  const double P = B - A * A / 3.0;
  const double Q = 2.0 * A * A * A / 27.0 - A * B / 3.0 + C;
  const double disc = Q * Q / 4.0 + P * P * P / 27.0;
  if (disc > 0) {
    const double sq = std::sqrt(disc);
    const double u = std::cbrt(-Q / 2.0 + sq), v = std::cbrt(-Q / 2.0 - sq);
    out[0] = u + v - sh;
    return 1;
  }
  if (P == 0.0) {
    out[0] = -sh;
    return 1;
  }
  const double m = 2.0 * std::sqrt(-P / 3.0);
  double arg = 3.0 * Q / (P * m);
  arg = arg < -1.0 ? -1.0 : (arg > 1.0 ? 1.0 : arg);
  const double th = std::acos(arg) / 3.0;
  for (int k = 0; k < 3; ++k) out[k] = m * std::cos(th - 2.0 * M_PI * k / 3.0) - sh;
  return 3;
}

static int real_roots_quadratic(double a, double b, double c, double out[2]) {  // a x^2 + b x + c, a != 0
  const double d = b * b - 4.0 * a * c;
  if (d < 0) return 0;
  const double sq = std::sqrt(d);
  const double q = -0.5 * (b + (b >= 0 ? sq : -sq));
  int n = 0;
  out[n++] = q / a;
  if (q != 0.0) out[n++] = c / q;
  else out[n++] = 0.0;
  return n;
}

// real roots of A4 x^4 + ... + A0 (degree detected relative to the largest coefficient), Newton-polished
static int real_roots_poly4(const double Ain[5], double out[4]) {
  double A[5];
  double mx = 0;
  for (int k = 0; k < 5; ++k) mx = std::max(mx, std::abs(Ain[k]));
  if (!(mx > 0) || !std::isfinite(mx)) return 0;
  for (int k = 0; k < 5; ++k) A[k] = Ain[k] / mx;
  int deg = 4;
  while (deg > 0 && std::abs(A[deg]) < 1e-13) --deg;
  int n = 0;
  if (deg == 0) return 0;
  if (deg == 1) {
    out[n++] = -A[0] / A[1];
  } else if (deg == 2) {
    n = real_roots_quadratic(A[2], A[1], A[0], out);
  } else if (deg == 3) {
    n = real_roots_cubic_monic(A[2] / A[3], A[1] / A[3], A[0] / A[3], out);
  } else {
    const double a = A[3] / A[4], b = A[2] / A[4], c = A[1] / A[4], d = A[0] / A[4];
    const double p = b - 3.0 * a * a / 8.0;
    const double q = c - a * b / 2.0 + a * a * a / 8.0;
    const double r = d - a * c / 4.0 + a * a * b / 16.0 - 3.0 * a * a * a * a / 256.0;
    double ys[4];
    int ny = 0;
    if (std::abs(q) < 1e-14 * (1.0 + std::abs(p) + std::abs(r))) {  // biquadratic
      double t[2];
      int nt = real_roots_quadratic(1.0, p, r, t);
      for (int k = 0; k < nt; ++k)
        if (t[k] >= 0) {
          const double sq = std::sqrt(t[k]);
          ys[ny++] = sq;
          ys[ny++] = -sq;
        }
    } else {
      double z[3];
      int nz = real_roots_cubic_monic(2.0 * p, p * p - 4.0 * r, -q * q, z);
      double z0 = z[0];
      for (int k = 1; k < nz; ++k) z0 = std::max(z0, z[k]);
      if (z0 > 0) {
        const double sgm = std::sqrt(z0);
        double t[2];
        int nt = real_roots_quadratic(1.0, sgm, (p + z0 - q / sgm) / 2.0, t);
        for (int k = 0; k < nt; ++k) ys[ny++] = t[k];
        nt = real_roots_quadratic(1.0, -sgm, (p + z0 + q / sgm) / 2.0, t);
        for (int k = 0; k < nt; ++k) ys[ny++] = t[k];
      }
    }
    for (int k = 0; k < ny; ++k) out[n++] = ys[k] - a / 4.0;
  }
  for (int k = 0; k < n; ++k) {  // Newton polish on the scaled polynomial
    double x = out[k];
    for (int it = 0; it < 3; ++it) {
      const double f = (((A[4] * x + A[3]) * x + A[2]) * x + A[1]) * x + A[0];
      const double df = ((4.0 * A[4] * x + 3.0 * A[3]) * x + 2.0 * A[2]) * x + A[1];
      if (df == 0.0 || !std::isfinite(f / df)) break;
      x = x - f / df;
    }
    out[k] = x;
  }
  return n;
}

static std::pair<double, double> solve_line_with_one_point(double lx, double ly, double lz, V2 p, V2 p1, V2 p2) {
  const double al1 = lx * p1.x + ly * p1.y, al2 = lx * p2.x + ly * p2.y;
  const double c = p1.x * p2.y - p1.y * p2.x, a = p1.x * p.y - p1.y * p.x, b = p.x * p2.y - p.y * p2.x;
  double A[5];
  A[4] = al1 * al1 * c * c * c;
  A[3] = al1 * (lz * c * c * c - 3.0 * al1 * c * c * b);
  A[2] = al1 * (3.0 * al1 * c * b * b - 3.0 * lz * c * c * b);
  A[1] = al1 * (3.0 * lz * c * b * b - al1 * b * b * b) - al2 * a * b * (al2 * a + lz * c);
  A[0] = lz * b * b * (al2 * a - al1 * b);
  double roots[4];
  const int n = real_roots_poly4(A, roots);
  std::pair<double, double> best{-1.0, -1.0};
  double best_err = std::numeric_limits<double>::max();
  for (int k = 0; k < n; ++k) {
    double u = roots[k];
    // Newton on the factored form: the expanded coefficients carry the cancellation of (c u - b)^3
    for (int it = 0; it < 4; ++it) {
      const double w = c * u - b;
      const double f = al1 * (al1 * u + lz) * (w * w * w) - al2 * a * b * (al2 * a * u + lz * w);
      const double df = al1 * al1 * (w * w * w) + 3.0 * al1 * (al1 * u + lz) * (w * w) * c - al2 * a * b * (al2 * a + lz * c);
      if (df == 0.0 || !std::isfinite(f / df)) break;
      u = u - f / df;
    }
    const double w = c * u - b;
    if (w == 0.0) continue;
    const double lam2 = a * u / w;
    if (!(u > 0) || !(lam2 > 0)) continue;  // cheirality (lambda <= 0 is skipped in the reference)
    const double e1 = al1 * u + lz, e2 = al2 * lam2 + lz;
    const double err = e1 * e1 + e2 * e2;
    if (err < best_err) {
      best_err = err;
      best = {u, lam2};
    }
  }
  return best;
}

// The reference's solver itself, term by term (solvers/triangulation/triangulate_line_with_one_point.cc:11-645): the
// generated polynomials are DATA here (onepoint_terms.inc, written by oracle/make_onepoint_terms.py from that file) and
// are evaluated in the reference's order -- a term left to right ((coef * f1) * f2) ..., a sum left to right,
// std::pow(x, 2) as x * x (GCC folds it), higher powers through libm -- followed by the same root loop (:556-644).
// poselib::univariate::solve_quartic_real is the stand-in the reference is compiled against here (PoseLib is not on
// disk).  Bit-identical to oracle/_ref (tests/test_oracle_vs_ref.py).  The expanded coefficients cancel over ~10
// digits, so their value -- and the reference's result -- moves by ~1e-6 relative with the last bit of a libm pow();
// the restated form above is what the device code follows, the two are compared with that tolerance on the CPU.
#include "onepoint_terms.inc"
#include "ref_shim/PoseLib/misc/univariate.h"
static int g_one_point_generated = 1;  // ora_set_one_point_solver: 1 = the reference's generated solver, 0 = restated problem
static inline double onepoint_pow(double x, int k) { return k == 1 ? x : (k == 2 ? x * x : std::pow(x, (double)k)); }
static double onepoint_eval(const OnePointTerm *T, int n, const double v[10]) {
  double acc = 0.0;
  for (int i = 0; i < n; ++i) {
    const OnePointTerm &t = T[i];
    double x;
    int k = 0;
    if (t.has_coef) {
      x = (double)(t.coef < 0 ? -t.coef : t.coef);
    } else {
      x = onepoint_pow(v[t.f[0] >> 4], t.f[0] & 15);
      k = 1;
    }
    for (; k < t.n; ++k) x = x * onepoint_pow(v[t.f[k] >> 4], t.f[k] & 15);
    if (i == 0) acc = t.coef < 0 ? -x : x;
    else acc = t.coef < 0 ? acc - x : acc + x;
  }
  return acc;
}
#define ONEPOINT_EVAL(name, v) onepoint_eval(kOnePoint_##name, (int)(sizeof(kOnePoint_##name) / sizeof(OnePointTerm)), v)
static std::pair<double, double> solve_line_with_one_point_generated(double lx, double ly, double lz, V2 p, V2 p1, V2 p2) {
  double v[10] = {lx, ly, lz, p.x, p.y, p1.x, p1.y, p2.x, p2.y, 0.0};
  const double c4 = ONEPOINT_EVAL(c4, v);
  const double c3 = 0.0;
  const double c2 = ONEPOINT_EVAL(c2, v);
  const double c1 = ONEPOINT_EVAL(c1, v);
  const double c0 = ONEPOINT_EVAL(c0, v);
  double mu_sols[4];
  const int sols = poselib::univariate::solve_quartic_real(c3 / c4, c2 / c4, c1 / c4, c0 / c4, mu_sols);
  std::pair<double, double> best{-1.0, -1.0};
  double best_err = std::numeric_limits<double>::max();
  for (int k = 0; k < sols; ++k) {
    v[9] = mu_sols[k];
    const double lambda_denom = ONEPOINT_EVAL(lambda_denom, v);
    const double lambda_1 = ONEPOINT_EVAL(lambda_1, v) / lambda_denom;
    const double lambda_2 = ONEPOINT_EVAL(lambda_2, v) / lambda_denom;
    if (lambda_1 <= 0 || lambda_2 <= 0) continue;  // cheirality
    const double err1 = lx * (lambda_1 * p1.x) + ly * (lambda_1 * p1.y) + lz;
    const double err2 = lx * (lambda_2 * p2.x) + ly * (lambda_2 * p2.y) + lz;
    const double err = err1 * err1 + err2 * err2;
    if (err < best_err) {
      best_err = err;
      best = {lambda_1, lambda_2};
    }
  }
  return best;
}

// Triangulation with a known point, asymmetric perspective to (view1, l1) -- functions.cc:325-383
static Line3d triangulate_line_with_one_point(const Line2d &l1, const CameraView &view1, const Line2d &l2,
                                              const CameraView &view2, const V3 &point) {
  V3 n1 = getNormalDirection(l1, view1);
  V3 C1 = view1.pose.center();
  V3 p = point - n1 * dot(n1, point - C1);
  V3 v1s = view1.ray_direction(l1.start);
  V3 v1e = view1.ray_direction(l1.end);
  V3 n2 = getNormalDirection(l2, view2);
  double alpha = (-1) * dot(n2, view2.pose.center());
  // plane-1 frame: columns v1s, the part of v1e orthogonal to it, their cross product
  V3 r0 = v1s;
  V3 r1 = normalized(v1e - v1s * dot(v1s, v1e));
  V3 r2 = normalized(cross(r0, r1));
  V3 t = C1;
  auto Rt = [&](const V3 &x) { return V3{dot(r0, x), dot(r1, x), dot(r2, x)}; };
  V3 v2_t = Rt(v1e);
  V3 p_t = Rt(p - C1);
  V3 n2_t = Rt(n2);
  double alpha_t = alpha + dot(n2, t);
  V2 input_p{p_t.x, p_t.y};
  V2 input_v1 = normalized(V2{1.0, 0.0});
  V2 input_v2 = normalized(V2{v2_t.x, v2_t.y});
  auto res = g_one_point_generated ? solve_line_with_one_point_generated(n2_t.x, n2_t.y, alpha_t, input_p, input_v1, input_v2)
                                   : solve_line_with_one_point(n2_t.x, n2_t.y, alpha_t, input_p, input_v1, input_v2);
  if (res.first < 0 || res.second < 0) return kSentinel();
  V2 ls2 = input_v1 * res.first, le2 = input_v2 * res.second;
  V3 lstart = r0 * ls2.x + r1 * ls2.y + t;  // R * (x, y, 0) + t
  V3 lend = r0 * le2.x + r1 * le2.y + t;
  double z_start = view1.pose.projdepth(lstart);
  double z_end = view1.pose.projdepth(lend);
  if (z_start < EPS || z_end < EPS) return kSentinel();
  double d21 = view2.pose.projdepth(lstart);
  double d22 = view2.pose.projdepth(lend);
  if (d21 < EPS || d22 < EPS) return kSentinel();
  return Line3d(lstart, lend, 1.0, z_start, z_end);
}

// Triangulation with known direction, asymmetric perspective to (view1, l1) -- functions.cc:385-442
static Line3d triangulate_line_with_direction(const Line2d &l1, const CameraView &view1, const Line2d &l2,
                                              const CameraView &view2, const V3 &direction) {
  // Step 1: project direction onto plane 1
  V3 n1 = getNormalDirection(l1, view1);
  V3 direc = direction - n1 * dot(n1, direction);
  if (norm(direc) < EPS) return kSentinel();
  direc = normalized(direc);
  // Step 2: parameterize on plane 1 (a1s * d1s - a1e * d1e = 0)
  V3 perp_direc = cross(n1, direc);
  V3 v1s = view1.ray_direction(l1.start);
  double a1s = dot(v1s, perp_direc);
  V3 v1e = view1.ray_direction(l1.end);
  double a1e = dot(v1e, perp_direc);
  const double MIN_VALUE = 0.001;
  if (a1s < 0) {
    a1s *= -1;
    a1e *= -1;
  }
  if (a1s < MIN_VALUE || a1e < MIN_VALUE) return kSentinel();
  // Step 3: min [(c1s * d1s - b)^2 + (c1e * d1e - b)^2]
  V3 C1 = view1.pose.center();
  V3 C2 = view2.pose.center();
  V3 n2 = getNormalDirection(l2, view2);
  double c1s = dot(n2, v1s);
  double c1e = dot(n2, v1e);
  double b = dot(n2, C2 - C1);
  double c1 = c1s;
  double c2 = c1e * a1s / a1e;
  double d1s_num = (c1 + c2) * b;
  double d1s_denom = (c1 * c1 + c2 * c2);
  double d1s = d1s_num / d1s_denom;
  double d1e = d1s * a1s / a1e;
  V3 lstart = v1s * d1s + C1;
  V3 lend = v1e * d1e + C1;
  double z_start = view1.pose.projdepth(lstart);
  double z_end = view1.pose.projdepth(lend);
  if (z_start < EPS || z_end < EPS) return kSentinel();
  double d21 = view2.pose.projdepth(lstart);
  double d22 = view2.pose.projdepth(lend);
  if (d21 < EPS || d22 < EPS) return kSentinel();
  if (std::isnan(lstart.x) || std::isnan(lend.x)) return kSentinel();
  return Line3d(lstart, lend, 1.0, z_start, z_end);
}

static Line3d triangulate_line(const Line2d &l1, const CameraView &view1, const Line2d &l2,
                               const CameraView &view2) {  // functions.cc:295-304
  auto res = line_triangulation(l1, view1, l2, view2);
  if (!res.second) return kSentinel();
  return res.first;
}

// ---------------------------------------------------------------------------------------------
// base/graph.cc (subset), merging/merging.cc:18-103, merging/aggregator.cc
// ---------------------------------------------------------------------------------------------
struct Graph {
  std::vector<std::pair<int, int>> nodes;  // (image_idx, line_idx) in creation order
  std::map<std::pair<int, int>, int> node_map;
  std::vector<std::tuple<double, int, int>> edges;  // (sim, node1, node2) in insertion order
  int FindOrCreateNode(int img, int line) {  // graph.cc:57-71
    auto it = node_map.find({img, line});
    if (it != node_map.end()) return it->second;
    nodes.push_back({img, line});
    int idx = int(nodes.size()) - 1;
    node_map.insert({{img, line}, idx});
    return idx;
  }
  void AddEdge(int n1, int n2, double sim) { edges.push_back({sim, n1, n2}); }  // graph.cc:81-87
};

static int union_find_get_root(int node_idx, std::vector<int> &parent) {  // graph.cc:156-165
  if (parent[node_idx] == -1) return node_idx;
  parent[node_idx] = union_find_get_root(parent[node_idx], parent);
  return parent[node_idx];
}

static std::vector<int> ComputeLineTrackLabelsGreedy(const std::vector<int> &node_img,
                                                     std::vector<std::tuple<double, int, int>> edges) {
  // merging/merging.cc:18-103
  const size_t n_nodes = node_img.size();
  std::sort(edges.begin(), edges.end());
  std::reverse(edges.begin(), edges.end());
  std::vector<int> parent_nodes(n_nodes, -1);
  std::vector<std::set<int>> images_in_track(n_nodes);
  for (size_t i = 0; i < n_nodes; ++i) images_in_track[i].insert(node_img[i]);
  // (nodes_in_track of the reference is maintained but never read: omitted)
  for (const auto &e : edges) {
    int node_idx1 = std::get<1>(e), node_idx2 = std::get<2>(e);
    int root1 = union_find_get_root(node_idx1, parent_nodes);
    int root2 = union_find_get_root(node_idx2, parent_nodes);
    if (root1 != root2) {
      if (images_in_track[root1].size() < images_in_track[root2].size()) {
        parent_nodes[root1] = root2;
        images_in_track[root2].insert(images_in_track[root1].begin(), images_in_track[root1].end());
        images_in_track[root1].clear();
      } else {
        parent_nodes[root2] = root1;
        images_in_track[root1].insert(images_in_track[root2].begin(), images_in_track[root2].end());
        images_in_track[root2].clear();
      }
    }
  }
  std::vector<int> track_labels(n_nodes, -1);
  int n_tracks = 0;
  for (size_t i = 0; i < n_nodes; ++i) {
    if (parent_nodes[i] == -1) continue;
    int p = parent_nodes[i];
    if (parent_nodes[p] == -1 && track_labels[p] == -1) track_labels[p] = n_tracks++;
  }
  for (size_t i = 0; i < n_nodes; ++i) {
    if (parent_nodes[i] == -1) continue;
    track_labels[i] = track_labels[union_find_get_root(int(i), parent_nodes)];
  }
  return track_labels;
}

// Track labels from the parent array as the reference derives them (merging/merging.cc:84-101; the same
// block closes the exhaustive and avg variants, :222-245 and :345-368).
static std::vector<int> labels_from_parents(std::vector<int> &parent_nodes) {
  const size_t n_nodes = parent_nodes.size();
  std::vector<int> track_labels(n_nodes, -1);
  int n_tracks = 0;
  for (size_t i = 0; i < n_nodes; ++i) {
    if (parent_nodes[i] == -1) continue;
    int p = parent_nodes[i];
    if (parent_nodes[p] == -1 && track_labels[p] == -1) track_labels[p] = n_tracks++;
  }
  for (size_t i = 0; i < n_nodes; ++i) {
    if (parent_nodes[i] == -1) continue;
    track_labels[i] = track_labels[union_find_get_root(int(i), parent_nodes)];
  }
  return track_labels;
}

// merging/merging.cc:105-245: a union is accepted only if every pair of lines of the two unions that
// overlap (one-way overlap of it1 on it2 > 0) passes check_connection in avgtest mode.
static std::vector<int> ComputeLineTrackLabelsExhaustive(const std::vector<int> &node_img,
                                                         std::vector<std::tuple<double, int, int>> edges,
                                                         const std::vector<Line3d> &line3d_list_nodes,
                                                         Linker3d linker3d) {
  linker3d.config.set_to_avgtest_merging();
  const size_t n_nodes = node_img.size();
  std::sort(edges.begin(), edges.end());
  std::reverse(edges.begin(), edges.end());
  std::vector<int> parent_nodes(n_nodes, -1);
  std::vector<std::set<int>> images_in_track(n_nodes);
  std::vector<std::vector<Line3d>> lines_in_track(n_nodes);
  for (size_t i = 0; i < n_nodes; ++i) {
    images_in_track[i].insert(node_img[i]);
    lines_in_track[i].push_back(line3d_list_nodes[i]);
  }
  for (const auto &e : edges) {
    int root1 = union_find_get_root(std::get<1>(e), parent_nodes);
    int root2 = union_find_get_root(std::get<2>(e), parent_nodes);
    if (root1 == root2) continue;
    bool flag = true;
    for (const Line3d &a : lines_in_track[root1]) {
      for (const Line3d &b : lines_in_track[root2]) {
        if (compute_overlap(a, b) <= 0) continue;
        if (!linker3d.check_connection(a, b)) {
          flag = false;
          break;
        }
      }
      if (!flag) break;
    }
    if (!flag) continue;
    int dst = root1, src = root2;
    if (images_in_track[root1].size() < images_in_track[root2].size()) std::swap(dst, src);
    parent_nodes[src] = dst;
    images_in_track[dst].insert(images_in_track[src].begin(), images_in_track[src].end());
    images_in_track[src].clear();
    lines_in_track[dst].insert(lines_in_track[dst].end(), lines_in_track[src].begin(), lines_in_track[src].end());
    lines_in_track[src].clear();
  }
  return labels_from_parents(parent_nodes);
}

// merging/merging.cc:247-368: a union is accepted if the running AVERAGE lines of the two unions pass
// check_connection in avgtest mode.  The averaged line is a default-constructed Line3d with only start/end
// set (:300-307, :323-330): uncertainty -1, which enters the perpendicular score through min(u1, u2).
static std::vector<int> ComputeLineTrackLabelsAvg(const std::vector<int> &node_img,
                                                  std::vector<std::tuple<double, int, int>> edges,
                                                  const std::vector<Line3d> &line3d_list_nodes, Linker3d linker3d) {
  linker3d.config.set_to_avgtest_merging();
  const size_t n_nodes = node_img.size();
  std::sort(edges.begin(), edges.end());
  std::reverse(edges.begin(), edges.end());
  std::vector<int> parent_nodes(n_nodes, -1);
  std::vector<std::set<int>> images_in_track(n_nodes);
  std::vector<std::pair<Line3d, int>> avgline_in_track(n_nodes);
  for (size_t i = 0; i < n_nodes; ++i) {
    images_in_track[i].insert(node_img[i]);
    avgline_in_track[i] = {line3d_list_nodes[i], 1};
  }
  for (const auto &e : edges) {
    int root1 = union_find_get_root(std::get<1>(e), parent_nodes);
    int root2 = union_find_get_root(std::get<2>(e), parent_nodes);
    if (root1 == root2) continue;
    if (!linker3d.check_connection(avgline_in_track[root1].first, avgline_in_track[root2].first)) continue;
    int dst = root1, src = root2;
    if (images_in_track[root1].size() < images_in_track[root2].size()) std::swap(dst, src);
    parent_nodes[src] = dst;
    images_in_track[dst].insert(images_in_track[src].begin(), images_in_track[src].end());
    images_in_track[src].clear();
    auto d1 = avgline_in_track[dst];
    auto d2 = avgline_in_track[src];
    Line3d newline;
    double n1 = d1.second, n2 = d2.second, nsum = d1.second + d2.second;
    newline.start = (d1.first.start * n1 + d2.first.start * n2) / nsum;
    newline.end = (d1.first.end * n1 + d2.first.end * n2) / nsum;
    avgline_in_track[dst] = {newline, d1.second + d2.second};
  }
  return labels_from_parents(parent_nodes);
}

// Eigen::JacobiSVD<MatrixXd>(rows, ComputeThinV).matrixV().col(0) of an n x 3 matrix (aggregator.cc:76-78,
// base_line_triangulator.cc:229-230) by Eigen 3.4's own procedure, sign included: oracle/eigen_svd_ref.h.
static V3 principal_direction(const std::vector<V3> &rows) {
  const int n = int(rows.size());
  ora_svd::Mat A(n, 3), V;
  for (int r = 0; r < n; ++r) {
    A(r, 0) = rows[size_t(r)].x;
    A(r, 1) = rows[size_t(r)].y;
    A(r, 2) = rows[size_t(r)].z;
  }
  std::vector<double> sv;
  ora_svd::jacobi_svd_thin_v(A, V, sv);
  return V3{V(0, 0), V(1, 0), V(2, 0)};
}


static Line3d aggregate_takebest(const std::vector<Line3d> &lines,
                                 const std::vector<double> &scores) {  // aggregator.cc:8-29
  int n_lines = int(lines.size());
  double best_score = 0.0;
  int best_idx = -1;
  double min_unc = std::numeric_limits<double>::max();
  for (int i = 0; i < n_lines; ++i) {
    if (scores[i] > best_score) {
      best_score = scores[i];
      best_idx = i;
    }
    if (lines[i].uncertainty < min_unc) min_unc = lines[i].uncertainty;
  }
  if (best_idx < 0) best_idx = 0;  // reference indexes lines[-1] (UB); unreachable on the path
  Line3d best = lines[best_idx];
  best.uncertainty = min_unc;
  return best;
}

static Line3d aggregate_line3d_list(const std::vector<Line3d> &lines,
                                    const std::vector<double> &scores,
                                    int num_outliers) {  // aggregator.cc:53-101
  int n_lines = int(lines.size());
  if (n_lines < 4) return aggregate_takebest(lines, scores);
  V3 center{0, 0, 0};
  for (int i = 0; i < n_lines; ++i) {
    center = center + lines[i].start;
    center = center + lines[i].end;
  }
  center = center / double(2 * n_lines);
  std::vector<V3> endpoints(size_t(n_lines) * 2);
  for (int i = 0; i < n_lines; ++i) {
    endpoints[2 * i] = lines[i].start - center;
    endpoints[2 * i + 1] = lines[i].end - center;
  }
  V3 direc = principal_direction(endpoints);
  direc = direc / norm(direc);
  std::vector<double> projections;
  for (int i = 0; i < n_lines; ++i) {
    projections.push_back(dot(lines[i].start - center, direc));
    projections.push_back(dot(lines[i].end - center, direc));
  }
  std::sort(projections.begin(), projections.end());
  double min_unc = std::numeric_limits<double>::max();
  for (int i = 0; i < n_lines; ++i)
    if (lines[i].uncertainty < min_unc) min_unc = lines[i].uncertainty;
  Line3d fl;
  fl.start = center + direc * projections[num_outliers];
  fl.end = center + direc * projections[n_lines * 2 - 1 - num_outliers];
  fl.uncertainty = min_unc;
  return fl;
}

struct LineTrack {  // base/linetrack.h:21-50 (fields the path fills)
  Line3d line;
  std::vector<int> image_id_list, line_id_list, node_id_list;
  std::vector<Line2d> line2d_list;
  std::vector<Line3d> line3d_list;
  std::vector<double> score_list;
  bool active = true;
  size_t count_lines() const { return line2d_list.size(); }
};

// ---------------------------------------------------------------------------------------------
// merging/merging_utils.cc:27-155, merging/merging.cc:513-644 -- the steps that follow
// ComputeLineTracks in runners/line_triangulation.py:171-200
// ---------------------------------------------------------------------------------------------
template <class L>
static double dist_endpoints_perpendicular_oneway(const L &l1, const L &l2) {  // line_dists.h:113-120
  auto d = dists_endpoints_perpendicular_oneway(l1, l2);
  return std::max(d.first, d.second);
}

static void CheckReprojection(std::vector<bool> &results, const LineTrack &tr,
                              const std::map<int, CameraView> &views, double th_angular2d,
                              double th_perp2d) {  // merging_utils.cc:27-49
  results.clear();
  for (size_t i = 0; i < tr.count_lines(); ++i) {
    const Line2d &line2d = tr.line2d_list[i];
    Line2d proj = tr.line.projection(views.at(tr.image_id_list[i]));
    double angle = compute_angle(line2d, proj);
    if (angle > th_angular2d) {
      results.push_back(false);
      continue;
    }
    double d = dist_endpoints_perpendicular_oneway(line2d, proj);
    if (d > th_perp2d) {
      results.push_back(false);
      continue;
    }
    results.push_back(true);
  }
}

static std::vector<LineTrack> FilterSupportingLines(const std::vector<LineTrack> &tracks,
                                                    const std::map<int, CameraView> &views,
                                                    double th_angular2d, double th_perp2d,
                                                    int num_outliers) {  // merging_utils.cc:51-83
  std::vector<LineTrack> out;
  for (const auto &tr : tracks) {
    std::vector<bool> res;
    CheckReprojection(res, tr, views, th_angular2d, th_perp2d);
    LineTrack nt;
    for (size_t k = 0; k < tr.count_lines(); ++k) {
      if (!res[k]) continue;
      nt.node_id_list.push_back(tr.node_id_list[k]);
      nt.image_id_list.push_back(tr.image_id_list[k]);
      nt.line_id_list.push_back(tr.line_id_list[k]);
      nt.line2d_list.push_back(tr.line2d_list[k]);
      nt.line3d_list.push_back(tr.line3d_list[k]);
      nt.score_list.push_back(tr.score_list[k]);
    }
    if (nt.count_lines() == 0) continue;
    nt.line = aggregate_line3d_list(nt.line3d_list, nt.score_list, num_outliers);
    out.push_back(nt);
  }
  return out;
}

static std::vector<LineTrack> FilterTracksBySensitivity(const std::vector<LineTrack> &tracks,
                                                        const std::map<int, CameraView> &views,
                                                        double th_angular3d,
                                                        int min_support_ns) {  // merging_utils.cc:85-128
  std::vector<LineTrack> out;
  for (const auto &tr : tracks) {
    std::set<int> support_images;
    for (size_t i = 0; i < tr.count_lines(); ++i) {
      double sens = tr.line.sensitivity(views.at(tr.image_id_list[i]));
      if (!(sens > th_angular3d)) support_images.insert(tr.image_id_list[i]);
    }
    if (int(support_images.size()) >= min_support_ns) out.push_back(tr);
  }
  return out;
}

static std::vector<LineTrack> FilterTracksByOverlap(const std::vector<LineTrack> &tracks,
                                                    const std::map<int, CameraView> &views,
                                                    double th_overlap,
                                                    int min_support_ns) {  // merging_utils.cc:130-155
  std::vector<LineTrack> out;
  for (const auto &tr : tracks) {
    std::set<int> support_images;
    for (size_t i = 0; i < tr.count_lines(); ++i) {
      Line2d proj = tr.line.projection(views.at(tr.image_id_list[i]));
      double overlap = compute_overlap(proj, tr.line2d_list[i]);
      if (overlap >= th_overlap) support_images.insert(tr.image_id_list[i]);
    }
    if (int(support_images.size()) >= min_support_ns) out.push_back(tr);
  }
  return out;
}

static std::vector<LineTrack> RemergeLineTracks(const std::vector<LineTrack> &tracks, Linker3d linker3d,
                                                int num_outliers) {  // merging/merging.cc:513-644
  linker3d.config.set_to_spatial_merging();
  const size_t n_tracks = tracks.size();
  std::set<std::pair<size_t, size_t>> edges;
  std::vector<std::set<std::pair<size_t, size_t>>> edges_per_track(n_tracks);
  std::vector<int> active_ids;
  for (size_t i = 0; i < n_tracks; ++i)
    if (tracks[i].active) active_ids.push_back(int(i));
  const size_t n_active = active_ids.size();
#pragma omp parallel for
  for (size_t k = 0; k < n_active; ++k) {
    size_t i = size_t(active_ids[k]);
    const Line3d &l1 = tracks[i].line;
    for (size_t j = 0; j < n_tracks; ++j) {
      if (i == j) continue;
      if (n_active == n_tracks) {
        if (i < j && (i + j) % 2 == 0) continue;
        if (i > j && (i + j) % 2 == 1) continue;
      }
      if (!linker3d.check_connection(l1, tracks[j].line)) continue;
      if (i < j) edges_per_track[i].insert({i, j});
      else edges_per_track[i].insert({j, i});
    }
  }
  for (size_t i = 0; i < n_tracks; ++i) edges.insert(edges_per_track[i].begin(), edges_per_track[i].end());
  std::vector<int> parent(n_tracks, -1);
  std::vector<size_t> group_size(n_tracks, 1);  // |tracks_in_group|
  for (const auto &e : edges) {
    size_t r1 = size_t(union_find_get_root(int(e.first), parent));
    size_t r2 = size_t(union_find_get_root(int(e.second), parent));
    if (r1 == r2) continue;
    if (group_size[r1] < group_size[r2]) {
      parent[r1] = int(r2);
      group_size[r2] += group_size[r1];
      group_size[r1] = 0;
    } else {
      parent[r2] = int(r1);
      group_size[r1] += group_size[r2];
      group_size[r2] = 0;
    }
  }
  std::vector<long> labels(n_tracks, -1);
  size_t n_groups = 0;
  for (size_t t = 0; t < n_tracks; ++t)
    if (parent[t] == -1) labels[t] = long(n_groups++);
  for (size_t t = 0; t < n_tracks; ++t)
    if (labels[t] == -1) labels[t] = labels[size_t(union_find_get_root(int(t), parent))];
  std::vector<LineTrack> out(n_groups);
  std::vector<int> counter(n_groups, 0);
  for (size_t t = 0; t < n_tracks; ++t) {
    const LineTrack &tr = tracks[t];
    LineTrack &g = out[size_t(labels[t])];
    counter[size_t(labels[t])]++;
    g.node_id_list.insert(g.node_id_list.end(), tr.node_id_list.begin(), tr.node_id_list.end());
    g.image_id_list.insert(g.image_id_list.end(), tr.image_id_list.begin(), tr.image_id_list.end());
    g.line_id_list.insert(g.line_id_list.end(), tr.line_id_list.begin(), tr.line_id_list.end());
    g.line2d_list.insert(g.line2d_list.end(), tr.line2d_list.begin(), tr.line2d_list.end());
    g.line3d_list.insert(g.line3d_list.end(), tr.line3d_list.begin(), tr.line3d_list.end());
    g.score_list.insert(g.score_list.end(), tr.score_list.begin(), tr.score_list.end());
  }
  for (size_t gidx = 0; gidx < n_groups; ++gidx) {
    out[gidx].line = aggregate_line3d_list(out[gidx].line3d_list, out[gidx].score_list, num_outliers);
    if (counter[gidx] == 1) out[gidx].active = false;
  }
  return out;
}

// ---------------------------------------------------------------------------------------------
// triangulation/base_line_triangulator.{h,cc} + global_line_triangulator.{h,cc}
// ---------------------------------------------------------------------------------------------
struct TriTuple {  // base_line_triangulator.h:17-18
  Line3d line;
  double score = 0.0;  // value-initialised tuple element
  int ng_img = 0, ng_line = 0;
};

struct Config {
  ora_config c;
};

struct Triangulator {
  ora_config cfg;
  bool faithful = true;
  std::string err;

  // ImageCollection slice
  std::map<int, CameraView> views;  // img_id -> view
  std::vector<int> img_ids;         // ascending

  std::map<int, std::vector<Line2d>> all_lines_2d_;
  std::map<int, std::vector<int>> neighbors_;
  std::map<int, std::vector<std::vector<std::pair<int, int>>>> edges_;
  std::map<int, std::vector<std::vector<TriTuple>>> tris_;
  std::map<int, std::vector<std::vector<TriTuple>>> tris_debug_;  // kept copy when debug_mode
  std::map<int, std::vector<int>> n_tris_;                        // stats (always)
  std::map<int, std::vector<std::vector<std::pair<int, int>>>> valid_edges_;  // (nb index, line)
  std::map<int, std::vector<TriTuple>> tris_best_;
  std::map<int, std::vector<uint8_t>> has_best_;
  std::map<int, std::vector<bool>> already_scored_;
  std::map<int, std::vector<bool>> valid_flags_;
  bool ranges_flag_ = false;
  std::pair<V3, V3> ranges_;
  Linker2d linker2d;
  Linker3d linker3d;
  std::vector<LineTrack> tracks_;
  int64_t stat_connections = 0, stat_candidates = 0, stat_pairs = 0, stat_graph_nodes = 0,
          stat_graph_edges = 0;
  double t_gen = 0, t_score = 0, t_tail = 0;

  // imagecols_->camview(img_id): by value (image_collection.cc:366-371)
  CameraView camview(int img_id) const { return views.at(img_id); }
  const CameraView &camview_ref(int img_id) const { return views.at(img_id); }

  size_t CountLines(int img_id) const { return all_lines_2d_.at(img_id).size(); }

  // vplib::VPResult (vplib/vpbase.h:18-47): labels[line] = VP index or -1; InitVPResults,
  // base_line_triangulator.h:47-49
  struct VPResult {
    std::vector<int> labels;
    std::vector<V3> vps;
    bool HasVP(int line_id) const { return labels.at(line_id) >= 0; }
    V3 GetVP(int line_id) const { return vps.at(labels.at(line_id)); }
  };
  std::map<int, VPResult> vpresults_;

  // structures::PL_Bipartite2d (structures/pl_bipartite_base.h:79-93): points (id -> xy, point3D_id) and
  // per line the ids of its neighbouring points in ascending id order (std::set); SetBipartites2d /
  // SetSfMPoints, base_line_triangulator.h:71-77
  struct Point2d {
    V2 p;
    int point3D_id;
  };
  struct Bipartite2d {
    std::map<int, Point2d> points;
    std::map<int, std::set<int>> nl2p;
    std::vector<int> neighbor_points(int line_id) const {
      auto it = nl2p.find(line_id);
      if (it == nl2p.end()) return {};
      return std::vector<int>(it->second.begin(), it->second.end());
    }
    Point2d point(int point_id) const { return points.at(point_id); }
  };
  bool use_pointsfm_ = false;
  std::map<int, Bipartite2d> all_bpt2ds_;
  std::map<int, V3> sfm_points_;

  void Init() {  // base_line_triangulator.cc:45-63 + global_line_triangulator.cc:31-57
    if (cfg.add_halfpix) {  // offsetHalfPixel, base_line_triangulator.cc:33-43
      for (int img_id : img_ids)
        for (auto &line : all_lines_2d_[img_id]) {
          line.start = line.start + V2{0.5, 0.5};
          line.end = line.end + V2{0.5, 0.5};
        }
    }
    for (int img_id : img_ids) {
      size_t n = all_lines_2d_.at(img_id).size();
      neighbors_[img_id] = {};
      edges_[img_id].assign(n, {});
      tris_[img_id].assign(n, {});
      tris_debug_[img_id].assign(n, {});
      n_tris_[img_id].assign(n, 0);
      valid_edges_[img_id].assign(n, {});
      tris_best_[img_id].assign(n, TriTuple());
      has_best_[img_id].assign(n, 0);
      already_scored_[img_id].assign(n, false);
    }
  }

  void triangulateOneNode(int img_id, int line_id);
  void scoreOneNode(int img_id, int line_id, const Linker2d &l2, const Linker3d &l3);
  void ScoringCallback(int img_id);
  void TriangulateImage(int img_id, const std::map<int, std::vector<std::pair<int, int>>> &matches);
  void TriangulateImageExhaustiveMatch(int img_id, const std::vector<int> &neighbors);
  void filterNodeByNumOuterEdges();
  void run_clustering(Graph &graph);
  void build_tracks_from_clusters(Graph &graph);
  void ComputeLineTracks();
};

static double now_s() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

void Triangulator::triangulateOneNode(int img_id, int line_id) {  // base_line_triangulator.cc:161-337
  auto &connections = edges_[img_id][line_id];
  const Line2d &l1 = all_lines_2d_[img_id][line_id];
  if (l1.length() <= cfg.min_length_2d) return;
  const CameraView view1s = camview(img_id);
  const CameraView &view1 = view1s;
  size_t n_conns = connections.size();
  std::vector<std::vector<TriTuple>> results(n_conns);
  stat_connections += int64_t(n_conns);

#pragma omp parallel for
  for (size_t conn_id = 0; conn_id < n_conns; ++conn_id) {
    int ng_img_id = connections[conn_id].first;
    int ng_line_id = connections[conn_id].second;
    const Line2d &l2 = all_lines_2d_[ng_img_id][ng_line_id];
    if (l2.length() <= cfg.min_length_2d) continue;
    // by-value copy per connection in the reference (line 179)
    CameraView view2_copy;
    const CameraView *view2p;
    if (faithful) {
      view2_copy = camview(ng_img_id);
      view2p = &view2_copy;
    } else {
      view2p = &camview_ref(ng_img_id);
    }
    const CameraView &view2 = *view2p;

    auto push = [&](Line3d line) {  // lines 257-264 / 272-279 / 318-324
      if (line.score > 0) {
        double u1 = line.computeUncertainty(view1, cfg.var2d);
        double u2 = line.computeUncertainty(view2, cfg.var2d);
        line.uncertainty = std::min(u1, u2);
        TriTuple t;
        t.line = line;
        t.score = -1.0;
        t.ng_img = ng_img_id;
        t.ng_line = ng_line_id;
        results[conn_id].push_back(t);
      }
    };
    // Step 1.1: many points -> fit a line through the shared 3D points (lines 183-236).  Step 1.2: one
    // known point (lines 238-248), through a restatement of the optimisation problem of the solver.
    if (use_pointsfm_ && (!cfg.disable_many_points_triangulation || !cfg.disable_one_point_triangulation)) {
      std::map<int, Point2d> points1;
      std::set<int> set1;
      std::map<int, std::pair<V2, V2>> points_info;
      for (int point_id : all_bpt2ds_.at(img_id).neighbor_points(line_id)) {
        Point2d p = all_bpt2ds_.at(img_id).point(point_id);
        set1.insert(p.point3D_id);
        points1.insert({p.point3D_id, p});
      }
      for (int point_id : all_bpt2ds_.at(ng_img_id).neighbor_points(ng_line_id)) {
        Point2d p = all_bpt2ds_.at(ng_img_id).point(point_id);
        if (set1.find(p.point3D_id) != set1.end()) {
          V2 p1 = points1.at(p.point3D_id).p;
          points_info.insert({p.point3D_id, {p1, p.p}});
        }
      }
      std::vector<V3> points;
      for (auto it = points_info.begin(); it != points_info.end(); ++it) {
        if (sfm_points_.empty()) {
          auto res = triangulate_point(it->second.first, view1, it->second.second, view2);
          if (res.second) points.push_back(res.first);
        } else {
          points.push_back(sfm_points_.at(it->first));
        }
      }
      if (!cfg.disable_many_points_triangulation && points.size() >= 2) {
        V3 center{0.0, 0.0, 0.0};
        for (size_t i = 0; i < points.size(); ++i) center = center + points[i];
        center = center / double(points.size());
        std::vector<V3> epoints(points.size());
        for (size_t i = 0; i < points.size(); ++i) epoints[i] = points[i] - center;
        // Eigen::JacobiSVD(epoints, ComputeThinV).matrixV().col(0).normalized(): stand-in, see principal_direction
        V3 direc = normalized(principal_direction(epoints));
        InfiniteLine3d inf_line(center, direc);
        push(triangulate_line_with_infinite_line(l1, view1, inf_line));
      }
      // Step 1.2: one point triangulation (lines 238-248)
      if (!cfg.disable_one_point_triangulation && !points.empty()) {
        for (const V3 &pt : points) push(triangulate_line_with_one_point(l1, view1, l2, view2, pt));
      }
    }

    // Step 2: triangulation with VPs (lines 250-281); note that BOTH directions are mapped with view1
    if (cfg.use_vp && !cfg.disable_vp_triangulation) {
      if (vpresults_.at(img_id).HasVP(line_id)) {
        V3 direc = getDirectionFromVP(vpresults_.at(img_id).GetVP(line_id), view1);
        push(triangulate_line_with_direction(l1, view1, l2, view2, direc));
      }
      if (vpresults_.at(ng_img_id).HasVP(ng_line_id)) {
        V3 direc = getDirectionFromVP(vpresults_.at(ng_img_id).GetVP(ng_line_id), view1);
        push(triangulate_line_with_direction(l1, view1, l2, view2, direc));
      }
    }

    // Step 3: algebraic line triangulation (lines 291-325)
    if (!cfg.disable_algebraic_triangulation) {
      V3 n2 = getNormalDirection(l2, view2);
      V3 ray1_start = view1.ray_direction(l1.start);
      double angle_start = 90 - std::acos(std::abs(dot(n2, ray1_start))) * 180.0 / M_PI;
      if (angle_start < cfg.line_tri_angle_threshold) continue;
      V3 ray1_end = view1.ray_direction(l1.end);
      double angle_end = 90 - std::acos(std::abs(dot(n2, ray1_end))) * 180.0 / M_PI;
      if (angle_end < cfg.line_tri_angle_threshold) continue;

      double IoU = compute_epipolar_IoU(l1, view1, l2, view2);
      if (IoU < cfg.IoU_threshold) continue;

      Line3d line;
      if (!cfg.use_endpoints_triangulation)
        line = triangulate_line(l1, view1, l2, view2);
      else
        line = triangulate_line_by_endpoints(l1, view1, l2, view2);
      if (line.sensitivity(view1) > cfg.sensitivity_threshold &&
          line.sensitivity(view2) > cfg.sensitivity_threshold)
        line.score = -1;
      push(line);
    }
  }
  for (size_t conn_id = 0; conn_id < n_conns; ++conn_id) {
    for (auto &t : results[conn_id]) {
      if (ranges_flag_) {
        if (!test_line_inside_ranges(t.line, ranges_)) continue;
      }
      tris_[img_id][line_id].push_back(t);
    }
  }
}

void Triangulator::TriangulateImage(
    int img_id, const std::map<int, std::vector<std::pair<int, int>>> &matches) {
  // base_line_triangulator.cc:71-109
  double t0 = now_s();
  neighbors_[img_id].clear();
  for (auto it = matches.begin(); it != matches.end(); ++it) {
    int ng_img_id = it->first;
    const auto &match_info = it->second;
    neighbors_[img_id].push_back(ng_img_id);
    for (size_t k = 0; k < match_info.size(); ++k) {
      int line_id = match_info[k].first;
      int ng_line_id = match_info[k].second;
      if (size_t(line_id) >= edges_[img_id].size())  // int -> size_t compare as in the reference
        throw std::runtime_error("IndexError! Out-of-index matches exist between image (img_id = " +
                                 std::to_string(img_id) + ") and neighbor image (img_id = " +
                                 std::to_string(ng_img_id) + ").");
      edges_[img_id][line_id].push_back({ng_img_id, ng_line_id});
    }
    for (size_t line_id = 0; line_id < CountLines(img_id); ++line_id) {
      triangulateOneNode(img_id, int(line_id));
      edges_[img_id][line_id].clear();
    }
  }
  t_gen += now_s() - t0;
  ScoringCallback(img_id);
}

void Triangulator::TriangulateImageExhaustiveMatch(int img_id, const std::vector<int> &neighbors) {
  // base_line_triangulator.cc:111-136
  double t0 = now_s();
  neighbors_[img_id] = neighbors;
  for (size_t nb = 0; nb < neighbors.size(); ++nb) {
    int ng_img_id = neighbors[nb];
    int n_lines_ng = int(all_lines_2d_.at(ng_img_id).size());
    size_t n_lines = CountLines(img_id);
    for (size_t line_id = 0; line_id < n_lines; ++line_id) {
      for (int ng_line_id = 0; ng_line_id < n_lines_ng; ++ng_line_id)
        edges_[img_id][line_id].push_back({ng_img_id, ng_line_id});
      triangulateOneNode(img_id, int(line_id));
      edges_[img_id][line_id].clear();
    }
  }
  t_gen += now_s() - t0;
  ScoringCallback(img_id);
}

void Triangulator::ScoringCallback(int img_id) {  // global_line_triangulator.cc:59-69
  double t0 = now_s();
  Linker2d l2 = linker2d;
  Linker3d l3 = linker3d;
  l3.config.set_to_shared_parent_scoring();
  for (size_t line_id = 0; line_id < CountLines(img_id); ++line_id)
    scoreOneNode(img_id, int(line_id), l2, l3);
  t_score += now_s() - t0;
}

void Triangulator::scoreOneNode(int img_id, int line_id, const Linker2d &lk2,
                                const Linker3d &lk3) {  // global_line_triangulator.cc:71-161
  if (already_scored_[img_id][line_id]) return;
  auto &tris = tris_[img_id][line_id];
  size_t n_tris = tris.size();
  n_tris_[img_id][line_id] = int(n_tris);
  stat_candidates += int64_t(n_tris);
  stat_pairs += int64_t(n_tris) * int64_t(n_tris);

  std::vector<double> scores(n_tris, 0);
#pragma omp parallel for
  for (size_t i = 0; i < n_tris; ++i) {
    std::map<int, std::vector<double>> score_table;
    const Line3d &l1 = tris[i].line;
    int src_img_id = tris[i].ng_img;  // shadows img_id in the reference (line 85)
    CameraView view1_copy;
    if (faithful) view1_copy = camview(src_img_id);  // line 87 (unused afterwards)
    for (size_t j = 0; j < n_tris; ++j) {
      if (i == j) continue;
      const Line3d &l2 = tris[j].line;
      int ng_img_id = tris[j].ng_img;
      int ng_line_id = tris[j].ng_line;
      if (ng_img_id == src_img_id) continue;
      CameraView view2_copy;
      const CameraView *view2p;
      if (faithful) {
        view2_copy = camview(ng_img_id);  // line 96: copied before the 3D test
        view2p = &view2_copy;
      } else {
        view2p = &camview_ref(ng_img_id);
      }
      double score3d = lk3.compute_score(l1, l2);
      if (score3d == 0) continue;
      double score2d =
          lk2.compute_score(l1.projection(*view2p), all_lines_2d_[ng_img_id][ng_line_id]);
      if (score2d == 0) continue;
      double score = std::min(score3d, score2d);
      score_table[ng_img_id].push_back(score);
    }
    for (auto it = score_table.begin(); it != score_table.end(); ++it)
      scores[i] += *std::max_element(it->second.begin(), it->second.end());
  }
  for (size_t i = 0; i < n_tris; ++i) tris[i].score = scores[i];

  // valid tris and connections (lines 118-142)
  std::map<int, int> reverse_mapper;
  int n_neighbors = int(neighbors_[img_id].size());
  for (int i = 0; i < n_neighbors; ++i) reverse_mapper.insert({neighbors_[img_id][i], i});
  std::vector<std::pair<double, int>> scores_to_sort;
  for (size_t tri_id = 0; tri_id < tris.size(); ++tri_id)
    scores_to_sort.push_back({tris[tri_id].score, int(tri_id)});
  std::sort(scores_to_sort.begin(), scores_to_sort.end(), std::greater<std::pair<double, int>>());
  int n_valid_conns = std::min(int(scores_to_sort.size()), cfg.max_valid_conns);
  for (int i = 0; i < n_valid_conns; ++i) {
    int tri_id = scores_to_sort[i].second;
    auto &tri = tris[tri_id];
    if (tri.score < cfg.fullscore_th) continue;
    valid_edges_[img_id][line_id].push_back({reverse_mapper.at(tri.ng_img), tri.ng_line});
  }

  // best tri (lines 144-153)
  double max_score = -1;
  for (size_t tri_id = 0; tri_id < n_tris; ++tri_id) {
    if (tris[tri_id].score > max_score) {
      tris_best_[img_id][line_id] = tris[tri_id];
      has_best_[img_id][line_id] = 1;
      max_score = tris[tri_id].score;
    }
  }
  if (cfg.debug_mode) tris_debug_[img_id][line_id] = tris;
  tris.clear();
  already_scored_[img_id][line_id] = true;
}

void Triangulator::filterNodeByNumOuterEdges() {  // global_line_triangulator.cc:168-232
  valid_flags_.clear();
  for (int img_id : img_ids) valid_flags_[img_id].assign(CountLines(img_id), true);
  std::map<int, std::vector<std::vector<std::pair<int, int>>>> parent_neighbors;
  std::map<int, std::vector<int>> counters;
  for (int img_id : img_ids) {
    size_t n = CountLines(img_id);
    parent_neighbors[img_id].assign(n, {});
    counters[img_id].assign(n, 0);
    for (size_t l = 0; l < n; ++l) counters[img_id][l] = int(valid_edges_.at(img_id)[l].size());
  }
  for (int img_id : img_ids) {
    for (size_t l = 0; l < CountLines(img_id); ++l) {
      for (auto &nd : valid_edges_.at(img_id)[l]) {
        int ng_img_id = neighbors_[img_id][nd.first];
        parent_neighbors[ng_img_id][nd.second].push_back({img_id, int(l)});
      }
      if (counters[img_id][l] < cfg.min_num_outer_edges) valid_flags_[img_id][l] = false;
    }
  }
  std::queue<std::pair<int, int>> q;
  for (int img_id : img_ids)
    for (size_t l = 0; l < CountLines(img_id); ++l)
      if (!valid_flags_[img_id][l]) q.push({img_id, int(l)});
  while (!q.empty()) {
    auto node = q.front();
    q.pop();
    for (auto &p : parent_neighbors[node.first][node.second]) {
      if (!valid_flags_[p.first][p.second]) continue;
      counters[p.first][p.second]--;
      if (counters[p.first][p.second] < cfg.min_num_outer_edges) {
        valid_flags_[p.first][p.second] = false;
        q.push(p);
      }
    }
  }
}

void Triangulator::run_clustering(Graph &graph) {  // global_line_triangulator.cc:234-291
  Linker3d lk3 = linker3d;
  lk3.config.set_to_spatial_merging();
  filterNodeByNumOuterEdges();
  std::set<std::pair<std::pair<int, int>, std::pair<int, int>>> edges;
  for (int img_id : img_ids) {
    for (size_t line_id = 0; line_id < CountLines(img_id); ++line_id) {
      for (auto &nd : valid_edges_[img_id][line_id]) {
        std::pair<int, int> node1{img_id, int(line_id)};
        if (!valid_flags_[node1.first][node1.second]) continue;
        std::pair<int, int> node2{neighbors_[img_id][nd.first], nd.second};
        if (!valid_flags_[node2.first][node2.second]) continue;
        if (node1.first > node2.first ||
            (node1.first == node2.first && node1.second > node2.second))
          std::swap(node1, node2);
        edges.insert({node1, node2});
      }
    }
  }
  for (auto it = edges.begin(); it != edges.end(); ++it) {
    int img_id1 = it->first.first, line_id1 = it->first.second;
    int img_id2 = it->second.first, line_id2 = it->second.second;
    if (faithful) {  // lines 269-270 copy two views; 277-281 compute two discarded 2D scores
      CameraView v1 = camview(img_id1), v2 = camview(img_id2);
      const Line3d &a = tris_best_[img_id1][line_id1].line;
      const Line3d &b = tris_best_[img_id2][line_id2].line;
      volatile double sink = linker2d.compute_score(a.projection(v2), all_lines_2d_[img_id2][line_id2]) +
                             linker2d.compute_score(b.projection(v1), all_lines_2d_[img_id1][line_id1]);
      (void)sink;
    }
    const Line3d &line1 = tris_best_[img_id1][line_id1].line;
    const Line3d &line2 = tris_best_[img_id2][line_id2].line;
    double score_3d = lk3.compute_score(line1, line2);
    double score = score_3d;  // line 283 overwrites min(score_3d, score_2d)
    if (score == 0) continue;
    int n1 = graph.FindOrCreateNode(img_id1, line_id1);
    int n2 = graph.FindOrCreateNode(img_id2, line_id2);
    graph.AddEdge(n1, n2, score);
  }
}

void Triangulator::build_tracks_from_clusters(Graph &graph) {  // global_line_triangulator.cc:293-351
  Linker3d lk3 = this->linker3d;  // :295-296 (the merging variants switch it to avgtest mode themselves too)
  lk3.config.set_to_avgtest_merging();
  std::vector<int> node_img;
  std::vector<Line3d> lines_nodes;  // :299-304
  for (auto &n : graph.nodes) {
    node_img.push_back(n.first);
    lines_nodes.push_back(tris_best_[n.first][n.second].line);
  }
  std::vector<int> track_labels;
  if (cfg.merging_strategy == 0)
    track_labels = ComputeLineTrackLabelsGreedy(node_img, graph.edges);
  else if (cfg.merging_strategy == 1)
    track_labels = ComputeLineTrackLabelsExhaustive(node_img, graph.edges, lines_nodes, lk3);
  else if (cfg.merging_strategy == 2)
    track_labels = ComputeLineTrackLabelsAvg(node_img, graph.edges, lines_nodes, lk3);
  else
    throw std::runtime_error("Error!The given merging strategy is not implemented");
  if (track_labels.empty()) return;
  int n_tracks = *std::max_element(track_labels.begin(), track_labels.end()) + 1;
  tracks_.clear();
  tracks_.resize(n_tracks);
  size_t n_nodes = graph.nodes.size();
  for (size_t node_id = 0; node_id < n_nodes; ++node_id) {
    int img_id = graph.nodes[node_id].first;
    int line_id = graph.nodes[node_id].second;
    int track_id = track_labels[node_id];
    if (track_id == -1) continue;
    auto &tr = tracks_[track_id];
    tr.node_id_list.push_back(int(node_id));
    tr.image_id_list.push_back(img_id);
    tr.line_id_list.push_back(line_id);
    tr.line2d_list.push_back(all_lines_2d_[img_id][line_id]);
    tr.line3d_list.push_back(tris_best_[img_id][line_id].line);
    tr.score_list.push_back(tris_best_[img_id][line_id].score);
  }
  for (auto &tr : tracks_)
    tr.line = aggregate_line3d_list(tr.line3d_list, tr.score_list, cfg.num_outliers_aggregator);
}

void Triangulator::ComputeLineTracks() {  // global_line_triangulator.cc:353-359
  double t0 = now_s();
  Graph g;
  run_clustering(g);
  stat_graph_nodes = int64_t(g.nodes.size());
  stat_graph_edges = int64_t(g.edges.size());
  build_tracks_from_clusters(g);
  t_tail += now_s() - t0;
}

// helpers for the C interface ------------------------------------------------------------------
static CameraView view_from_cam11(const double cam[11]) {
  CameraView v;
  v.cam.params = {cam[0], cam[1], cam[2], cam[3]};
  // CameraPose(qvec, tvec) normalises qvec once at construction (camera.h:95-96)
  double n = std::sqrt((cam[4] * cam[4] + cam[6] * cam[6]) + (cam[5] * cam[5] + cam[7] * cam[7]));
  for (int i = 0; i < 4; ++i) v.pose.q[i] = n > 0 ? cam[4 + i] / n : cam[4 + i];
  v.pose.t = V3{cam[8], cam[9], cam[10]};
  v.name = "none";
  return v;
}
static Line2d seg_to_line(const double s[4]) {
  Line2d l;
  l.start = V2{s[0], s[1]};
  l.end = V2{s[2], s[3]};
  return l;
}
static Line3d line_from10(const double a[10]) {
  Line3d l(V3{a[0], a[1], a[2]}, V3{a[3], a[4], a[5]}, a[9], a[6], a[7], a[8]);
  return l;
}
static void line_to10(const Line3d &l, double a[10]) {
  a[0] = l.start.x; a[1] = l.start.y; a[2] = l.start.z;
  a[3] = l.end.x;   a[4] = l.end.y;   a[5] = l.end.z;
  a[6] = l.depths[0]; a[7] = l.depths[1];
  a[8] = l.uncertainty;
  a[9] = l.score;
}
static void set_linkers(const ora_config &c, Linker2d &l2, Linker3d &l3) {
  l2.config.score_th = c.l2_score_th; l2.config.th_angle = c.l2_th_angle;
  l2.config.th_overlap = c.l2_th_overlap; l2.config.th_smartoverlap = c.l2_th_smartoverlap;
  l2.config.th_smartangle = c.l2_th_smartangle; l2.config.th_perp = c.l2_th_perp;
  l2.config.th_innerseg = c.l2_th_innerseg;
  l2.config.use_angle = c.l2_use_angle; l2.config.use_overlap = c.l2_use_overlap;
  l2.config.use_smartangle = c.l2_use_smartangle; l2.config.use_perp = c.l2_use_perp;
  l2.config.use_innerseg = c.l2_use_innerseg;
  l3.config.score_th = c.l3_score_th; l3.config.th_angle = c.l3_th_angle;
  l3.config.th_overlap = c.l3_th_overlap; l3.config.th_smartoverlap = c.l3_th_smartoverlap;
  l3.config.th_smartangle = c.l3_th_smartangle; l3.config.th_perp = c.l3_th_perp;
  l3.config.th_innerseg = c.l3_th_innerseg; l3.config.th_scaleinv = c.l3_th_scaleinv;
  l3.config.use_angle = c.l3_use_angle; l3.config.use_overlap = c.l3_use_overlap;
  l3.config.use_smartangle = c.l3_use_smartangle; l3.config.use_perp = c.l3_use_perp;
  l3.config.use_innerseg = c.l3_use_innerseg; l3.config.use_scaleinv = c.l3_use_scaleinv;
}

}  // namespace ora

// =============================================================================================
// C interface
// =============================================================================================
struct ora_ctx {
  ora::Triangulator t;
};

#define ORA_TRY(ctx, ...)                 \
  try {                                   \
    __VA_ARGS__;                          \
    return 0;                             \
  } catch (const std::exception &e) {     \
    (ctx)->t.err = e.what();              \
    return -1;                            \
  }

extern "C" {

void ora_config_default(ora_config *c) {
  std::memset(c, 0, sizeof(*c));
  c->min_length_2d = 20.0;
  c->line_tri_angle_threshold = 5.0;
  c->IoU_threshold = 0.1;
  c->sensitivity_threshold = 70.0;
  c->var2d = 2.0;
  c->fullscore_th = 1.0;
  c->max_valid_conns = 1000;
  c->min_num_outer_edges = 1;
  c->merging_strategy = 0;
  c->num_outliers_aggregator = 2;
  ora::Linker2dCfg l2;
  c->l2_score_th = l2.score_th; c->l2_th_angle = l2.th_angle; c->l2_th_overlap = l2.th_overlap;
  c->l2_th_smartoverlap = l2.th_smartoverlap; c->l2_th_smartangle = l2.th_smartangle;
  c->l2_th_perp = l2.th_perp; c->l2_th_innerseg = l2.th_innerseg;
  c->l2_use_angle = l2.use_angle; c->l2_use_overlap = l2.use_overlap;
  c->l2_use_smartangle = l2.use_smartangle; c->l2_use_perp = l2.use_perp;
  c->l2_use_innerseg = l2.use_innerseg;
  ora::Linker3dCfg l3;
  c->l3_score_th = l3.score_th; c->l3_th_angle = l3.th_angle; c->l3_th_overlap = l3.th_overlap;
  c->l3_th_smartoverlap = l3.th_smartoverlap; c->l3_th_smartangle = l3.th_smartangle;
  c->l3_th_perp = l3.th_perp; c->l3_th_innerseg = l3.th_innerseg; c->l3_th_scaleinv = l3.th_scaleinv;
  c->l3_use_angle = l3.use_angle; c->l3_use_overlap = l3.use_overlap;
  c->l3_use_smartangle = l3.use_smartangle; c->l3_use_perp = l3.use_perp;
  c->l3_use_innerseg = l3.use_innerseg; c->l3_use_scaleinv = l3.use_scaleinv;
}

ora_ctx *ora_create(const ora_config *cfg, int faithful) {
  ora_ctx *ctx = new ora_ctx();
  ctx->t.cfg = *cfg;
  ctx->t.faithful = faithful != 0;
  ora::set_linkers(*cfg, ctx->t.linker2d, ctx->t.linker3d);
  return ctx;
}
void ora_destroy(ora_ctx *ctx) { delete ctx; }
void ora_set_num_threads(int n) { omp_set_num_threads(n); }
int ora_get_max_threads(void) { return omp_get_max_threads(); }
const char *ora_last_error(ora_ctx *ctx) { return ctx->t.err.c_str(); }

int ora_set_ranges(ora_ctx *ctx, const double lo[3], const double hi[3]) {
  ctx->t.ranges_flag_ = true;
  ctx->t.ranges_ = {ora::V3{lo[0], lo[1], lo[2]}, ora::V3{hi[0], hi[1], hi[2]}};
  return 0;
}
int ora_unset_ranges(ora_ctx *ctx) {
  ctx->t.ranges_flag_ = false;
  return 0;
}

int ora_init(ora_ctx *ctx, int n_img, const int32_t *img_ids, const double *kvec,
             const double *qvec, const double *tvec, const int64_t *seg_off, const double *segs) {
  ORA_TRY(ctx, {
    auto &t = ctx->t;
    for (int i = 0; i < n_img; ++i) {
      double cam[11];
      for (int k = 0; k < 4; ++k) cam[k] = kvec[4 * i + k];
      for (int k = 0; k < 4; ++k) cam[4 + k] = qvec[4 * i + k];
      for (int k = 0; k < 3; ++k) cam[8 + k] = tvec[3 * i + k];
      ora::CameraView v = ora::view_from_cam11(cam);
      v.name = "image_" + std::to_string(img_ids[i]) + "_some/longer/path/to/file.png";
      t.views[img_ids[i]] = v;
      std::vector<ora::Line2d> lines;
      for (int64_t s = seg_off[i]; s < seg_off[i + 1]; ++s) lines.push_back(ora::seg_to_line(segs + 4 * s));
      t.all_lines_2d_[img_ids[i]] = lines;
    }
    t.img_ids.clear();
    for (auto &kv : t.views) t.img_ids.push_back(kv.first);
    t.Init();
  })
}

int ora_init_vp(ora_ctx *ctx, int n_img, const int32_t *img_ids, const int64_t *label_off, const int32_t *labels,
                const int64_t *vp_off, const double *vps) {  // InitVPResults
  ORA_TRY(ctx, {
    auto &t = ctx->t;
    t.vpresults_.clear();
    for (int i = 0; i < n_img; ++i) {
      ora::Triangulator::VPResult r;
      r.labels.assign(labels + label_off[i], labels + label_off[i + 1]);
      for (int64_t v = vp_off[i]; v < vp_off[i + 1]; ++v) r.vps.push_back(ora::V3{vps[3 * v], vps[3 * v + 1], vps[3 * v + 2]});
      t.vpresults_[img_ids[i]] = r;
    }
  })
}

int ora_set_bipartites(ora_ctx *ctx, int n_img, const int32_t *img_ids, const int64_t *pt_off, const int32_t *pt_ids,
                       const double *pt_xy, const int32_t *pt_p3d, const int64_t *line_off, const int64_t *lp_off,
                       const int32_t *lp_ptids) {  // SetBipartites2d
  ORA_TRY(ctx, {
    auto &t = ctx->t;
    t.all_bpt2ds_.clear();
    for (int i = 0; i < n_img; ++i) {
      ora::Triangulator::Bipartite2d b;
      for (int64_t k = pt_off[i]; k < pt_off[i + 1]; ++k)
        b.points[pt_ids[k]] = ora::Triangulator::Point2d{ora::V2{pt_xy[2 * k], pt_xy[2 * k + 1]}, pt_p3d[k]};
      for (int64_t l = line_off[i]; l < line_off[i + 1]; ++l)
        for (int64_t e = lp_off[l]; e < lp_off[l + 1]; ++e) b.nl2p[int(l - line_off[i])].insert(lp_ptids[e]);
      t.all_bpt2ds_[img_ids[i]] = b;
    }
    t.use_pointsfm_ = true;
  })
}

int ora_set_sfm_points(ora_ctx *ctx, int64_t n, const int32_t *ids, const double *xyz) {  // SetSfMPoints
  ORA_TRY(ctx, {
    auto &t = ctx->t;
    t.sfm_points_.clear();
    for (int64_t k = 0; k < n; ++k) t.sfm_points_[ids[k]] = ora::V3{xyz[3 * k], xyz[3 * k + 1], xyz[3 * k + 2]};
  })
}

int ora_triangulate_image(ora_ctx *ctx, int img_id, int n_nb, const int32_t *nb_ids,
                          const int64_t *m_off, const int32_t *m_pairs) {
  ORA_TRY(ctx, {
    std::map<int, std::vector<std::pair<int, int>>> matches;
    for (int k = 0; k < n_nb; ++k) {
      auto &v = matches[nb_ids[k]];
      for (int64_t r = m_off[k]; r < m_off[k + 1]; ++r) v.push_back({m_pairs[2 * r], m_pairs[2 * r + 1]});
    }
    ctx->t.TriangulateImage(img_id, matches);
  })
}

int ora_triangulate_image_exhaustive(ora_ctx *ctx, int img_id, int n_nb, const int32_t *nb_ids) {
  ORA_TRY(ctx, {
    std::vector<int> nb(nb_ids, nb_ids + n_nb);
    ctx->t.TriangulateImageExhaustiveMatch(img_id, nb);
  })
}

int ora_compute_tracks(ora_ctx *ctx) { ORA_TRY(ctx, { ctx->t.ComputeLineTracks(); }) }

int64_t ora_num_nodes(ora_ctx *ctx) {
  int64_t n = 0;
  for (int id : ctx->t.img_ids) n += int64_t(ctx->t.CountLines(id));
  return n;
}

int ora_get_num_tris(ora_ctx *ctx, int32_t *out) {
  int64_t g = 0;
  for (int id : ctx->t.img_ids)
    for (int v : ctx->t.n_tris_.at(id)) out[g++] = v;
  return 0;
}

int ora_get_best(ora_ctx *ctx, double *out_line10, double *out_score, int32_t *out_src2,
                 uint8_t *out_has_best) {
  int64_t g = 0;
  for (int id : ctx->t.img_ids) {
    auto &best = ctx->t.tris_best_.at(id);
    auto &hb = ctx->t.has_best_.at(id);
    for (size_t l = 0; l < best.size(); ++l, ++g) {
      out_has_best[g] = hb[l];
      if (hb[l]) {
        ora::line_to10(best[l].line, out_line10 + 10 * g);
        out_score[g] = best[l].score;
        out_src2[2 * g] = best[l].ng_img;
        out_src2[2 * g + 1] = best[l].ng_line;
      } else {
        for (int k = 0; k < 10; ++k) out_line10[10 * g + k] = 0.0;
        out_score[g] = 0.0;
        out_src2[2 * g] = out_src2[2 * g + 1] = 0;
      }
    }
  }
  return 0;
}

int64_t ora_num_valid_edges(ora_ctx *ctx) {
  int64_t n = 0;
  for (int id : ctx->t.img_ids)
    for (auto &v : ctx->t.valid_edges_.at(id)) n += int64_t(v.size());
  return n;
}

int ora_get_valid_edges(ora_ctx *ctx, int64_t *out_off, int32_t *out_edges2) {
  int64_t g = 0, e = 0;
  out_off[0] = 0;
  for (int id : ctx->t.img_ids)
    for (auto &v : ctx->t.valid_edges_.at(id)) {
      for (auto &p : v) {
        out_edges2[2 * e] = p.first;
        out_edges2[2 * e + 1] = p.second;
        ++e;
      }
      out_off[++g] = e;
    }
  return 0;
}

int64_t ora_num_all_tris(ora_ctx *ctx) {
  int64_t n = 0;
  for (int id : ctx->t.img_ids)
    for (auto &v : ctx->t.tris_debug_.at(id)) n += int64_t(v.size());
  return n;
}

int ora_get_all_tris(ora_ctx *ctx, int64_t *out_off, double *out_line10, double *out_score,
                     int32_t *out_src2) {
  int64_t g = 0, e = 0;
  out_off[0] = 0;
  for (int id : ctx->t.img_ids)
    for (auto &v : ctx->t.tris_debug_.at(id)) {
      for (auto &t : v) {
        ora::line_to10(t.line, out_line10 + 10 * e);
        out_score[e] = t.score;
        out_src2[2 * e] = t.ng_img;
        out_src2[2 * e + 1] = t.ng_line;
        ++e;
      }
      out_off[++g] = e;
    }
  return 0;
}

int64_t ora_num_tracks(ora_ctx *ctx) { return int64_t(ctx->t.tracks_.size()); }
int64_t ora_num_track_members(ora_ctx *ctx) {
  int64_t n = 0;
  for (auto &tr : ctx->t.tracks_) n += int64_t(tr.image_id_list.size());
  return n;
}
int ora_get_tracks(ora_ctx *ctx, double *out_line7, int64_t *out_off, int32_t *out_img_ids,
                   int32_t *out_line_ids, int32_t *out_node_ids, double *out_scores,
                   double *out_line3d10) {
  int64_t e = 0, ti = 0;
  out_off[0] = 0;
  for (auto &tr : ctx->t.tracks_) {
    double *o = out_line7 + 7 * ti;
    o[0] = tr.line.start.x; o[1] = tr.line.start.y; o[2] = tr.line.start.z;
    o[3] = tr.line.end.x;   o[4] = tr.line.end.y;   o[5] = tr.line.end.z;
    o[6] = tr.line.uncertainty;
    for (size_t k = 0; k < tr.image_id_list.size(); ++k, ++e) {
      out_img_ids[e] = tr.image_id_list[k];
      out_line_ids[e] = tr.line_id_list[k];
      out_node_ids[e] = tr.node_id_list[k];
      out_scores[e] = tr.score_list[k];
      const ora::Line3d &l = tr.line3d_list[k];
      double *p = out_line3d10 + 10 * e;  // the supporting Line3d in full (linebase.h:37-60)
      p[0] = l.start.x; p[1] = l.start.y; p[2] = l.start.z;
      p[3] = l.end.x;   p[4] = l.end.y;   p[5] = l.end.z;
      p[6] = l.depths[0]; p[7] = l.depths[1]; p[8] = l.uncertainty; p[9] = l.score;
    }
    out_off[++ti] = e;
  }
  return 0;
}

int ora_get_stats(ora_ctx *ctx, int64_t out[8]) {
  auto &t = ctx->t;
  out[0] = t.stat_connections;
  out[1] = t.stat_candidates;
  out[2] = t.stat_pairs;
  out[3] = ora_num_valid_edges(ctx);
  out[4] = t.stat_graph_nodes;
  out[5] = t.stat_graph_edges;
  out[6] = int64_t(t.tracks_.size());
  out[7] = 0;
  return 0;
}
int ora_get_timers(ora_ctx *ctx, double out[4]) {
  out[0] = ctx->t.t_gen;
  out[1] = ctx->t.t_score;
  out[2] = ctx->t.t_tail;
  out[3] = 0;
  return 0;
}

// ---- track sets: post-triangulation filters and remerge on flat arrays ----
struct ora_trackset {
  std::vector<ora::LineTrack> tracks;
};

ora_trackset *ora_ts_from_ctx(ora_ctx *ctx) {
  auto *ts = new ora_trackset();
  ts->tracks = ctx->t.tracks_;
  return ts;
}
void ora_ts_destroy(ora_trackset *ts) { delete ts; }
int64_t ora_ts_num_tracks(ora_trackset *ts) { return int64_t(ts->tracks.size()); }
int64_t ora_ts_num_members(ora_trackset *ts) {
  int64_t n = 0;
  for (auto &t : ts->tracks) n += int64_t(t.count_lines());
  return n;
}
int ora_ts_get(ora_trackset *ts, double *line7, uint8_t *active, int64_t *off, int32_t *img, int32_t *lid,
               int32_t *nid, double *score, double *line2d4, double *line3d10) {
  int64_t e = 0, ti = 0;
  off[0] = 0;
  for (auto &tr : ts->tracks) {
    double *o = line7 + 7 * ti;
    o[0] = tr.line.start.x; o[1] = tr.line.start.y; o[2] = tr.line.start.z;
    o[3] = tr.line.end.x;   o[4] = tr.line.end.y;   o[5] = tr.line.end.z;
    o[6] = tr.line.uncertainty;
    active[ti] = tr.active ? 1 : 0;
    for (size_t k = 0; k < tr.count_lines(); ++k, ++e) {
      img[e] = tr.image_id_list[k]; lid[e] = tr.line_id_list[k]; nid[e] = tr.node_id_list[k];
      score[e] = tr.score_list[k];
      line2d4[4 * e] = tr.line2d_list[k].start.x; line2d4[4 * e + 1] = tr.line2d_list[k].start.y;
      line2d4[4 * e + 2] = tr.line2d_list[k].end.x; line2d4[4 * e + 3] = tr.line2d_list[k].end.y;
      ora::line_to10(tr.line3d_list[k], line3d10 + 10 * e);
    }
    off[++ti] = e;
  }
  return 0;
}
int ora_ts_filter_by_reprojection(ora_ctx *ctx, ora_trackset *ts, double th_angular2d, double th_perp2d,
                                  int num_outliers) {
  ORA_TRY(ctx, { ts->tracks = ora::FilterSupportingLines(ts->tracks, ctx->t.views, th_angular2d, th_perp2d, num_outliers); })
}
int ora_ts_filter_by_sensitivity(ora_ctx *ctx, ora_trackset *ts, double th_angular3d, int min_supports) {
  ORA_TRY(ctx, { ts->tracks = ora::FilterTracksBySensitivity(ts->tracks, ctx->t.views, th_angular3d, min_supports); })
}
int ora_ts_filter_by_overlap(ora_ctx *ctx, ora_trackset *ts, double th_overlap, int min_supports) {
  ORA_TRY(ctx, { ts->tracks = ora::FilterTracksByOverlap(ts->tracks, ctx->t.views, th_overlap, min_supports); })
}
/* one pass of RemergeLineTracks; the linker is taken from cfg's l3_* fields */
int ora_ts_remerge_once(ora_ctx *ctx, ora_trackset *ts, const ora_config *linker_cfg, int num_outliers) {
  ORA_TRY(ctx, {
    ora::Linker2d l2;
    ora::Linker3d l3;
    ora::set_linkers(*linker_cfg, l2, l3);
    ts->tracks = ora::RemergeLineTracks(ts->tracks, l3, num_outliers);
  })
}

// ---- free functions ----
void ora_get_normal_direction(const double seg[4], const double cam[11], double out[3]) {
  ora::V3 n = ora::getNormalDirection(ora::seg_to_line(seg), ora::view_from_cam11(cam));
  out[0] = n.x; out[1] = n.y; out[2] = n.z;
}
static void m3_out(const ora::M3 &m, double out[9]) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) out[3 * i + j] = m.m[i][j];
}
void ora_compute_essential_matrix(const double cam1[11], const double cam2[11], double out[9]) {
  m3_out(ora::compute_essential_matrix(ora::view_from_cam11(cam1), ora::view_from_cam11(cam2)), out);
}
void ora_compute_fundamental_matrix(const double cam1[11], const double cam2[11], double out[9]) {
  m3_out(ora::compute_fundamental_matrix(ora::view_from_cam11(cam1), ora::view_from_cam11(cam2)), out);
}
double ora_compute_epipolar_IoU(const double seg1[4], const double cam1[11], const double seg2[4],
                                const double cam2[11]) {
  return ora::compute_epipolar_IoU(ora::seg_to_line(seg1), ora::view_from_cam11(cam1),
                                   ora::seg_to_line(seg2), ora::view_from_cam11(cam2));
}
int ora_triangulate_point(const double p1[2], const double cam1[11], const double p2[2],
                          const double cam2[11], double out[3]) {
  auto r = ora::triangulate_point(ora::V2{p1[0], p1[1]}, ora::view_from_cam11(cam1),
                                  ora::V2{p2[0], p2[1]}, ora::view_from_cam11(cam2));
  out[0] = r.first.x; out[1] = r.first.y; out[2] = r.first.z;
  return r.second ? 1 : 0;
}
void ora_get_direction_from_vp(const double vp[3], const double cam[11], double out[3]) {
  ora::V3 d = ora::getDirectionFromVP(ora::V3{vp[0], vp[1], vp[2]}, ora::view_from_cam11(cam));
  out[0] = d.x; out[1] = d.y; out[2] = d.z;
}
void ora_triangulate_line_with_direction(const double seg1[4], const double cam1[11], const double seg2[4],
                                         const double cam2[11], const double dir[3], double out10[10]) {
  ora::line_to10(ora::triangulate_line_with_direction(ora::seg_to_line(seg1), ora::view_from_cam11(cam1),
                                                      ora::seg_to_line(seg2), ora::view_from_cam11(cam2),
                                                      ora::V3{dir[0], dir[1], dir[2]}),
                 out10);
}
void ora_set_one_point_solver(int generated) { ora::g_one_point_generated = generated ? 1 : 0; }
int ora_get_one_point_solver(void) { return ora::g_one_point_generated; }
void ora_triangulate_line_with_one_point(const double seg1[4], const double cam1[11], const double seg2[4],
                                         const double cam2[11], const double point[3], double out10[10]) {
  ora::line_to10(ora::triangulate_line_with_one_point(ora::seg_to_line(seg1), ora::view_from_cam11(cam1),
                                                      ora::seg_to_line(seg2), ora::view_from_cam11(cam2),
                                                      ora::V3{point[0], point[1], point[2]}),
                 out10);
}
void ora_triangulate_line(const double seg1[4], const double cam1[11], const double seg2[4],
                          const double cam2[11], double out10[10]) {
  ora::line_to10(ora::triangulate_line(ora::seg_to_line(seg1), ora::view_from_cam11(cam1),
                                       ora::seg_to_line(seg2), ora::view_from_cam11(cam2)),
                 out10);
}
void ora_triangulate_line_by_endpoints(const double seg1[4], const double cam1[11],
                                       const double seg2[4], const double cam2[11],
                                       double out10[10]) {
  ora::line_to10(
      ora::triangulate_line_by_endpoints(ora::seg_to_line(seg1), ora::view_from_cam11(cam1),
                                         ora::seg_to_line(seg2), ora::view_from_cam11(cam2)),
      out10);
}
void ora_cam_project(const double cam[11], const double p[3], double out[2]) {
  ora::V2 r = ora::view_from_cam11(cam).projection(ora::V3{p[0], p[1], p[2]});
  out[0] = r.x; out[1] = r.y;
}
void ora_cam_ray_direction(const double cam[11], const double p2d[2], double out[3]) {
  ora::V3 r = ora::view_from_cam11(cam).ray_direction(ora::V2{p2d[0], p2d[1]});
  out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
double ora_cam_projdepth(const double cam[11], const double p[3]) {
  return ora::view_from_cam11(cam).pose.projdepth(ora::V3{p[0], p[1], p[2]});
}
void ora_cam_R(const double cam[11], double out[9]) { m3_out(ora::view_from_cam11(cam).R(), out); }
void ora_cam_center(const double cam[11], double out[3]) {
  ora::V3 c = ora::view_from_cam11(cam).pose.center();
  out[0] = c.x; out[1] = c.y; out[2] = c.z;
}
double ora_line3d_sensitivity(const double line10[10], const double cam[11]) {
  return ora::line_from10(line10).sensitivity(ora::view_from_cam11(cam));
}
double ora_line3d_uncertainty(const double line10[10], const double cam[11], double var2d) {
  return ora::line_from10(line10).computeUncertainty(ora::view_from_cam11(cam), var2d);
}
double ora_linker2d_score(const ora_config *cfg, const double seg1[4], const double seg2[4]) {
  ora::Linker2d l2;
  ora::Linker3d l3;
  ora::set_linkers(*cfg, l2, l3);
  return l2.compute_score(ora::seg_to_line(seg1), ora::seg_to_line(seg2));
}
double ora_linker3d_score(const ora_config *cfg, int mode3d, const double a[10], const double b[10]) {
  ora::Linker2d l2;
  ora::Linker3d l3;
  ora::set_linkers(*cfg, l2, l3);
  if (mode3d == 1) l3.config.set_to_shared_parent_scoring();
  if (mode3d == 2) l3.config.set_to_spatial_merging();
  if (mode3d == 3) l3.config.set_to_avgtest_merging();
  return l3.compute_score(ora::line_from10(a), ora::line_from10(b));
}
int ora_track_labels_greedy(int n_nodes, const int32_t *node_img, int64_t n_edges,
                            const double *edge_sim, const int32_t *edge_nodes2,
                            int32_t *out_labels) {
  std::vector<int> ni(node_img, node_img + n_nodes);
  std::vector<std::tuple<double, int, int>> edges;
  for (int64_t e = 0; e < n_edges; ++e)
    edges.push_back({edge_sim[e], edge_nodes2[2 * e], edge_nodes2[2 * e + 1]});
  auto labels = ora::ComputeLineTrackLabelsGreedy(ni, edges);
  for (int i = 0; i < n_nodes; ++i) out_labels[i] = labels[i];
  return 0;
}
void ora_aggregate_line3d_list(int n, const double *lines10, const double *scores,
                               int num_outliers, double out7[7]) {
  std::vector<ora::Line3d> lines;
  std::vector<double> sc(scores, scores + n);
  for (int i = 0; i < n; ++i) lines.push_back(ora::line_from10(lines10 + 10 * i));
  ora::Line3d l = ora::aggregate_line3d_list(lines, sc, num_outliers);
  out7[0] = l.start.x; out7[1] = l.start.y; out7[2] = l.start.z;
  out7[3] = l.end.x;   out7[4] = l.end.y;   out7[5] = l.end.z;
  out7[6] = l.uncertainty;
}

}  // extern "C"
