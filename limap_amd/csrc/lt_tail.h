// lt_tail.h -- host-side pieces of the tail shared by ComputeLineTracks and the post-triangulation
// steps: union-find root lookup (base/graph.cc:156-165), principal axis, Aggregator
// (merging/aggregator.cc:8-101).
#pragma once

#include "lt_geom.h"

#include <omp.h>

#include <algorithm>
#include <cmath>
#include <vector>

namespace lt {

// The host loops here are short (10^3..10^6 cheap items): a bounded team avoids the fork/join cost of a
// 256-thread default on big hosts.
inline int host_threads() {
  int n = omp_get_max_threads();
  return n > 16 ? 16 : (n < 1 ? 1 : n);
}

inline int uf_root(int i, std::vector<int> &parent) {  // base/graph.cc:156-165 (iterative path compression)
  int r = i;
  while (parent[r] != -1) r = parent[r];
  while (parent[i] != -1) {
    int nx = parent[i];
    if (nx != r) parent[i] = r;
    i = nx;
  }
  return r;
}

// Principal axis of a centred point set = first right-singular vector of the n x 3 point matrix, by one-sided
// (Hestenes) Jacobi rotations of the columns: what Eigen::JacobiSVD(points, ComputeThinV).matrixV().col(0) stands
// for in merging/aggregator.cc:76-78.  An SVD leaves the sign of a singular vector open; it decides which end of the
// aggregated line is `start`.  The rule here -- the component of largest magnitude is positive -- is the one of the
// CPU checker this backend is tested against and of the Eigen stand-in the reference sources are compiled with for that
// checker; a real Eigen build may orient a track the other way round.  The rotations work on the
// points themselves, in the same order as there, so the direction agrees to the last bit and no start / end swap is
// left to tolerate in the comparisons.
// `pts` is overwritten.
inline void principal_axis(std::vector<d3> &pts, double out[3]) {
  double V[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
  const int n = (int)pts.size();
  auto col = [&](int r, int c) -> double & { return c == 0 ? pts[(size_t)r].x : (c == 1 ? pts[(size_t)r].y : pts[(size_t)r].z); };
  for (int sweep = 0; sweep < 60; ++sweep) {
    double off = 0;
    for (int p = 0; p < 2; ++p)
      for (int q = p + 1; q < 3; ++q) {
        double alpha = 0, beta = 0, gamma = 0;
        for (int r = 0; r < n; ++r) {
          alpha += col(r, p) * col(r, p);
          beta += col(r, q) * col(r, q);
          gamma += col(r, p) * col(r, q);
        }
        if (gamma == 0.0) continue;
        off = std::max(off, std::fabs(gamma) / std::sqrt(alpha * beta + 1e-300));
        const double zeta = (beta - alpha) / (2.0 * gamma);
        const double t = (zeta >= 0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
        const double c = 1.0 / std::sqrt(1.0 + t * t), s = c * t;
        for (int r = 0; r < n; ++r) {
          const double a = col(r, p), b = col(r, q);
          col(r, p) = c * a - s * b;
          col(r, q) = s * a + c * b;
        }
        for (int r = 0; r < 3; ++r) {
          const double a = V[r][p], b = V[r][q];
          V[r][p] = c * a - s * b;
          V[r][q] = s * a + c * b;
        }
      }
    if (off < 1e-15) break;
  }
  int best = 0;
  double best_n = -1;
  for (int c = 0; c < 3; ++c) {
    double sq = 0;
    for (int r = 0; r < n; ++r) sq += col(r, c) * col(r, c);
    if (sq > best_n) {
      best_n = sq;
      best = c;
    }
  }
  const double d[3] = {V[0][best], V[1][best], V[2][best]};
  const double ax = std::fabs(d[0]), ay = std::fabs(d[1]), az = std::fabs(d[2]);
  const double lead = (ax >= ay && ax >= az) ? d[0] : (ay >= az ? d[1] : d[2]);
  const double sgn = lead < 0 ? -1.0 : 1.0;
  for (int k = 0; k < 3; ++k) out[k] = sgn * d[k];
}

// Aggregator::aggregate_line3d_list, merging/aggregator.cc:53-101 (+ takebest :8-29).  Two interfaces: a list of
// candidate records (track post-processing) and (table, index list) for the tracks of ComputeLineTracks.
struct AggScratch {
  std::vector<d3> pts, rot;
  std::vector<double> proj;
};
template <class GetLine>
inline void aggregate_impl(GetLine line, const double *scores, int n, int num_outliers, double out7[7], AggScratch &sc) {
  double min_unc = kMaxDist;
  for (int i = 0; i < n; ++i)
    if (line(i).unc < min_unc) min_unc = line(i).unc;
  if (n < 4) {
    double best_score = 0.0;
    int best = -1;
    for (int i = 0; i < n; ++i)
      if (scores[i] > best_score) {
        best_score = scores[i];
        best = i;
      }
    if (best < 0) best = 0;
    for (int k = 0; k < 3; ++k) {
      out7[k] = line(best).s[k];
      out7[3 + k] = line(best).e[k];
    }
    out7[6] = min_unc;
    return;
  }
  d3 center = mk3(0, 0, 0);
  for (int i = 0; i < n; ++i) {
    center = add(center, mk3(line(i).s[0], line(i).s[1], line(i).s[2]));
    center = add(center, mk3(line(i).e[0], line(i).e[1], line(i).e[2]));
  }
  double dn = (double)(2 * n);
  center = mk3(center.x / dn, center.y / dn, center.z / dn);
  std::vector<d3> &pts = sc.pts;
  pts.resize(2 * (size_t)n);
  for (int i = 0; i < n; ++i) {
    pts[2 * i] = sub(mk3(line(i).s[0], line(i).s[1], line(i).s[2]), center);
    pts[2 * i + 1] = sub(mk3(line(i).e[0], line(i).e[1], line(i).e[2]), center);
  }
  double dv[3];
  sc.rot = pts;  // the rotations overwrite their matrix; the projections below use the points
  principal_axis(sc.rot, dv);
  d3 direc = mk3(dv[0], dv[1], dv[2]);
  double nn = std::sqrt(sqn(direc));
  direc = mk3(direc.x / nn, direc.y / nn, direc.z / nn);
  std::vector<double> &proj = sc.proj;
  proj.resize(2 * (size_t)n);
  for (int i = 0; i < 2 * n; ++i) proj[i] = dot(pts[i], direc);
  std::sort(proj.begin(), proj.end());
  double a = proj[num_outliers], b = proj[2 * n - 1 - num_outliers];
  out7[0] = center.x + direc.x * a; out7[1] = center.y + direc.y * a; out7[2] = center.z + direc.z * a;
  out7[3] = center.x + direc.x * b; out7[4] = center.y + direc.y * b; out7[5] = center.z + direc.z * b;
  out7[6] = min_unc;
}
inline void aggregate(const std::vector<const Cand *> &lines, const std::vector<double> &scores, int num_outliers,
                      double out7[7]) {
  AggScratch sc;
  aggregate_impl([&](int i) -> const Cand & { return *lines[(size_t)i]; }, scores.data(), (int)lines.size(), num_outliers,
                 out7, sc);
}
inline void aggregate(const Cand *table, const long long *idx, const double *scores, int n, int num_outliers,
                      double out7[7], AggScratch &sc) {
  aggregate_impl([&](int i) -> const Cand & { return table[idx[i]]; }, scores, n, num_outliers, out7, sc);
}

}  // namespace lt
