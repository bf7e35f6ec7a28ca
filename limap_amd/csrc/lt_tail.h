// lt_tail.h -- host-side pieces of the tail shared by ComputeLineTracks and the post-triangulation
// steps: union-find root lookup (base/graph.cc:156-165), principal axis, Aggregator
// (merging/aggregator.cc:8-101).
#pragma once

#include "lt_geom.h"
#include "lt_svd.h"

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

// Principal axis of a centred point set = Eigen::JacobiSVD(points, ComputeThinV).matrixV().col(0) of
// merging/aggregator.cc:76-78, computed by Eigen 3.4's own procedure (lt_svd.h: column-pivoted Householder QR, two-sided
// Jacobi sweeps, descending sort) so that the SIGN of the direction -- it decides which end of the aggregated line is
// `start` -- is the one an Eigen build produces (rounds 1-4: one-sided Jacobi with a sign rule of our own).
inline void principal_axis(const std::vector<d3> &pts, double out[3]) {
  const int n = (int)pts.size();
  // (the threads of the host team aggregate thousands of tracks: storage kept per thread)
  static thread_local lt_svd::Mat A, V;
  static thread_local lt_svd::Scratch sc;
  static thread_local std::vector<double> sv;
  A.reset(n, 3);
  for (int r = 0; r < n; ++r) {
    A(r, 0) = pts[(size_t)r].x;
    A(r, 1) = pts[(size_t)r].y;
    A(r, 2) = pts[(size_t)r].z;
  }
  lt_svd::jacobi_svd_thin_v(A, V, sv, sc);
  out[0] = V(0, 0); out[1] = V(1, 0); out[2] = V(2, 0);
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
  principal_axis(pts, dv);
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
inline void aggregate(const Cand *table, const int *idx, const double *scores, int n, int num_outliers, double out7[7],
                      AggScratch &sc) {
  aggregate_impl([&](int i) -> const Cand & { return table[idx[i]]; }, scores, n, num_outliers, out7, sc);
}

}  // namespace lt
