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

// principal axis of a point set: eigenvector of the largest eigenvalue of the 3x3 scatter matrix
// (cyclic Jacobi).  Replaces Eigen::JacobiSVD(...).matrixV().col(0) (merging/aggregator.cc:76-78);
// sign fixed so that the largest-magnitude component is positive.
inline void principal_axis(const std::vector<d3> &pts, double out[3]) {
  double A[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
  for (const d3 &p : pts) {
    double v[3] = {p.x, p.y, p.z};
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) A[i][j] += v[i] * v[j];
  }
  double V[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
  for (int sweep = 0; sweep < 64; ++sweep) {
    double off = std::fabs(A[0][1]) + std::fabs(A[0][2]) + std::fabs(A[1][2]);
    double diag = std::fabs(A[0][0]) + std::fabs(A[1][1]) + std::fabs(A[2][2]);
    if (off <= 1e-18 * diag || off == 0.0) break;
    for (int p = 0; p < 2; ++p)
      for (int q = p + 1; q < 3; ++q) {
        if (A[p][q] == 0.0) continue;
        double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
        double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
        double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < 3; ++k) {
          double akp = A[k][p], akq = A[k][q];
          A[k][p] = c * akp - s * akq;
          A[k][q] = s * akp + c * akq;
        }
        for (int k = 0; k < 3; ++k) {
          double apk = A[p][k], aqk = A[q][k];
          A[p][k] = c * apk - s * aqk;
          A[q][k] = s * apk + c * aqk;
        }
        for (int k = 0; k < 3; ++k) {
          double vkp = V[k][p], vkq = V[k][q];
          V[k][p] = c * vkp - s * vkq;
          V[k][q] = s * vkp + c * vkq;
        }
      }
  }
  int b = 0;
  if (A[1][1] > A[b][b]) b = 1;
  if (A[2][2] > A[b][b]) b = 2;
  double d[3] = {V[0][b], V[1][b], V[2][b]};
  double ax = std::fabs(d[0]), ay = std::fabs(d[1]), az = std::fabs(d[2]);
  double lead = (ax >= ay && ax >= az) ? d[0] : (ay >= az ? d[1] : d[2]);
  double sgn = lead < 0 ? -1.0 : 1.0;
  for (int k = 0; k < 3; ++k) out[k] = sgn * d[k];
}

// Aggregator::aggregate_line3d_list, merging/aggregator.cc:53-101 (+ takebest :8-29).  Two interfaces: a list of
// candidate records (track post-processing) and (table, index list) for the tracks of ComputeLineTracks.
struct AggScratch {
  std::vector<d3> pts;
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

}  // namespace lt
