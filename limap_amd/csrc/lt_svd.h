// lt_svd.h -- the right singular vectors of a small dense matrix by the procedure of Eigen 3.4's
// JacobiSVD<MatrixXd> (default preconditioner), as far as `JacobiSVD(A, ComputeThinV).matrixV()` goes.
//
// Why: merging/aggregator.cc:76-78 takes `svd.matrixV().col(0)` of the (2 n x 3) matrix of centred endpoints as the
// direction of an aggregated track.  An SVD leaves the SIGN of a singular vector open, and that sign decides which
// end of `LineTrack.line` is `start`: to return what a limap built against Eigen returns, the sign has to come out of
// the same sequence of operations, not out of a rule of ours (rounds 1-4 used a one-sided Jacobi iteration with
// "largest component positive").  The procedure, from Eigen 3.4.0's sources as published (file: function):
//   SVD/JacobiSVD.h: JacobiSVD::compute        scale by the largest |entry|; QR preconditioner; sweeps over (p, q),
//                                              p = 1 .. n-1, q = 0 .. p-1, while any |w(p,q)|, |w(q,p)| exceeds
//                                              max(DBL_MIN, 2 eps maxDiagEntry); two-sided rotation of the 2x2 block;
//                                              singular values = |diagonal| (a negative one flips U's column, NOT V's);
//                                              selection sort, descending, first maximum, swapping V's columns
//   SVD/JacobiSVD.h: qr_preconditioner_impl<ColPivHouseholderQRPreconditioner, ...>
//                                              rows > cols: w = R (upper triangle), V = column permutation;
//                                              cols > rows: QR of the adjoint, w = R^T, V = thin Q; square: w = A, V = I
//   QR/ColPivHouseholderQR.h: computeInPlace   pivot = first largest updated column norm, LAPACK-style norm downdate
//   Householder/Householder.h: makeHouseholder, applyHouseholderOnTheLeft
//   misc/RealSvd2x2.h: real_2x2_jacobi_svd;  Jacobi/Jacobi.h: makeJacobi, operator*, apply_rotation_in_the_plane
// What stays an ASSUMPTION (Eigen itself is not on disk here): reductions (norms, the dot products inside the Householder
// application) are summed in index order -- Eigen's vectorised reductions add in a packet-dependent order --, and no
// product-sum is contracted into an FMA.  Both change last bits of the direction, not its sign or which column sorts
// first (outside exact ties).
// (The CPU checker of the test suite carries its own copy of this procedure; it includes no product code.)
#pragma once

#include <cfloat>
#include <cmath>
#include <utility>
#include <vector>

namespace lt_svd {

struct Rot {
  double c, s;
};

// Jacobi.h: JacobiRotation::makeJacobi(x, y, z) for the selfadjoint 2x2 matrix [x y; y z]
inline bool make_jacobi(double x, double y, double z, Rot &r) {
  const double deno = 2.0 * std::fabs(y);
  if (deno < DBL_MIN) {
    r.c = 1.0;
    r.s = 0.0;
    return false;
  }
  const double tau = (x - z) / deno;
  const double w = std::sqrt(tau * tau + 1.0);
  const double t = tau > 0.0 ? 1.0 / (tau + w) : 1.0 / (tau - w);
  const double sign_t = t > 0.0 ? 1.0 : -1.0;
  const double n = 1.0 / std::sqrt(t * t + 1.0);
  r.s = -sign_t * (y / std::fabs(y)) * std::fabs(t) * n;
  r.c = n;
  return true;
}

// RealSvd2x2.h: real_2x2_jacobi_svd on the block [m00 m01; m10 m11]
inline void real_2x2_jacobi_svd(double m00, double m01, double m10, double m11, Rot &j_left, Rot &j_right) {
  Rot rot1;
  const double t = m00 + m11;
  const double d = m10 - m01;
  if (std::fabs(d) < DBL_MIN) {
    rot1.s = 0.0;
    rot1.c = 1.0;
  } else {
    const double u = t / d;
    const double tmp = std::sqrt(1.0 + u * u);
    rot1.s = 1.0 / tmp;
    rot1.c = u / tmp;
  }
  // m.applyOnTheLeft(0, 1, rot1): x = row 0, y = row 1 (apply_rotation_in_the_plane returns early for the identity)
  if (!(rot1.c == 1.0 && rot1.s == 0.0)) {
    const double a0 = m00, a1 = m01, b0 = m10, b1 = m11;
    m00 = rot1.c * a0 + rot1.s * b0;
    m01 = rot1.c * a1 + rot1.s * b1;
    m10 = -rot1.s * a0 + rot1.c * b0;
    m11 = -rot1.s * a1 + rot1.c * b1;
  }
  make_jacobi(m00, m01, m11, j_right);
  // *j_left = rot1 * j_right->transpose();  transpose() = (c, -s);  (a * b).c = a.c b.c - a.s b.s, .s = a.c b.s + a.s b.c
  const double oc = j_right.c, os = -j_right.s;
  j_left.c = rot1.c * oc - rot1.s * os;
  j_left.s = rot1.c * os + rot1.s * oc;
}

// Column-major (rows x cols) work matrix
struct Mat {
  int rows = 0, cols = 0;
  std::vector<double> a;
  Mat() {}
  Mat(int r, int c) : rows(r), cols(c), a((size_t)r * (size_t)c, 0.0) {}
  void reset(int r, int c) {  // zero matrix of the new shape, keeping the storage
    rows = r;
    cols = c;
    a.assign((size_t)r * (size_t)c, 0.0);
  }
  double &operator()(int i, int j) { return a[(size_t)j * (size_t)rows + (size_t)i]; }
  double operator()(int i, int j) const { return a[(size_t)j * (size_t)rows + (size_t)i]; }
};

// apply_rotation_in_the_plane on two strided vectors
inline void rotate(double *x, int incx, double *y, int incy, int n, Rot j) {
  if (j.c == 1.0 && j.s == 0.0) return;
  for (int i = 0; i < n; ++i) {
    const double xi = x[(size_t)i * incx], yi = y[(size_t)i * incy];
    x[(size_t)i * incx] = j.c * xi + j.s * yi;
    y[(size_t)i * incy] = -j.s * xi + j.c * yi;
  }
}

// ColPivHouseholderQR::computeInPlace.  qr: in = the matrix, out = R in the upper triangle, essential parts below;
// h = the Householder coefficients, perm = indices of the column permutation (P(perm[i], i) = 1).
// Work storage of a decomposition, reusable across calls (the aggregation of a scene decomposes thousands of small matrices)
struct Scratch {
  Mat qr, w;
  std::vector<double> h, upd, dir, tmp;
  std::vector<int> perm, transp;
};

inline void colpiv_householder_qr(Mat &qr, std::vector<double> &h, std::vector<int> &perm, Scratch &sc) {
  const int rows = qr.rows, cols = qr.cols, size = rows < cols ? rows : cols;
  h.assign((size_t)size, 0.0);
  std::vector<int> &transp = sc.transp;
  std::vector<double> &upd = sc.upd, &dir = sc.dir, &tmp = sc.tmp;
  transp.assign((size_t)cols, 0);
  upd.assign((size_t)cols, 0.0);
  dir.assign((size_t)cols, 0.0);
  tmp.assign((size_t)cols, 0.0);
  auto col_norm = [&](int j, int r0) {
    double s = 0.0;
    for (int i = r0; i < rows; ++i) s += qr(i, j) * qr(i, j);
    return std::sqrt(s);
  };
  for (int k = 0; k < cols; ++k) upd[(size_t)k] = dir[(size_t)k] = col_norm(k, 0);
  const double norm_downdate_threshold = std::sqrt(DBL_EPSILON);
  for (int k = 0; k < size; ++k) {
    int big = k;
    for (int j = k + 1; j < cols; ++j)
      if (upd[(size_t)j] > upd[(size_t)big]) big = j;  // maxCoeff(&index): first maximum
    transp[(size_t)k] = big;
    if (k != big) {
      for (int i = 0; i < rows; ++i) std::swap(qr(i, k), qr(i, big));
      std::swap(upd[(size_t)k], upd[(size_t)big]);
      std::swap(dir[(size_t)k], dir[(size_t)big]);
    }
    // makeHouseholderInPlace on col(k).tail(rows - k)
    double tau, beta;
    {
      double tail_sq = 0.0;
      for (int i = k + 1; i < rows; ++i) tail_sq += qr(i, k) * qr(i, k);
      const double c0 = qr(k, k);
      if (tail_sq <= DBL_MIN) {
        tau = 0.0;
        beta = c0;
        for (int i = k + 1; i < rows; ++i) qr(i, k) = 0.0;
      } else {
        beta = std::sqrt(c0 * c0 + tail_sq);
        if (c0 >= 0.0) beta = -beta;
        const double den = c0 - beta;
        for (int i = k + 1; i < rows; ++i) qr(i, k) = qr(i, k) / den;
        tau = (beta - c0) / beta;
      }
    }
    h[(size_t)k] = tau;
    qr(k, k) = beta;
    // bottomRightCorner(rows - k, cols - k - 1).applyHouseholderOnTheLeft(essential, tau, workspace)
    if (cols - k - 1 > 0) {
      if (rows - k == 1) {
        for (int j = k + 1; j < cols; ++j) qr(k, j) *= 1.0 - tau;
      } else if (tau != 0.0) {
        for (int j = k + 1; j < cols; ++j) {
          double t = 0.0;
          for (int i = k + 1; i < rows; ++i) t += qr(i, k) * qr(i, j);
          tmp[(size_t)j] = t + qr(k, j);
        }
        for (int j = k + 1; j < cols; ++j) qr(k, j) -= tau * tmp[(size_t)j];
        for (int j = k + 1; j < cols; ++j)
          for (int i = k + 1; i < rows; ++i) qr(i, j) -= (tau * qr(i, k)) * tmp[(size_t)j];
      }
    }
    // norm downdate (LAPACK xGEQPF / lawn176)
    for (int j = k + 1; j < cols; ++j) {
      if (upd[(size_t)j] != 0.0) {
        double temp = std::fabs(qr(k, j)) / upd[(size_t)j];
        temp = (1.0 + temp) * (1.0 - temp);
        temp = temp < 0.0 ? 0.0 : temp;
        const double ratio = upd[(size_t)j] / dir[(size_t)j];
        const double temp2 = temp * (ratio * ratio);
        if (temp2 <= norm_downdate_threshold) {
          dir[(size_t)j] = col_norm(j, k + 1);
          upd[(size_t)j] = dir[(size_t)j];
        } else {
          upd[(size_t)j] *= std::sqrt(temp);
        }
      }
    }
  }
  perm.resize((size_t)cols);
  for (int i = 0; i < cols; ++i) perm[(size_t)i] = i;
  for (int k = 0; k < size; ++k) std::swap(perm[(size_t)k], perm[(size_t)transp[(size_t)k]]);  // applyTranspositionOnTheRight
}

// JacobiSVD<MatrixXd>(A, ComputeThinV): V (cols x min(rows, cols), column-major) and the singular values, sorted.
// Returns false for a matrix with a non-finite entry (Eigen: InvalidInput; V is then left as allocated -- here zero).
inline bool jacobi_svd_thin_v(const Mat &A, Mat &V, std::vector<double> &sv, Scratch &sc) {
  const int rows = A.rows, cols = A.cols, n = rows < cols ? rows : cols;
  V.reset(cols, n);
  sv.assign((size_t)n, 0.0);
  const double precision = 2.0 * DBL_EPSILON;
  const double consider_as_zero = DBL_MIN;
  double scale = 0.0;
  for (double x : A.a) {
    const double ax = std::fabs(x);
    if (!(ax <= scale)) scale = ax;  // maxCoeff<PropagateNaN>
    if (scale != scale) break;
  }
  if (!std::isfinite(scale)) return false;
  if (scale == 0.0) scale = 1.0;
  Mat &w = sc.w, &qr = sc.qr;
  std::vector<double> &h = sc.h;
  std::vector<int> &perm = sc.perm;
  w.reset(n, n);
  if (rows > cols) {
    qr.reset(rows, cols);
    for (size_t k = 0; k < A.a.size(); ++k) qr.a[k] = A.a[k] / scale;
    colpiv_householder_qr(qr, h, perm, sc);
    for (int j = 0; j < n; ++j)
      for (int i = 0; i <= j; ++i) w(i, j) = qr(i, j);
    for (int i = 0; i < cols; ++i) V(perm[(size_t)i], i) = 1.0;
  } else if (cols > rows) {
    qr.reset(cols, rows);  // the adjoint
    for (int i = 0; i < rows; ++i)
      for (int j = 0; j < cols; ++j) qr(j, i) = A(i, j) / scale;
    colpiv_householder_qr(qr, h, perm, sc);
    for (int j = 0; j < n; ++j)
      for (int i = 0; i <= j; ++i) w(j, i) = qr(i, j);  // R^T
    // V = Q applied to the (cols x rows) identity: householderQ().applyThisOnTheLeft -- H_{n-1} first, H_0 last, each
    // on the bottom (cols - k) rows
    for (int i = 0; i < n; ++i) V(i, i) = 1.0;
    std::vector<double> &tmp = sc.tmp;
    tmp.assign((size_t)n, 0.0);
    for (int k = n - 1; k >= 0; --k) {
      const double tau = h[(size_t)k];
      const int sub = cols - k;  // rows of the block the reflector acts on
      if (sub == 1) {
        for (int j = 0; j < n; ++j) V(k, j) *= 1.0 - tau;
      } else if (tau != 0.0) {
        for (int j = 0; j < n; ++j) {
          double t = 0.0;
          for (int i = k + 1; i < cols; ++i) t += qr(i, k) * V(i, j);
          tmp[(size_t)j] = t + V(k, j);
        }
        for (int j = 0; j < n; ++j) V(k, j) -= tau * tmp[(size_t)j];
        for (int j = 0; j < n; ++j)
          for (int i = k + 1; i < cols; ++i) V(i, j) -= (tau * qr(i, k)) * tmp[(size_t)j];
      }
    }
  } else {
    for (size_t k = 0; k < A.a.size(); ++k) w.a[k] = A.a[k] / scale;
    for (int i = 0; i < n; ++i) V(i, i) = 1.0;
  }
  double max_diag = 0.0;
  for (int i = 0; i < n; ++i) max_diag = std::fabs(w(i, i)) > max_diag ? std::fabs(w(i, i)) : max_diag;
  bool finished = false;
  while (!finished) {
    finished = true;
    for (int p = 1; p < n; ++p)
      for (int q = 0; q < p; ++q) {
        const double pm = precision * max_diag;
        const double threshold = consider_as_zero > pm ? consider_as_zero : pm;
        if (std::fabs(w(p, q)) > threshold || std::fabs(w(q, p)) > threshold) {
          finished = false;
          Rot jl, jr;
          real_2x2_jacobi_svd(w(p, p), w(p, q), w(q, p), w(q, q), jl, jr);
          rotate(&w.a[(size_t)p], n, &w.a[(size_t)q], n, n, jl);                        // w.applyOnTheLeft(p, q, j_left): rows
          const Rot jrt{jr.c, -jr.s};                                                    // applyOnTheRight uses j.transpose()
          rotate(&w.a[(size_t)p * n], 1, &w.a[(size_t)q * n], 1, n, jrt);              // w.applyOnTheRight(p, q, j_right): columns
          rotate(&V.a[(size_t)p * cols], 1, &V.a[(size_t)q * cols], 1, cols, jrt);     // V.applyOnTheRight(p, q, j_right)
          const double dp = std::fabs(w(p, p)), dq = std::fabs(w(q, q));
          const double m2 = dp > dq ? dp : dq;
          max_diag = max_diag > m2 ? max_diag : m2;
        }
      }
  }
  for (int i = 0; i < n; ++i) sv[(size_t)i] = std::fabs(w(i, i)) * scale;  // (a negative diagonal flips U's column only)
  for (int i = 0; i < n; ++i) {
    int pos = i;
    for (int k = i + 1; k < n; ++k)
      if (sv[(size_t)k] > sv[(size_t)pos]) pos = k;
    if (sv[(size_t)pos] == 0.0) break;
    if (pos != i) {
      std::swap(sv[(size_t)i], sv[(size_t)pos]);
      for (int r = 0; r < cols; ++r) std::swap(V(r, i), V(r, pos));
    }
  }
  return true;
}

inline bool jacobi_svd_thin_v(const Mat &A, Mat &V, std::vector<double> &sv) {
  Scratch sc;
  return jacobi_svd_thin_v(A, V, sv, sc);
}

}  // namespace lt_svd
