// oracle/ref_shim/pybind11/eigen.h -- shadows pybind11's Eigen casters (which need the real Eigen) with a
// numpy <-> Eigen-stand-in caster, so that the reference's dict constructors / as_dict() methods compile and
// work (TEST INFRASTRUCTURE).  Vectors map to 1-D arrays, matrices to 2-D arrays, like pybind11/eigen.h.
#pragma once
#include <Eigen/Core>
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>

namespace pybind11 {
namespace detail {
template <class T, int R, int C>
struct type_caster<Eigen::Matrix<T, R, C>> {
  using M = Eigen::Matrix<T, R, C>;
  PYBIND11_TYPE_CASTER(M, const_name("numpy.ndarray"));
  bool load(handle src, bool) {
    auto arr = array_t<T, array::c_style | array::forcecast>::ensure(src);
    if (!arr) return false;
    if (arr.ndim() == 1) {
      const Eigen::Index n = arr.shape(0);
      if (R != Eigen::Dynamic && C != Eigen::Dynamic && R * C != n) return false;
      if (C == 1 || (C == Eigen::Dynamic && R == Eigen::Dynamic)) value.resize(n, 1); else value.resize(1, n);
      for (Eigen::Index k = 0; k < n; ++k) value.data()[k] = arr.at(k);
      return true;
    }
    if (arr.ndim() != 2) return false;
    const Eigen::Index r = arr.shape(0), c = arr.shape(1);
    if ((R != Eigen::Dynamic && R != r) || (C != Eigen::Dynamic && C != c)) return false;
    value.resize(r, c);
    for (Eigen::Index i = 0; i < r; ++i)
      for (Eigen::Index j = 0; j < c; ++j) value(i, j) = arr.at(i, j);
    return true;
  }
  static handle cast(const M &m, return_value_policy, handle) {
    if (R != Eigen::Dynamic && C != Eigen::Dynamic && (R == 1 || C == 1)) {
      array_t<T> a(static_cast<size_t>(m.size()));
      for (Eigen::Index k = 0; k < m.size(); ++k) a.mutable_at(k) = m.data()[k];
      return a.release();
    }
    array_t<T> a({static_cast<size_t>(m.rows()), static_cast<size_t>(m.cols())});
    for (Eigen::Index i = 0; i < m.rows(); ++i)
      for (Eigen::Index j = 0; j < m.cols(); ++j) a.mutable_at(i, j) = m.coeff(i, j);
    return a.release();
  }
};
}  // namespace detail
}  // namespace pybind11
