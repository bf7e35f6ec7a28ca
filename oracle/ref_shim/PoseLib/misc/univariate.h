// oracle/ref_shim/PoseLib/misc/univariate.h -- stand-in for PoseLib's real quartic root finder
// (TEST INFRASTRUCTURE; PoseLib a84c545a9895e46d12a3f5ccde2581c25e6a6953 is what the reference pins,
// cmake/FindDependencies.cmake).  Same contract as poselib::univariate::solve_quartic_real: the real roots of
// x^4 + b x^3 + c x^2 + d x + e, count returned.  Own method: depressed quartic, resolvent cubic (trigonometric /
// Cardano), two quadratics, every root polished by Newton steps on the monic quartic.  Roots agree with
// PoseLib's to rounding, not bit for bit -- which is why everything downstream of the one-point proposal is
// compared with a tolerance.
#pragma once
#include <algorithm>
#include <cmath>

namespace poselib {
namespace univariate {

inline int solve_quadratic_real(double a, double b, double c, double roots[2]) {
  if (a == 0.0) {
    if (b == 0.0) return 0;
    roots[0] = -c / b;
    return 1;
  }
  const double disc = b * b - 4.0 * a * c;
  if (disc < 0.0) return 0;
  const double sq = std::sqrt(disc);
  const double q = -0.5 * (b + (b >= 0 ? sq : -sq));
  roots[0] = q / a;
  roots[1] = q != 0.0 ? c / q : roots[0];
  return 2;
}

// one real root of z^3 + a2 z^2 + a1 z + a0, the LARGEST one
inline double largest_cubic_root(double a2, double a1, double a0) {
  const double q = (a2 * a2 - 3.0 * a1) / 9.0;
  const double r = (2.0 * a2 * a2 * a2 - 9.0 * a2 * a1 + 27.0 * a0) / 54.0;
  double z;
  if (r * r < q * q * q) {
    const double th = std::acos(std::max(-1.0, std::min(1.0, r / std::sqrt(q * q * q))));
    const double m = -2.0 * std::sqrt(q);
    const double z0 = m * std::cos(th / 3.0) - a2 / 3.0;
    const double z1 = m * std::cos((th + 2.0 * M_PI) / 3.0) - a2 / 3.0;
    const double z2 = m * std::cos((th - 2.0 * M_PI) / 3.0) - a2 / 3.0;
    z = std::max(z0, std::max(z1, z2));
  } else {
    const double A = -std::copysign(std::cbrt(std::abs(r) + std::sqrt(r * r - q * q * q)), r);
    const double B = A != 0.0 ? q / A : 0.0;
    z = (A + B) - a2 / 3.0;
  }
  for (int it = 0; it < 8; ++it) {  // polish
    const double f = ((z + a2) * z + a1) * z + a0;
    const double df = (3.0 * z + 2.0 * a2) * z + a1;
    if (df == 0.0) break;
    const double dz = f / df;
    z -= dz;
    if (std::abs(dz) <= 1e-16 * std::abs(z)) break;
  }
  return z;
}

inline int solve_quartic_real(double b, double c, double d, double e, double roots[4]) {
  // depressed form y^4 + p y^2 + q y + r, x = y - b/4
  const double b2 = b * b;
  const double p = c - 0.375 * b2;
  const double q = d - 0.5 * b * c + 0.125 * b2 * b;
  const double r = e - 0.25 * b * d + 0.0625 * b2 * c - (3.0 / 256.0) * b2 * b2;
  int n = 0;
  const double scale = std::max(std::abs(p), std::max(std::cbrt(q * q), std::sqrt(std::abs(r))));
  if (std::abs(q) <= 1e-14 * scale * std::sqrt(scale)) {  // biquadratic
    double t[2];
    const int nt = solve_quadratic_real(1.0, p, r, t);
    for (int k = 0; k < nt; ++k)
      if (t[k] >= 0.0) {
        const double s = std::sqrt(t[k]);
        roots[n++] = s;
        roots[n++] = -s;
      }
  } else {
    // resolvent z^3 + 2p z^2 + (p^2 - 4r) z - q^2 = 0 has a positive root
    const double z = largest_cubic_root(2.0 * p, p * p - 4.0 * r, -q * q);
    if (!(z > 0.0)) return 0;
    const double s = std::sqrt(z);
    double t[2];
    int nt = solve_quadratic_real(1.0, s, 0.5 * (p + z - q / s), t);
    for (int k = 0; k < nt; ++k) roots[n++] = t[k];
    nt = solve_quadratic_real(1.0, -s, 0.5 * (p + z + q / s), t);
    for (int k = 0; k < nt; ++k) roots[n++] = t[k];
  }
  for (int k = 0; k < n; ++k) {
    double x = roots[k] - 0.25 * b;
    for (int it = 0; it < 10; ++it) {
      const double f = (((x + b) * x + c) * x + d) * x + e;
      const double df = ((4.0 * x + 3.0 * b) * x + 2.0 * c) * x + d;
      if (df == 0.0) break;
      const double dx = f / df;
      x -= dx;
      if (std::abs(dx) <= 1e-16 * std::abs(x)) break;
    }
    roots[k] = x;
  }
  return n;
}

}  // namespace univariate
}  // namespace poselib
