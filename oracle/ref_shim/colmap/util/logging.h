// oracle/ref_shim/colmap/util/logging.h -- stand-in for COLMAP's THROW_CHECK* / glog CHECK* macros
// (TEST INFRASTRUCTURE).  THROW_CHECK* throw std::invalid_argument like COLMAP's; CHECK* (glog: abort) throw
// std::logic_error here so that a test can observe them.
#pragma once
#include <iostream>
#include <sstream>
#include <stdexcept>
#include <string>

namespace colmap_shim {
struct NullStream {
  template <class T>
  NullStream &operator<<(const T &) { return *this; }
  NullStream &operator<<(std::ostream &(*)(std::ostream &)) { return *this; }
};
template <class E>
[[noreturn]] inline void fail(const char *expr, const char *file, int line) {
  std::ostringstream s;
  s << "[" << file << ":" << line << "] Check failed: " << expr;
  throw E(s.str());
}
}  // namespace colmap_shim

#define SHIM_CHECK_(E, cond, text) \
  if (!(cond)) colmap_shim::fail<E>(text, __FILE__, __LINE__)
#define THROW_CHECK(c) SHIM_CHECK_(std::invalid_argument, (c), #c)
#define THROW_CHECK_EQ(a, b) SHIM_CHECK_(std::invalid_argument, (a) == (b), #a " == " #b)
#define THROW_CHECK_NE(a, b) SHIM_CHECK_(std::invalid_argument, (a) != (b), #a " != " #b)
#define THROW_CHECK_LT(a, b) SHIM_CHECK_(std::invalid_argument, (a) < (b), #a " < " #b)
#define THROW_CHECK_LE(a, b) SHIM_CHECK_(std::invalid_argument, (a) <= (b), #a " <= " #b)
#define THROW_CHECK_GT(a, b) SHIM_CHECK_(std::invalid_argument, (a) > (b), #a " > " #b)
#define THROW_CHECK_GE(a, b) SHIM_CHECK_(std::invalid_argument, (a) >= (b), #a " >= " #b)
#define THROW_CHECK_NOTNULL(p) (p)
#define CHECK(c) SHIM_CHECK_(std::logic_error, (c), #c)
#define CHECK_EQ(a, b) SHIM_CHECK_(std::logic_error, (a) == (b), #a " == " #b)
#define CHECK_NE(a, b) SHIM_CHECK_(std::logic_error, (a) != (b), #a " != " #b)
#define CHECK_LT(a, b) SHIM_CHECK_(std::logic_error, (a) < (b), #a " < " #b)
#define CHECK_LE(a, b) SHIM_CHECK_(std::logic_error, (a) <= (b), #a " <= " #b)
#define CHECK_GT(a, b) SHIM_CHECK_(std::logic_error, (a) > (b), #a " > " #b)
#define CHECK_GE(a, b) SHIM_CHECK_(std::logic_error, (a) >= (b), #a " >= " #b)
#define CHECK_NOTNULL(p) (p)
#define LOG(level) colmap_shim::NullStream()
#define VLOG(level) colmap_shim::NullStream()
