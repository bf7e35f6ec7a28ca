// oracle/ref_shim/colmap/sensor/models.h -- stand-in for the part of COLMAP's camera-model registry that
// limap::Camera reaches on the triangulation path (TEST INFRASTRUCTURE).  Only the two UNDISTORTED models exist
// here -- BaseLineTriangulator::Init requires them (triangulation/base_line_triangulator.cc:49): SIMPLE_PINHOLE
// (f, cx, cy) and PINHOLE (fx, fy, cx, cy), parameter orders as in COLMAP's sensor/models.h.
#pragma once
#include <stdexcept>
#include <string>
#include <vector>

namespace colmap {
enum class CameraModelId { kInvalid = -1, kSimplePinhole = 0, kPinhole = 1 };

struct SimplePinholeCameraModel {
  static constexpr CameraModelId model_id = CameraModelId::kSimplePinhole;
};
struct PinholeCameraModel {
  static constexpr CameraModelId model_id = CameraModelId::kPinhole;
};

inline CameraModelId CameraModelNameToId(const std::string &name) {
  if (name == "SIMPLE_PINHOLE") return CameraModelId::kSimplePinhole;
  if (name == "PINHOLE") return CameraModelId::kPinhole;
  throw std::domain_error("ref_shim: only SIMPLE_PINHOLE / PINHOLE camera models exist here (" + name + ")");
}
inline std::string CameraModelIdToName(CameraModelId id) {
  if (id == CameraModelId::kSimplePinhole) return "SIMPLE_PINHOLE";
  if (id == CameraModelId::kPinhole) return "PINHOLE";
  return "INVALID";
}
inline size_t CameraModelNumParams(CameraModelId id) {
  if (id == CameraModelId::kSimplePinhole) return 3;
  if (id == CameraModelId::kPinhole) return 4;
  throw std::domain_error("ref_shim: camera model does not exist");
}
inline std::vector<double> CameraModelInitializeParams(CameraModelId id, double f, size_t width, size_t height) {
  if (id == CameraModelId::kSimplePinhole) return {f, width / 2.0, height / 2.0};
  if (id == CameraModelId::kPinhole) return {f, f, width / 2.0, height / 2.0};
  throw std::domain_error("ref_shim: camera model does not exist");
}
}  // namespace colmap
