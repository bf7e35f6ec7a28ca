// oracle/ref_shim/colmap/scene/camera.h -- stand-in for colmap::Camera restricted to the pinhole models
// (TEST INFRASTRUCTURE; see colmap/sensor/models.h of this directory).  Member names and meanings follow COLMAP
// 3.9's struct Camera (camera_id, model_id, width, height, params, has_prior_focal_length): for these models
// CalibrationMatrix() is [[fx 0 cx][0 fy cy][0 0 1]], FocalLength() = params[0], Rescale scales the focal
// lengths and principal point by the per-axis size ratio.
#pragma once
#include <cstddef>
#include <vector>

#include <Eigen/Core>
#include <colmap/sensor/models.h>
#include <colmap/util/types.h>

namespace colmap {
struct Camera {
  camera_t camera_id = static_cast<camera_t>(-1);
  CameraModelId model_id = CameraModelId::kInvalid;
  size_t width = 0;
  size_t height = 0;
  std::vector<double> params;
  bool has_prior_focal_length = false;

  std::vector<size_t> FocalLengthIdxs() const {
    return model_id == CameraModelId::kSimplePinhole ? std::vector<size_t>{0} : std::vector<size_t>{0, 1};
  }
  std::vector<size_t> PrincipalPointIdxs() const {
    return model_id == CameraModelId::kSimplePinhole ? std::vector<size_t>{1, 2} : std::vector<size_t>{2, 3};
  }
  double FocalLength() const { return params[0]; }
  double FocalLengthX() const { return params[0]; }
  double FocalLengthY() const { return model_id == CameraModelId::kSimplePinhole ? params[0] : params[1]; }
  double PrincipalPointX() const { return params[PrincipalPointIdxs()[0]]; }
  double PrincipalPointY() const { return params[PrincipalPointIdxs()[1]]; }
  Eigen::Matrix3d CalibrationMatrix() const {
    Eigen::Matrix3d K = Eigen::Matrix3d::Identity();
    K(0, 0) = FocalLengthX();
    K(1, 1) = FocalLengthY();
    K(0, 2) = PrincipalPointX();
    K(1, 2) = PrincipalPointY();
    return K;
  }
  bool VerifyParams() const {
    return (model_id == CameraModelId::kSimplePinhole || model_id == CameraModelId::kPinhole) &&
           params.size() == CameraModelNumParams(model_id);
  }
  bool IsUndistorted() const { return true; }  // both models of this shim are distortion-free
  void Rescale(size_t new_width, size_t new_height) {
    const double sx = static_cast<double>(new_width) / static_cast<double>(width);
    const double sy = static_cast<double>(new_height) / static_cast<double>(height);
    width = new_width;
    height = new_height;
    if (model_id == CameraModelId::kSimplePinhole) {
      params[0] *= (sx + sy) / 2.0;
      params[1] *= sx;
      params[2] *= sy;
    } else {
      params[0] *= sx;
      params[1] *= sy;
      params[2] *= sx;
      params[3] *= sy;
    }
  }
};
}  // namespace colmap
