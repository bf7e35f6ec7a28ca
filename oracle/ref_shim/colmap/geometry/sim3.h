// oracle/ref_shim/colmap/geometry/sim3.h -- colmap::Sim3d as the plain aggregate COLMAP 3.9 declares
// (TEST INFRASTRUCTURE); only ImageCollection::apply_similarity_transform reads it (off the path).
#pragma once
#include <Eigen/Core>
#include <Eigen/Geometry>
namespace colmap {
struct Sim3d {
  double scale = 1;
  Eigen::Quaterniond rotation = Eigen::Quaterniond::Identity();
  Eigen::Vector3d translation = Eigen::Vector3d::Zero();
};
}  // namespace colmap
