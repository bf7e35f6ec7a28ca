// oracle/ref_shim/colmap/util/types.h -- stand-in for COLMAP's id typedefs (TEST INFRASTRUCTURE; COLMAP
// 1443d525960551e11b71f7fa5d11c76f5c2fab42 is the version the reference pins, cmake/FindDependencies.cmake:61)
#pragma once
#include <cstdint>
#include <Eigen/Core>
#include <Eigen/Geometry>
#include <Eigen/SVD>  // COLMAP's headers pull in the dense modules the reference's files rely on
namespace colmap {
using camera_t = uint32_t;
using image_t = uint32_t;
using point2D_t = uint32_t;
using point3D_t = uint64_t;
}  // namespace colmap
