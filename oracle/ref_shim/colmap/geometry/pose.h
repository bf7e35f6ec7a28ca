// oracle/ref_shim/colmap/geometry/pose.h -- limap/base/pose.h includes this header only for the Eigen types
// (TEST INFRASTRUCTURE).
#pragma once
#include <Eigen/Core>
#include <Eigen/Geometry>
