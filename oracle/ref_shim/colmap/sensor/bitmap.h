// oracle/ref_shim/colmap/sensor/bitmap.h -- colmap::Bitmap without an image library: nothing can be read
// (TEST INFRASTRUCTURE).  Only CameraView::get_initial_focal_length touches it (base/camera_view.cc:84-100),
// which is off the triangulation path.
#pragma once
#include <string>
namespace colmap {
class Bitmap {
 public:
  bool Read(const std::string &, bool = true) { return false; }
  int Width() const { return 1; }
  int Height() const { return 1; }
  bool ExifFocalLength(double *) const { return false; }
};
}  // namespace colmap
