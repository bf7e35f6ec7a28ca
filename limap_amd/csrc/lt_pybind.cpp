// lt_pybind.cpp -- the thin pybind11 shim over the C ABI of include/limap_amd.h (module limap_amd._lt_pybind).
//
// Mirrors the pybind surface of the reference's `limap._limap._triangulation.GlobalLineTriangulator`
// (src/limap/triangulation/bindings.cc:78-119) at the level of plain numpy arrays: same method names, same
// argument meaning, exceptions of the same Python types (std::runtime_error -> RuntimeError, COLMAP THROW_CHECK /
// std::out_of_range -> ValueError / IndexError).  No arithmetic lives here: every call forwards to one lt_* entry
// point, with the GIL released around the blocking ones (the reference holds it, bindings.cc has no
// gil_scoped_release; a caller that runs the matcher in another thread gains).  limap's own value types
// (Line2d, ImageCollection, VPResult, PL_Bipartite2d ...) are duck-typed one level up, in
// limap_amd/triangulation.py, which uses this module for the per-image calls when it is importable.
//
// Build: limap_amd/csrc/Makefile (g++, pybind11 headers of the installed package, links liblimap_amd.so).
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/limap_amd.h"

namespace py = pybind11;

namespace {

template <class T>
using carr = py::array_t<T, py::array::c_style | py::array::forcecast>;

void assign_cfg(lt_config &c, const py::dict &d) {
  // GlobalLineTriangulatorConfig(py::dict): present keys overwrite the defaults, unknown keys are ignored
  // (internal/helpers.h:25-27; base_line_triangulator.cc:16-31; global_line_triangulator.cc:18-29)
#define LT_KEY(name, type) \
  if (d.contains(#name) && !d[#name].is_none()) c.name = (decltype(c.name))d[#name].cast<type>();
  LT_KEY(debug_mode, bool) LT_KEY(add_halfpix, bool) LT_KEY(use_vp, bool) LT_KEY(use_endpoints_triangulation, bool)
  LT_KEY(disable_many_points_triangulation, bool) LT_KEY(disable_one_point_triangulation, bool)
  LT_KEY(disable_algebraic_triangulation, bool) LT_KEY(disable_vp_triangulation, bool)
  LT_KEY(min_length_2d, double) LT_KEY(line_tri_angle_threshold, double) LT_KEY(IoU_threshold, double)
  LT_KEY(sensitivity_threshold, double) LT_KEY(var2d, double) LT_KEY(fullscore_th, double)
  LT_KEY(max_valid_conns, int) LT_KEY(min_num_outer_edges, int) LT_KEY(num_outliers_aggregator, int)
#undef LT_KEY
  if (d.contains("merging_strategy") && !d["merging_strategy"].is_none()) {
    const std::string s = d["merging_strategy"].cast<std::string>();
    c.merging_strategy = s == "greedy" ? 0 : (s == "exhaustive" ? 1 : (s == "avg" ? 2 : 99));
  }
#define LT_SUB(prefix, name, type) \
  if (sub.contains(#name)) c.prefix##name = (decltype(c.prefix##name))sub[#name].cast<type>();
  if (d.contains("linker2d_config") && !d["linker2d_config"].is_none()) {
    py::dict sub = d["linker2d_config"];
    LT_SUB(l2_, score_th, double) LT_SUB(l2_, th_angle, double) LT_SUB(l2_, th_overlap, double)
    LT_SUB(l2_, th_smartoverlap, double) LT_SUB(l2_, th_smartangle, double) LT_SUB(l2_, th_perp, double)
    LT_SUB(l2_, th_innerseg, double) LT_SUB(l2_, use_angle, bool) LT_SUB(l2_, use_overlap, bool)
    LT_SUB(l2_, use_smartangle, bool) LT_SUB(l2_, use_perp, bool) LT_SUB(l2_, use_innerseg, bool)
  }
  if (d.contains("linker3d_config") && !d["linker3d_config"].is_none()) {
    py::dict sub = d["linker3d_config"];
    LT_SUB(l3_, score_th, double) LT_SUB(l3_, th_angle, double) LT_SUB(l3_, th_overlap, double)
    LT_SUB(l3_, th_smartoverlap, double) LT_SUB(l3_, th_smartangle, double) LT_SUB(l3_, th_perp, double)
    LT_SUB(l3_, th_innerseg, double) LT_SUB(l3_, th_scaleinv, double) LT_SUB(l3_, use_angle, bool)
    LT_SUB(l3_, use_overlap, bool) LT_SUB(l3_, use_smartangle, bool) LT_SUB(l3_, use_perp, bool)
    LT_SUB(l3_, use_innerseg, bool) LT_SUB(l3_, use_scaleinv, bool)
  }
#undef LT_SUB
}

class Triangulator {
 public:
  // GlobalLineTriangulator(dict) -- bindings.cc:79-80
  explicit Triangulator(const py::dict &cfg, int device = 0) : own_(true) {
    lt_config c;
    lt_config_default(&c);
    assign_cfg(c, cfg);
    ctx_ = lt_create(&c, device);
    if (!ctx_) throw std::runtime_error("limap_amd: lt_create failed -- no usable HIP device (this backend has no CPU fallback)");
  }
  // non-owning view of a context created elsewhere (limap_amd._capi.Context): lets the Python mirror route its
  // per-image calls through this module while the rarely used entry points stay on ctypes
  explicit Triangulator(uintptr_t handle) : ctx_(reinterpret_cast<lt_ctx *>(handle)), own_(false) {
    if (!ctx_) throw std::invalid_argument("null lt_ctx handle");
  }
  ~Triangulator() {
    if (own_ && ctx_) {
      py::gil_scoped_release nogil;
      lt_destroy(ctx_);
    }
  }
  Triangulator(const Triangulator &) = delete;
  Triangulator &operator=(const Triangulator &) = delete;

  uintptr_t handle() const { return reinterpret_cast<uintptr_t>(ctx_); }

  void chk(int rc) const {
    if (rc == LT_OK) return;
    const std::string msg = lt_last_error(ctx_);
    if (rc == LT_ERR_ARGUMENT) {
      if (msg.rfind("unknown", 0) == 0) throw py::index_error(msg);  // std::map::at on an image id
      throw py::value_error(msg);                                     // THROW_CHECK_*
    }
    throw std::runtime_error(msg);
  }

  void SetRanges(const std::pair<carr<double>, carr<double>> &r) {  // bindings.cc:94
    if (r.first.size() != 3 || r.second.size() != 3) throw py::value_error("ranges must be two 3-vectors");
    chk(lt_set_ranges(ctx_, r.first.data(), r.second.data()));
  }
  void UnsetRanges() { chk(lt_unset_ranges(ctx_)); }

  // Init on flat arrays (the C ABI's form of Init(all_2d_segs, imagecols), bindings.cc:81)
  void InitArrays(carr<int32_t> ids, carr<double> kvec, carr<double> qvec, carr<double> tvec, carr<int64_t> seg_off,
                  carr<double> segs) {
    const int n = (int)ids.size();
    if (kvec.size() != 4 * (py::ssize_t)n || qvec.size() != 4 * (py::ssize_t)n || tvec.size() != 3 * (py::ssize_t)n ||
        seg_off.size() != n + 1)
      throw py::value_error("InitArrays: kvec (n,4), qvec (n,4), tvec (n,3), seg_off (n+1) expected");
    if (segs.size() != 4 * seg_off.data()[n]) throw py::value_error("InitArrays: segs must have seg_off[-1] rows of 4");
    int rc;
    {
      py::gil_scoped_release nogil;
      rc = lt_init(ctx_, n, ids.data(), kvec.data(), qvec.data(), tvec.data(), seg_off.data(), segs.data());
    }
    chk(rc);
  }

  // TriangulateImage(img_id, matches: dict[int -> ndarray (K,2) int]) -- bindings.cc:83; arrays that are not
  // C-contiguous int32 are converted by copy, like pybind11's Eigen::MatrixXi caster does
  void TriangulateImage(int img_id, const py::dict &matches) {
    const size_t n = matches.size();
    std::vector<int32_t> nb;
    std::vector<const int32_t *> rows;
    std::vector<int64_t> cnt;
    std::vector<carr<int32_t>> keep;  // owners of the (possibly converted) arrays
    nb.reserve(n); rows.reserve(n); cnt.reserve(n); keep.reserve(n);
    for (auto item : matches) {
      nb.push_back(item.first.cast<int32_t>());
      py::array a = py::array::ensure(item.second);
      if (!a) throw py::value_error("matches: array expected");
      if (a.size() != 0 && (a.ndim() != 2 || a.shape(1) != 2))
        throw py::value_error("Check failed: match_info.cols() == 2");  // base_line_triangulator.cc:79
      keep.emplace_back(carr<int32_t>::ensure(a));
      if (!keep.back()) throw py::value_error("matches: integer array expected");
      rows.push_back(keep.back().data());
      cnt.push_back(a.size() / 2);
    }
    int rc;
    {
      py::gil_scoped_release nogil;
      rc = lt_triangulate_image_rows(ctx_, img_id, (int)n, nb.data(), rows.data(), cnt.data());
    }
    chk(rc);
  }

  // TriangulateAll(matches_by_image: dict[int -> dict[int -> ndarray (K,2) int]]): the caller's TriangulateImage loop
  // (runners/line_triangulation.py:160-167) as one native call -- the images in the dict's order, one pass over all rows
  // (lt_triangulate_all_rows).  No reference counterpart.
  void TriangulateAll(const py::dict &by_image) {
    std::vector<int32_t> ids, nb;
    std::vector<int64_t> nb_off(1, 0), cnt;
    std::vector<const int32_t *> rows;
    std::vector<carr<int32_t>> keep;
    for (auto im : by_image) {
      ids.push_back(im.first.cast<int32_t>());
      if (!py::isinstance<py::dict>(im.second)) throw py::value_error("TriangulateAll: dict of dicts expected");
      py::dict matches = py::reinterpret_borrow<py::dict>(im.second);
      for (auto item : matches) {
        nb.push_back(item.first.cast<int32_t>());
        py::array a = py::array::ensure(item.second);
        if (!a) throw py::value_error("matches: array expected");
        if (a.size() != 0 && (a.ndim() != 2 || a.shape(1) != 2))
          throw py::value_error("Check failed: match_info.cols() == 2");  // base_line_triangulator.cc:79
        keep.emplace_back(carr<int32_t>::ensure(a));
        if (!keep.back()) throw py::value_error("matches: integer array expected");
        rows.push_back(keep.back().data());
        cnt.push_back(a.size() / 2);
      }
      nb_off.push_back((int64_t)nb.size());
    }
    int rc;
    {
      py::gil_scoped_release nogil;
      rc = lt_triangulate_all_rows(ctx_, (int)ids.size(), ids.data(), nb_off.data(), nb.data(), rows.data(), cnt.data());
    }
    chk(rc);
  }

  void TriangulateImageExhaustiveMatch(int img_id, const std::vector<int32_t> &neighbors) {  // bindings.cc:84-85
    chk(lt_triangulate_image_exhaustive(ctx_, img_id, (int)neighbors.size(), neighbors.data()));
  }

  // ComputeLineTracks() -- bindings.cc:88; the tracks as CSR arrays (LineTrack fields, base/linetrack.h:33-42)
  py::dict ComputeLineTracks() {
    int rc;
    {
      py::gil_scoped_release nogil;
      rc = lt_compute_tracks(ctx_);
    }
    chk(rc);
    return GetTracks();
  }
  py::dict GetTracks() {
    const int64_t T = lt_num_tracks(ctx_), M = lt_num_track_members(ctx_);
    py::array_t<double> line({(py::ssize_t)T, (py::ssize_t)7}), scores((py::ssize_t)M), l3d({(py::ssize_t)M, (py::ssize_t)10});
    py::array_t<int64_t> off((py::ssize_t)T + 1);
    py::array_t<int32_t> img((py::ssize_t)M), lid((py::ssize_t)M), nid((py::ssize_t)M);
    // (a zero-size numpy array still has a valid data pointer)
    chk(lt_get_tracks(ctx_, line.mutable_data(), off.mutable_data(), img.mutable_data(), lid.mutable_data(),
                      nid.mutable_data(), scores.mutable_data(), l3d.mutable_data()));
    py::dict d;
    d["line"] = line; d["off"] = off; d["image_ids"] = img; d["line_ids"] = lid; d["node_ids"] = nid;
    d["scores"] = scores; d["line3d"] = l3d;
    return d;
  }

  int64_t CountImages() const { return lt_count_images(ctx_); }  // base_line_triangulator.h:84
  int64_t CountLines(int img_id) const {                          // :85-87 (std::map::at)
    const int64_t n = lt_count_lines(ctx_, img_id);
    if (n < 0) throw py::index_error(lt_last_error(ctx_));
    return n;
  }

  // per-node results as arrays: what GetAllBestTris / GetBestScoredTriNode read (global_line_triangulator.cc:496-541)
  py::dict GetBest() {
    int rc;
    {
      py::gil_scoped_release nogil;
      rc = lt_flush(ctx_);
    }
    chk(rc);
    const py::ssize_t G = (py::ssize_t)lt_num_nodes(ctx_);
    py::array_t<double> line({G, (py::ssize_t)10}), score(G);
    py::array_t<int32_t> src({G, (py::ssize_t)2});
    py::array_t<uint8_t> has(G);
    chk(lt_get_best(ctx_, line.mutable_data(), score.mutable_data(), src.mutable_data(), has.mutable_data()));
    py::dict d;
    d["line"] = line; d["score"] = score; d["src"] = src; d["has_best"] = has;
    return d;
  }

  py::dict Stats() {
    int64_t s[8];
    chk(lt_get_stats(ctx_, s));
    const char *keys[] = {"connections", "candidates", "pairs", "valid_edges", "graph_nodes", "graph_edges", "tracks", "nodes"};
    py::dict d;
    for (int k = 0; k < 8; ++k) d[keys[k]] = s[k];
    return d;
  }

 private:
  lt_ctx *ctx_ = nullptr;
  bool own_ = false;
};

}  // namespace

PYBIND11_MODULE(_lt_pybind, m) {
  m.doc() = "pybind11 shim over liblimap_amd.so (C ABI: include/limap_amd.h); array-level mirror of "
            "limap._limap._triangulation.GlobalLineTriangulator";
  m.def("abi_version", &lt_abi_version);
  py::class_<Triangulator>(m, "GlobalLineTriangulator")
      .def(py::init<const py::dict &, int>(), py::arg("cfg"), py::arg("device") = 0)
      .def(py::init<uintptr_t>(), py::arg("handle"))
      .def_property_readonly("handle", &Triangulator::handle)
      .def("SetRanges", &Triangulator::SetRanges)
      .def("UnsetRanges", &Triangulator::UnsetRanges)
      .def("InitArrays", &Triangulator::InitArrays)
      .def("TriangulateImage", &Triangulator::TriangulateImage)
.def("TriangulateAll", &Triangulator::TriangulateAll)
      .def("TriangulateImageExhaustiveMatch", &Triangulator::TriangulateImageExhaustiveMatch)
      .def("ComputeLineTracks", &Triangulator::ComputeLineTracks)
      .def("GetTracks", &Triangulator::GetTracks)
      .def("CountImages", &Triangulator::CountImages)
      .def("CountLines", &Triangulator::CountLines)
      .def("GetBest", &Triangulator::GetBest)
      .def("Stats", &Triangulator::Stats);
}
