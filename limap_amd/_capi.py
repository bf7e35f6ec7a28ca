"""ctypes binding of liblimap_amd.so (C ABI declared in include/limap_amd.h).

The library is built in-tree by ``limap_amd.build.build_extension()`` (hipcc, gfx950).  There is
no CPU fallback: if the shared object is missing, or no HIP device is visible, every entry point
raises ``RuntimeError``.
"""
import ctypes as C
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# LIMAP_AMD_LIB: developer override (A/B builds of the extension); the default is the in-tree library
LIB_PATH = os.environ.get("LIMAP_AMD_LIB") or os.path.join(_HERE, "liblimap_amd.so")

EXPORTED_SYMBOLS = [
    "lt_config_default", "lt_abi_version", "lt_sizeof_config", "lt_create", "lt_destroy", "lt_last_error", "lt_set_stream", "lt_set_ranges",
    "lt_unset_ranges", "lt_init", "lt_init_vp", "lt_set_bipartites", "lt_set_sfm_points", "lt_init_device", "lt_refresh_scene_device", "lt_set_scene_chunks",
    "lt_refresh_scene_chunks", "lt_triangulate_image", "lt_triangulate_image_rows", "lt_triangulate_all_rows",
    "lt_triangulate_image_exhaustive", "lt_upload", "lt_run_device", "lt_download", "lt_flush",
    "lt_compute_tracks", "lt_compute_tracks_begin", "lt_compute_tracks_end", "lt_count_images", "lt_count_lines", "lt_num_nodes", "lt_get_best",
    "lt_get_num_tris", "lt_get_valid_flags", "lt_num_valid_edges", "lt_get_valid_edges", "lt_num_all_tris", "lt_get_all_tris",
    "lt_num_tracks", "lt_num_track_members", "lt_get_tracks", "lt_image_results_size",
    "lt_export_image_results", "lt_import_image_results", "lt_export_images_size", "lt_export_images_packed", "lt_import_images_packed", "lt_shard_node_bytes", "lt_shard_count", "lt_shard_build", "lt_shard_export", "lt_shard_import", "lt_ts_from_ctx", "lt_ts_create", "lt_ts_destroy",
    "lt_ts_num_tracks", "lt_ts_num_members", "lt_ts_get", "lt_ts_filter_by_reprojection",
    "lt_ts_filter_by_sensitivity", "lt_ts_filter_by_overlap", "lt_ts_remerge_once", "lt_get_stats", "lt_get_timers", "lt_get_timer_sums", "lt_run_device_async", "lt_sync",
    "lt_release_cached_memory", "lt_reserve_host",
    "lt_fn_get_normal_direction", "lt_fn_get_direction_from_vp", "lt_fn_triangulate_point",
    "lt_fn_triangulate_line_with_direction", "lt_fn_triangulate_line_with_one_point", "lt_fn_compute_fundamental_matrix", "lt_fn_compute_epipolar_IoU",
    "lt_fn_triangulate_line", "lt_fn_aggregate_line3d_list", "lt_fn_pack_match_rows",
    "lt_fn_compressed_block_words", "lt_fn_pack_match_rows_compressed",
]


class LtConfig(C.Structure):
    """lt_config of include/limap_amd.h (field for field)."""
    _fields_ = [
        ("debug_mode", C.c_int32), ("add_halfpix", C.c_int32), ("use_vp", C.c_int32),
        ("use_endpoints_triangulation", C.c_int32), ("disable_many_points_triangulation", C.c_int32),
        ("disable_one_point_triangulation", C.c_int32), ("disable_algebraic_triangulation", C.c_int32),
        ("disable_vp_triangulation", C.c_int32),
        ("min_length_2d", C.c_double), ("line_tri_angle_threshold", C.c_double), ("IoU_threshold", C.c_double),
        ("sensitivity_threshold", C.c_double), ("var2d", C.c_double), ("fullscore_th", C.c_double),
        ("max_valid_conns", C.c_int32), ("min_num_outer_edges", C.c_int32), ("merging_strategy", C.c_int32),
        ("num_outliers_aggregator", C.c_int32),
        ("l2_score_th", C.c_double), ("l2_th_angle", C.c_double), ("l2_th_overlap", C.c_double),
        ("l2_th_smartoverlap", C.c_double), ("l2_th_smartangle", C.c_double), ("l2_th_perp", C.c_double),
        ("l2_th_innerseg", C.c_double),
        ("l2_use_angle", C.c_int32), ("l2_use_overlap", C.c_int32), ("l2_use_smartangle", C.c_int32),
        ("l2_use_perp", C.c_int32), ("l2_use_innerseg", C.c_int32), ("_pad0", C.c_int32),
        ("l3_score_th", C.c_double), ("l3_th_angle", C.c_double), ("l3_th_overlap", C.c_double),
        ("l3_th_smartoverlap", C.c_double), ("l3_th_smartangle", C.c_double), ("l3_th_perp", C.c_double),
        ("l3_th_innerseg", C.c_double), ("l3_th_scaleinv", C.c_double),
        ("l3_use_angle", C.c_int32), ("l3_use_overlap", C.c_int32), ("l3_use_smartangle", C.c_int32),
        ("l3_use_perp", C.c_int32), ("l3_use_innerseg", C.c_int32), ("l3_use_scaleinv", C.c_int32),
    ]


BASE_KEYS = [
    "debug_mode", "add_halfpix", "use_vp", "use_endpoints_triangulation",
    "disable_many_points_triangulation", "disable_one_point_triangulation",
    "disable_algebraic_triangulation", "disable_vp_triangulation", "min_length_2d",
    "line_tri_angle_threshold", "IoU_threshold", "sensitivity_threshold", "var2d",
    "fullscore_th", "max_valid_conns", "min_num_outer_edges", "num_outliers_aggregator",
]
L2_KEYS = ["score_th", "th_angle", "th_overlap", "th_smartoverlap", "th_smartangle", "th_perp", "th_innerseg",
           "use_angle", "use_overlap", "use_smartangle", "use_perp", "use_innerseg"]
L3_KEYS = L2_KEYS[:7] + ["th_scaleinv"] + L2_KEYS[7:] + ["use_scaleinv"]
MERGING = {"greedy": 0, "exhaustive": 1, "avg": 2}

_lib = None


def _preload_torch_hip_runtime():
    """One HIP runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64.so (file name without
    version, SONAME libamdhip64.so.7).  If this extension is loaded first it pulls in /opt/rocm's
    libamdhip64.so.7, and a later `import torch` then loads the bundled copy AS WELL (the loader matches
    torch's NEEDED entry by file name, not by SONAME) -- the second runtime finds no usable GPU
    ("No HIP GPUs are available") and streams could not be shared anyway.  Loading torch's copy first makes
    this extension's NEEDED libamdhip64.so.7 resolve to it, whatever the import order.  No torch, no-op."""
    if "torch" in sys.modules:
        return
    # LIMAP_AMD_SYSTEM_HIP=1: a process that will never import torch and uses /opt/rocm's stack throughout (e.g. one that
    # loads liblimap_amd_rccl.so, which links the system's librccl) keeps the system runtime
    if os.environ.get("LIMAP_AMD_SYSTEM_HIP") == "1":
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.submodule_search_locations:
            return
        cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
        if os.path.exists(cand):
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
    except Exception:
        pass  # fall back to the system runtime


def load_library():
    """dlopen liblimap_amd.so and declare the prototypes.  Raises if the extension is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"limap_amd: HIP extension {LIB_PATH} is not built (run `python -c 'import __graft_entry__ as g; "
            "g.build()'` or `make -C limap_amd/csrc`); there is no CPU fallback")
    _preload_torch_hip_runtime()
    L = C.CDLL(LIB_PATH)
    vp, i32p, i64p, dp, u8p = C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_double), C.POINTER(C.c_uint8)
    L.lt_config_default.argtypes = [C.POINTER(LtConfig)]
    L.lt_config_default.restype = None
    L.lt_sizeof_config.restype = C.c_uint64
    if L.lt_sizeof_config() != C.sizeof(LtConfig):
        raise RuntimeError("limap_amd: lt_config layout mismatch between liblimap_amd.so and the Python binding")
    L.lt_create.argtypes = [C.POINTER(LtConfig), C.c_int]
    L.lt_create.restype = vp
    L.lt_destroy.argtypes = [vp]
    L.lt_destroy.restype = None
    L.lt_release_cached_memory.argtypes = []
    L.lt_release_cached_memory.restype = None
    L.lt_reserve_host.argtypes = [C.c_uint64, C.c_int]
    L.lt_reserve_host.restype = C.c_int
    L.lt_last_error.argtypes = [vp]
    L.lt_last_error.restype = C.c_char_p
    L.lt_set_stream.argtypes = [vp, vp]
    L.lt_set_ranges.argtypes = [vp, dp, dp]
    L.lt_unset_ranges.argtypes = [vp]
    L.lt_init.argtypes = [vp, C.c_int, i32p, dp, dp, dp, i64p, dp]
    L.lt_init_vp.argtypes = [vp, C.c_int, i32p, i64p, i32p, i64p, dp]
    L.lt_set_bipartites.argtypes = [vp, C.c_int, i32p, i64p, i32p, dp, i32p, i64p, i64p, i32p]
    L.lt_set_sfm_points.argtypes = [vp, C.c_int64, i32p, dp]
    L.lt_init_device.argtypes = [vp, C.c_int, i32p, vp, vp, vp, i64p, vp]
    L.lt_refresh_scene_device.argtypes = [vp, vp, vp, vp, vp]
    vpp = C.POINTER(C.c_void_p)
    L.lt_set_scene_chunks.argtypes = [vp, C.c_int, i32p, vpp, vpp, vpp, vpp]
    L.lt_refresh_scene_chunks.argtypes = [vp]
    L.lt_triangulate_image.argtypes = [vp, C.c_int, C.c_int, i32p, i64p, i32p]
    L.lt_triangulate_image_rows.argtypes = [vp, C.c_int, C.c_int, i32p, C.POINTER(C.c_void_p), i64p]
    L.lt_triangulate_all_rows.argtypes = [vp, C.c_int, i32p, i64p, i32p, C.POINTER(C.c_void_p), i64p]
    L.lt_triangulate_image_exhaustive.argtypes = [vp, C.c_int, C.c_int, i32p]
    for n in ("lt_upload", "lt_run_device", "lt_run_device_async", "lt_sync", "lt_download", "lt_flush", "lt_compute_tracks",
              "lt_compute_tracks_begin", "lt_compute_tracks_end"):
        getattr(L, n).argtypes = [vp]
    for n in ("lt_count_images", "lt_num_nodes", "lt_num_valid_edges", "lt_num_all_tris", "lt_num_tracks",
              "lt_num_track_members"):
        getattr(L, n).argtypes = [vp]
        getattr(L, n).restype = C.c_int64
    L.lt_count_lines.argtypes = [vp, C.c_int]
    L.lt_count_lines.restype = C.c_int64
    L.lt_get_best.argtypes = [vp, dp, dp, i32p, u8p]
    L.lt_get_num_tris.argtypes = [vp, i32p]
    L.lt_get_valid_edges.argtypes = [vp, i64p, i32p]
    L.lt_get_valid_flags.argtypes = [vp, u8p]
    L.lt_get_all_tris.argtypes = [vp, i64p, dp, dp, i32p]
    L.lt_get_tracks.argtypes = [vp, dp, i64p, i32p, i32p, i32p, dp, dp]
    L.lt_image_results_size.argtypes = [vp, C.c_int, i64p]
    L.lt_image_results_size.restype = C.c_int64
    L.lt_export_image_results.argtypes = [vp, C.c_int, i32p, i32p, dp, dp, i32p, i32p, i64p, i32p]
    L.lt_import_image_results.argtypes = [vp, C.c_int, C.c_int, i32p, dp, dp, i32p, i32p, i64p, i32p]
    L.lt_export_images_size.argtypes = [vp, C.c_int, i32p, i64p, i64p]
    L.lt_export_images_packed.argtypes = [vp, C.c_int, i32p, i32p, dp]
    L.lt_import_images_packed.argtypes = [vp, i32p, C.c_int64, dp, C.c_int64]
    L.lt_shard_node_bytes.argtypes = []
    L.lt_shard_count.argtypes = [vp, i64p]
    L.lt_shard_build.argtypes = [vp, C.c_int64]
    L.lt_shard_export.argtypes = [vp, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]
    L.lt_shard_import.argtypes = [vp, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p]
    L.lt_ts_from_ctx.argtypes = [vp]
    L.lt_ts_from_ctx.restype = vp
    L.lt_ts_create.argtypes = [C.c_int64, dp, u8p, i64p, i32p, i32p, i32p, dp, dp, dp]
    L.lt_ts_create.restype = vp
    L.lt_ts_destroy.argtypes = [vp]
    L.lt_ts_destroy.restype = None
    for n in ("lt_ts_num_tracks", "lt_ts_num_members"):
        getattr(L, n).argtypes = [vp]
        getattr(L, n).restype = C.c_int64
    L.lt_ts_get.argtypes = [vp, dp, u8p, i64p, i32p, i32p, i32p, dp, dp, dp]
    L.lt_ts_filter_by_reprojection.argtypes = [vp, vp, C.c_double, C.c_double, C.c_int]
    L.lt_ts_filter_by_sensitivity.argtypes = [vp, vp, C.c_double, C.c_int]
    L.lt_ts_filter_by_overlap.argtypes = [vp, vp, C.c_double, C.c_int]
    L.lt_ts_remerge_once.argtypes = [vp, vp, C.POINTER(LtConfig), C.c_int]
    L.lt_get_stats.argtypes = [vp, i64p]
    L.lt_get_timers.argtypes = [vp, dp]
    L.lt_get_timer_sums.argtypes = [vp, dp, C.POINTER(C.c_int64), C.c_int]
    L.lt_fn_get_normal_direction.argtypes = [vp, dp, dp, dp]
    L.lt_fn_get_direction_from_vp.argtypes = [vp, dp, dp, dp]
    L.lt_fn_triangulate_point.argtypes = [vp, dp, dp, dp, dp, dp, C.POINTER(C.c_int)]
    L.lt_fn_triangulate_line_with_direction.argtypes = [vp, dp, dp, dp, dp, dp, dp]
    L.lt_fn_triangulate_line_with_one_point.argtypes = [vp, dp, dp, dp, dp, dp, dp]
    L.lt_fn_compute_fundamental_matrix.argtypes = [vp, dp, dp, dp]
    L.lt_fn_compute_epipolar_IoU.argtypes = [vp, dp, dp, dp, dp, dp]
    L.lt_fn_aggregate_line3d_list.argtypes = [C.c_int, dp, dp, C.c_int, dp]
    L.lt_fn_pack_match_rows.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int]
    L.lt_fn_compressed_block_words.argtypes = [C.c_int64]
    L.lt_fn_compressed_block_words.restype = C.c_int64
    L.lt_fn_pack_match_rows_compressed.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int]
    L.lt_fn_triangulate_line.argtypes = [vp, dp, dp, dp, dp, C.c_int, dp]
    _lib = L
    return L


def config_from_dict(d=None):
    """``GlobalLineTriangulatorConfig(py::dict)`` semantics (internal/helpers.h:25-27 of the
    reference): keys that are present overwrite the C++ defaults, unknown keys are ignored."""
    cfg = LtConfig()
    load_library().lt_config_default(C.byref(cfg))
    d = d or {}
    for k in BASE_KEYS:
        if k in d and d[k] is not None:
            setattr(cfg, k, type(getattr(cfg, k))(d[k]))
    if "merging_strategy" in d:
        cfg.merging_strategy = MERGING.get(d["merging_strategy"], 99)
    for prefix, keys, sub in (("l2_", L2_KEYS, "linker2d_config"), ("l3_", L3_KEYS, "linker3d_config")):
        subd = d.get(sub) or {}
        for k in keys:
            if k in subd:
                cur = getattr(cfg, prefix + k)
                setattr(cfg, prefix + k, type(cur)(subd[k]))
    return cfg


def f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


def ptr(a, t):
    return a.ctypes.data_as(C.POINTER(t))


class Context:
    """Owns one ``lt_ctx``; thin, typed access to the C ABI on numpy arrays."""

    def __init__(self, cfg_dict=None, device=0, cfg_struct=None):
        self.L = load_library()
        self.cfg = cfg_struct if cfg_struct is not None else config_from_dict(cfg_dict)
        h = self.L.lt_create(C.byref(self.cfg), int(device))
        if not h:
            raise RuntimeError("limap_amd: lt_create failed -- no usable HIP device (this backend has no CPU fallback)")
        self.h = C.c_void_p(h)
        self.device = int(device)

    def close(self):
        if getattr(self, "h", None):
            self.L.lt_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def chk(self, rc):
        if rc != 0:
            msg = self.L.lt_last_error(self.h).decode(errors="replace")
            if rc == -2:
                raise IndexError(msg) if msg.startswith("unknown") else ValueError(msg)
            raise RuntimeError(msg)

    # --- setup ---
    def set_stream(self, stream_ptr):
        self.chk(self.L.lt_set_stream(self.h, C.c_void_p(stream_ptr)))

    def set_ranges(self, lo, hi):
        lo, hi = f64(lo).reshape(3), f64(hi).reshape(3)
        self.chk(self.L.lt_set_ranges(self.h, ptr(lo, C.c_double), ptr(hi, C.c_double)))

    def unset_ranges(self):
        self.chk(self.L.lt_unset_ranges(self.h))

    def init(self, img_ids, kvec, qvec, tvec, seg_off, segs):
        img_ids, kvec, qvec, tvec = i32(img_ids), f64(kvec), f64(qvec), f64(tvec)
        seg_off, segs = i64(seg_off), f64(segs)
        n = len(img_ids)
        assert kvec.shape == (n, 4) and qvec.shape == (n, 4) and tvec.shape == (n, 3) and len(seg_off) == n + 1
        assert segs.shape == (int(seg_off[-1]) - int(seg_off[0]), 4) or segs.size == 0
        self.chk(self.L.lt_init(self.h, n, ptr(img_ids, C.c_int32), ptr(kvec, C.c_double), ptr(qvec, C.c_double),
                                ptr(tvec, C.c_double), ptr(seg_off, C.c_int64), ptr(segs, C.c_double)))

    def init_device(self, img_ids, d_kvec, d_qvec, d_tvec, seg_off, d_segs):
        img_ids, seg_off = i32(img_ids), i64(seg_off)
        self.chk(self.L.lt_init_device(self.h, len(img_ids), ptr(img_ids, C.c_int32), C.c_void_p(d_kvec),
                                       C.c_void_p(d_qvec), C.c_void_p(d_tvec), ptr(seg_off, C.c_int64),
                                       C.c_void_p(d_segs)))

    def refresh_scene_device(self, d_kvec, d_qvec, d_tvec, d_segs):
        self.chk(self.L.lt_refresh_scene_device(self.h, C.c_void_p(d_kvec), C.c_void_p(d_qvec), C.c_void_p(d_tvec),
                                                C.c_void_p(d_segs)))

    def set_scene_chunks(self, img_begin, d_k, d_q, d_t, d_s):
        n = len(img_begin)
        ib = i32(img_begin)
        arr = lambda ps: (C.c_void_p * n)(*[C.c_void_p(int(p)) for p in ps])
        self.chk(self.L.lt_set_scene_chunks(self.h, n, ptr(ib, C.c_int32), arr(d_k), arr(d_q), arr(d_t), arr(d_s)))

    def refresh_scene_chunks(self):
        self.chk(self.L.lt_refresh_scene_chunks(self.h))

    def triangulate_image(self, img_id, nb_ids, m_off, m_pairs):
        nb_ids, m_off, m_pairs = i32(nb_ids), i64(m_off), i32(m_pairs)
        self.chk(self.L.lt_triangulate_image(self.h, int(img_id), len(nb_ids), ptr(nb_ids, C.c_int32),
                                             ptr(m_off, C.c_int64), ptr(m_pairs, C.c_int32)))

    def init_vp(self, vpresults):
        """vpresults: dict img_id -> (labels (M,) int, vps (V,3) float) -- the content of vplib.VPResult."""
        ids = sorted(int(k) for k in vpresults)
        lab_off, vp_off = np.zeros(len(ids) + 1, np.int64), np.zeros(len(ids) + 1, np.int64)
        labs, vps = [np.zeros(0, np.int32)], [np.zeros((0, 3))]
        for n, i in enumerate(ids):
            lab, v = vpresults[i]
            lab, v = i32(np.asarray(lab).reshape(-1)), f64(np.asarray(v, float).reshape(-1, 3))
            labs.append(lab); vps.append(v)
            lab_off[n + 1] = lab_off[n] + len(lab)
            vp_off[n + 1] = vp_off[n] + len(v)
        labs = i32(np.concatenate(labs)); vps = f64(np.concatenate(vps, 0))
        if labs.size == 0:
            labs = np.zeros(1, np.int32)
        if vps.size == 0:
            vps = np.zeros((1, 3))
        self.chk(self.L.lt_init_vp(self.h, len(ids), ptr(i32(ids), C.c_int32), ptr(lab_off, C.c_int64),
                                   ptr(labs, C.c_int32), ptr(vp_off, C.c_int64), ptr(vps, C.c_double)))

    def set_bipartites(self, flat):
        """flat: dict of the CSR arrays of lt_set_bipartites (see triangulation.flatten_bipartites)."""
        self.chk(self.L.lt_set_bipartites(self.h, len(flat["img_ids"]), ptr(i32(flat["img_ids"]), C.c_int32),
                                          ptr(i64(flat["pt_off"]), C.c_int64), ptr(i32(flat["pt_ids"]), C.c_int32),
                                          ptr(f64(flat["pt_xy"]), C.c_double), ptr(i32(flat["pt_p3d"]), C.c_int32),
                                          ptr(i64(flat["line_off"]), C.c_int64), ptr(i64(flat["lp_off"]), C.c_int64),
                                          ptr(i32(flat["lp_ptids"]), C.c_int32)))

    def set_sfm_points(self, ids, xyz):
        ids, xyz = i32(ids), f64(np.asarray(xyz, float).reshape(-1, 3))
        n = len(ids)
        if n == 0:
            ids, xyz = np.zeros(1, np.int32), np.zeros((1, 3))
        self.chk(self.L.lt_set_sfm_points(self.h, n, ptr(ids, C.c_int32), ptr(xyz, C.c_double)))

    @property
    def rows_fn_addr(self):
        """Address of lt_triangulate_image_rows (for the CPython marshalling helper)."""
        a = getattr(self, "_rows_fn_addr", None)
        if a is None:
            a = self._rows_fn_addr = C.cast(self.L.lt_triangulate_image_rows, C.c_void_p).value
        return a

    def triangulate_image_rows(self, img_id, nb_ids, arrays):
        """arrays[k]: C-contiguous int32 (K,2) rows of neighbour nb_ids[k] (kept alive for the call)."""
        n = len(arrays)
        nb = i32(nb_ids)
        ptrs = (C.c_void_p * max(n, 1))(*[a.ctypes.data for a in arrays])
        cnt = np.fromiter((a.shape[0] for a in arrays), np.int64, n) if n else np.zeros(1, np.int64)
        self.chk(self.L.lt_triangulate_image_rows(self.h, int(img_id), n, ptr(nb, C.c_int32), ptrs, ptr(cnt, C.c_int64)))

    def triangulate_all_rows(self, img_ids, nb_lists, arrays):
        """One call for the TriangulateImage loop: image img_ids[k] has the neighbours nb_lists[k] with the C-contiguous
        int32 (K,2) row arrays arrays[k] (same order).  lt_triangulate_all_rows."""
        ids = i32(img_ids)
        nb_off = np.zeros(len(ids) + 1, np.int64)
        nb_off[1:] = np.cumsum([len(n) for n in nb_lists]) if len(ids) else 0
        nb = i32([x for n in nb_lists for x in n]) if nb_off[-1] else np.zeros(1, np.int32)
        flat = [a for arrs in arrays for a in arrs]
        ptrs = (C.c_void_p * max(len(flat), 1))(*[a.ctypes.data for a in flat])
        cnt = np.fromiter((a.shape[0] for a in flat), np.int64, len(flat)) if flat else np.zeros(1, np.int64)
        self.chk(self.L.lt_triangulate_all_rows(self.h, len(ids), ptr(ids, C.c_int32), ptr(nb_off, C.c_int64), ptr(nb, C.c_int32),
                                                ptrs, ptr(cnt, C.c_int64)))

    def triangulate_image_exhaustive(self, img_id, nb_ids):
        nb_ids = i32(nb_ids)
        self.chk(self.L.lt_triangulate_image_exhaustive(self.h, int(img_id), len(nb_ids), ptr(nb_ids, C.c_int32)))

    def upload(self):
        self.chk(self.L.lt_upload(self.h))

    def run_device(self, wait=True):
        """wait=False: enqueue only (lt_run_device_async); a run still in flight from the previous such call is
        completed after the new one is enqueued and its error, if any, is raised here.  sync() completes the
        run in flight."""
        self.chk(self.L.lt_run_device(self.h) if wait else self.L.lt_run_device_async(self.h))

    def sync(self):
        self.chk(self.L.lt_sync(self.h))

    def download(self):
        self.chk(self.L.lt_download(self.h))

    def flush(self):
        self.chk(self.L.lt_flush(self.h))

    def compute_tracks(self):
        self.chk(self.L.lt_compute_tracks(self.h))

    def compute_tracks_begin(self):
        """Device half of ComputeLineTracks enqueued behind the resident run; the next run may be enqueued before
        compute_tracks_end() does the host half (include/limap_amd.h: lt_compute_tracks_begin)."""
        self.chk(self.L.lt_compute_tracks_begin(self.h))

    def compute_tracks_end(self):
        self.chk(self.L.lt_compute_tracks_end(self.h))

    # --- getters ---
    def num_nodes(self):
        return int(self.L.lt_num_nodes(self.h))

    def count_images(self):
        return int(self.L.lt_count_images(self.h))

    def count_lines(self, img_id):
        n = int(self.L.lt_count_lines(self.h, int(img_id)))
        if n < 0:
            raise IndexError(self.L.lt_last_error(self.h).decode())
        return n

    def get_best(self):
        n = self.num_nodes()
        line = np.zeros((n, 10)); score = np.zeros(n); src = np.zeros((n, 2), np.int32); has = np.zeros(n, np.uint8)
        self.chk(self.L.lt_get_best(self.h, ptr(line, C.c_double), ptr(score, C.c_double), ptr(src, C.c_int32),
                                    ptr(has, C.c_uint8)))
        return dict(line=line, score=score, src=src, has_best=has)

    def get_num_tris(self):
        out = np.zeros(self.num_nodes(), np.int32)
        self.chk(self.L.lt_get_num_tris(self.h, ptr(out, C.c_int32)))
        return out

    def get_valid_flags(self):
        out = np.zeros(self.num_nodes(), np.uint8)
        self.chk(self.L.lt_get_valid_flags(self.h, ptr(out, C.c_uint8)))
        return out.astype(bool)

    def get_valid_edges(self):
        ne = int(self.L.lt_num_valid_edges(self.h))
        if ne < 0:
            self.chk(-1)
        off = np.zeros(self.num_nodes() + 1, np.int64); edges = np.zeros((max(ne, 1), 2), np.int32)
        self.chk(self.L.lt_get_valid_edges(self.h, ptr(off, C.c_int64), ptr(edges, C.c_int32)))
        return off, edges[:ne]

    def get_all_tris(self):
        nt = int(self.L.lt_num_all_tris(self.h))
        if nt < 0:
            self.chk(-1)
        off = np.zeros(self.num_nodes() + 1, np.int64); line = np.zeros((max(nt, 1), 10)); score = np.zeros(max(nt, 1))
        src = np.zeros((max(nt, 1), 2), np.int32)
        self.chk(self.L.lt_get_all_tris(self.h, ptr(off, C.c_int64), ptr(line, C.c_double), ptr(score, C.c_double),
                                        ptr(src, C.c_int32)))
        return dict(off=off, line=line[:nt], score=score[:nt], src=src[:nt])

    def get_tracks(self):
        T = int(self.L.lt_num_tracks(self.h)); M = int(self.L.lt_num_track_members(self.h))
        line = np.zeros((max(T, 1), 7)); off = np.zeros(T + 1, np.int64)
        img = np.zeros(max(M, 1), np.int32); lid = np.zeros(max(M, 1), np.int32); nid = np.zeros(max(M, 1), np.int32)
        sc = np.zeros(max(M, 1)); l3d = np.zeros((max(M, 1), 10))
        self.chk(self.L.lt_get_tracks(self.h, ptr(line, C.c_double), ptr(off, C.c_int64), ptr(img, C.c_int32),
                                      ptr(lid, C.c_int32), ptr(nid, C.c_int32), ptr(sc, C.c_double),
                                      ptr(l3d, C.c_double)))
        return dict(line=line[:T], off=off, image_ids=img[:M], line_ids=lid[:M], node_ids=nid[:M], scores=sc[:M],
                    line3d=l3d[:M])

    def export_image_results(self, img_id):
        """Per-node results of one image this context triangulated (dict of numpy arrays)."""
        ne = C.c_int64(0)
        m = int(self.L.lt_image_results_size(self.h, int(img_id), C.byref(ne)))
        if m < 0:
            self.chk(-1)
        nb = np.zeros(255, np.int32); n_nb = C.c_int32(0)
        line = np.zeros((max(m, 1), 10)); score = np.zeros(max(m, 1)); src = np.zeros((max(m, 1), 2), np.int32)
        nt = np.zeros(max(m, 1), np.int32); eoff = np.zeros(m + 1, np.int64); edges = np.zeros((max(ne.value, 1), 2), np.int32)
        self.chk(self.L.lt_export_image_results(self.h, int(img_id), ptr(nb, C.c_int32), C.byref(n_nb), ptr(line, C.c_double),
                                                ptr(score, C.c_double), ptr(src, C.c_int32), ptr(nt, C.c_int32),
                                                ptr(eoff, C.c_int64), ptr(edges, C.c_int32)))
        return dict(img_id=int(img_id), nb_ids=nb[:n_nb.value].copy(), line=line[:m], score=score[:m], src=src[:m],
                    n_tris=nt[:m], edge_off=eoff, edges=edges[:ne.value])

    def import_image_results(self, r):
        nb = i32(r["nb_ids"]); line = f64(r["line"]).reshape(-1, 10); score = f64(r["score"]); src = i32(r["src"]).reshape(-1, 2)
        nt = i32(r["n_tris"]); eoff = i64(r["edge_off"]); edges = i32(r["edges"]).reshape(-1, 2)
        if edges.size == 0:
            edges = np.zeros((1, 2), np.int32)
        if line.size == 0:
            line = np.zeros((1, 10)); score = np.zeros(1); src = np.zeros((1, 2), np.int32); nt = np.zeros(1, np.int32)
        self.chk(self.L.lt_import_image_results(self.h, int(r["img_id"]), len(nb), ptr(nb, C.c_int32), ptr(line, C.c_double),
                                                ptr(score, C.c_double), ptr(src, C.c_int32), ptr(nt, C.c_int32),
                                                ptr(eoff, C.c_int64), ptr(edges, C.c_int32)))

    def export_images_packed(self, img_ids):
        """Per-node results of the given images as (int32 blob, float64 blob) -- dist.pack_image_results' layout."""
        ids = i32(img_ids)
        ni, nd = C.c_int64(0), C.c_int64(0)
        self.chk(self.L.lt_export_images_size(self.h, len(ids), ptr(ids, C.c_int32), C.byref(ni), C.byref(nd)))
        ints, dbls = np.empty(ni.value, np.int32), np.empty(max(nd.value, 1), np.float64)
        self.chk(self.L.lt_export_images_packed(self.h, len(ids), ptr(ids, C.c_int32), ptr(ints, C.c_int32), ptr(dbls, C.c_double)))
        return ints, dbls[:nd.value]

    def import_images_packed(self, ints, dbls):
        ints, dbls = i32(ints), f64(dbls)
        d = dbls if dbls.size else np.zeros(1)
        self.chk(self.L.lt_import_images_packed(self.h, ptr(ints, C.c_int32), len(ints), ptr(d, C.c_double), len(dbls)))

    # --- shards of a multi-GPU run, device to device (include/limap_amd.h: lt_shard_*) ---
    def shard_node_bytes(self):
        return int(self.L.lt_shard_node_bytes())

    def shard_count(self):
        n = C.c_int64(0)
        self.chk(self.L.lt_shard_count(self.h, C.byref(n)))
        return int(n.value)

    def shard_build(self, total_keys):
        self.chk(self.L.lt_shard_build(self.h, int(total_keys)))

    def shard_export(self, g_lo, g_hi, nodes_ptr, keys_ptr):
        self.chk(self.L.lt_shard_export(self.h, int(g_lo), int(g_hi), C.c_void_p(int(nodes_ptr)), C.c_void_p(int(keys_ptr))))

    def shard_import(self, g_lo, g_hi, nodes_ptr, n_keys, keys_ptr):
        self.chk(self.L.lt_shard_import(self.h, int(g_lo), int(g_hi), C.c_void_p(int(nodes_ptr)), int(n_keys),
                                        C.c_void_p(int(keys_ptr))))

    def stats(self):
        out = np.zeros(8, np.int64)
        self.chk(self.L.lt_get_stats(self.h, ptr(out, C.c_int64)))
        keys = ["connections", "candidates", "pairs", "valid_edges", "graph_nodes", "graph_edges", "tracks", "nodes"]
        return dict(zip(keys, out.tolist()))

    _TIMER_KEYS = ["run", "invariants", "sort", "gen", "compact", "score", "select", "gather", "upload", "download",
                   "tail", "pairs_eval", "buffer", "k_gates", "k_tri_rows", "k_score3", "survivors", "ex_slots", "ex_cap",
                   "score_fused", "line_slots", "score_two_kernels", "tail_device", "tail_unionfind"]

    def timers(self):
        out = np.zeros(24)
        self.chk(self.L.lt_get_timers(self.h, ptr(out, C.c_double)))
        return dict(zip(self._TIMER_KEYS, out.tolist()))

    def timer_sums(self, reset=False):
        """-> (dict of the timer slots summed over the runs since the last reset, number of runs)."""
        out = np.zeros(24)
        n = C.c_int64(0)
        self.chk(self.L.lt_get_timer_sums(self.h, ptr(out, C.c_double), C.byref(n), 1 if reset else 0))
        return dict(zip(self._TIMER_KEYS, out.tolist())), int(n.value)

    # --- free functions ---
    def fn_normal_direction(self, seg, cam):
        out = np.zeros(3)
        self.chk(self.L.lt_fn_get_normal_direction(self.h, ptr(f64(seg), C.c_double), ptr(f64(cam), C.c_double),
                                                   ptr(out, C.c_double)))
        return out

    def fn_direction_from_vp(self, vp, cam):
        out = np.zeros(3)
        self.chk(self.L.lt_fn_get_direction_from_vp(self.h, ptr(f64(vp), C.c_double), ptr(f64(cam), C.c_double),
                                                    ptr(out, C.c_double)))
        return out

    def fn_triangulate_point(self, p1, cam1, p2, cam2):
        out = np.zeros(3)
        ok = C.c_int(0)
        self.chk(self.L.lt_fn_triangulate_point(self.h, ptr(f64(p1), C.c_double), ptr(f64(cam1), C.c_double),
                                                ptr(f64(p2), C.c_double), ptr(f64(cam2), C.c_double),
                                                ptr(out, C.c_double), C.byref(ok)))
        return out, bool(ok.value)

    def fn_triangulate_line_with_direction(self, seg1, cam1, seg2, cam2, direction):
        out = np.zeros(10)
        self.chk(self.L.lt_fn_triangulate_line_with_direction(self.h, ptr(f64(seg1), C.c_double),
                                                              ptr(f64(cam1), C.c_double), ptr(f64(seg2), C.c_double),
                                                              ptr(f64(cam2), C.c_double), ptr(f64(direction), C.c_double),
                                                              ptr(out, C.c_double)))
        return out

    def fn_triangulate_line_with_one_point(self, seg1, cam1, seg2, cam2, point):
        out = np.zeros(10)
        self.chk(self.L.lt_fn_triangulate_line_with_one_point(self.h, ptr(f64(seg1), C.c_double),
                                                              ptr(f64(cam1), C.c_double), ptr(f64(seg2), C.c_double),
                                                              ptr(f64(cam2), C.c_double), ptr(f64(point), C.c_double),
                                                              ptr(out, C.c_double)))
        return out

    def fn_fundamental_matrix(self, cam1, cam2):
        out = np.zeros(9)
        self.chk(self.L.lt_fn_compute_fundamental_matrix(self.h, ptr(f64(cam1), C.c_double), ptr(f64(cam2), C.c_double),
                                                         ptr(out, C.c_double)))
        return out.reshape(3, 3)

    def fn_epipolar_iou(self, seg1, cam1, seg2, cam2):
        out = C.c_double(0)
        self.chk(self.L.lt_fn_compute_epipolar_IoU(self.h, ptr(f64(seg1), C.c_double), ptr(f64(cam1), C.c_double),
                                                   ptr(f64(seg2), C.c_double), ptr(f64(cam2), C.c_double), C.byref(out)))
        return out.value

    def fn_triangulate_line(self, seg1, cam1, seg2, cam2, by_endpoints=False):
        out = np.zeros(10)
        self.chk(self.L.lt_fn_triangulate_line(self.h, ptr(f64(seg1), C.c_double), ptr(f64(cam1), C.c_double),
                                               ptr(f64(seg2), C.c_double), ptr(f64(cam2), C.c_double),
                                               int(bool(by_endpoints)), ptr(out, C.c_double)))
        return out
