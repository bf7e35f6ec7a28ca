"""ctypes binding of the CPU ORACLE (oracle/lt_oracle.cpp).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package (limap_amd/) must never import this module.

Pinned against the reference's own sources compiled into oracle/_ref (oracle/ref.py, tests/test_oracle_vs_ref.py).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "build", "liblt_oracle.so")


def build(force=False):
    """Compile oracle/lt_oracle.cpp with g++ (see oracle/Makefile)."""
    src = os.path.join(_HERE, "lt_oracle.cpp")
    if (
        force
        or not os.path.exists(_LIB_PATH)
        or os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(src), os.path.getmtime(os.path.join(_HERE, "lt_oracle.h")),
                                             os.path.getmtime(os.path.join(_HERE, "onepoint_terms.inc")))
    ):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


class OraConfig(C.Structure):
    _fields_ = [
        ("debug_mode", C.c_int32),
        ("add_halfpix", C.c_int32),
        ("use_vp", C.c_int32),
        ("use_endpoints_triangulation", C.c_int32),
        ("disable_many_points_triangulation", C.c_int32),
        ("disable_one_point_triangulation", C.c_int32),
        ("disable_algebraic_triangulation", C.c_int32),
        ("disable_vp_triangulation", C.c_int32),
        ("min_length_2d", C.c_double),
        ("line_tri_angle_threshold", C.c_double),
        ("IoU_threshold", C.c_double),
        ("sensitivity_threshold", C.c_double),
        ("var2d", C.c_double),
        ("fullscore_th", C.c_double),
        ("max_valid_conns", C.c_int32),
        ("min_num_outer_edges", C.c_int32),
        ("merging_strategy", C.c_int32),
        ("num_outliers_aggregator", C.c_int32),
        ("l2_score_th", C.c_double),
        ("l2_th_angle", C.c_double),
        ("l2_th_overlap", C.c_double),
        ("l2_th_smartoverlap", C.c_double),
        ("l2_th_smartangle", C.c_double),
        ("l2_th_perp", C.c_double),
        ("l2_th_innerseg", C.c_double),
        ("l2_use_angle", C.c_int32),
        ("l2_use_overlap", C.c_int32),
        ("l2_use_smartangle", C.c_int32),
        ("l2_use_perp", C.c_int32),
        ("l2_use_innerseg", C.c_int32),
        ("_pad0", C.c_int32),
        ("l3_score_th", C.c_double),
        ("l3_th_angle", C.c_double),
        ("l3_th_overlap", C.c_double),
        ("l3_th_smartoverlap", C.c_double),
        ("l3_th_smartangle", C.c_double),
        ("l3_th_perp", C.c_double),
        ("l3_th_innerseg", C.c_double),
        ("l3_th_scaleinv", C.c_double),
        ("l3_use_angle", C.c_int32),
        ("l3_use_overlap", C.c_int32),
        ("l3_use_smartangle", C.c_int32),
        ("l3_use_perp", C.c_int32),
        ("l3_use_innerseg", C.c_int32),
        ("l3_use_scaleinv", C.c_int32),
    ]


_BASE_KEYS = [
    "debug_mode", "add_halfpix", "use_vp", "use_endpoints_triangulation",
    "disable_many_points_triangulation", "disable_one_point_triangulation",
    "disable_algebraic_triangulation", "disable_vp_triangulation", "min_length_2d",
    "line_tri_angle_threshold", "IoU_threshold", "sensitivity_threshold", "var2d",
    "fullscore_th", "max_valid_conns", "min_num_outer_edges", "num_outliers_aggregator",
]
_L2_KEYS = ["score_th", "th_angle", "th_overlap", "th_smartoverlap", "th_smartangle", "th_perp",
            "th_innerseg", "use_angle", "use_overlap", "use_smartangle", "use_perp", "use_innerseg"]
_L3_KEYS = _L2_KEYS[:7] + ["th_scaleinv"] + _L2_KEYS[7:] + ["use_scaleinv"]

_lib = None


def _prototype(L):
    """Result / argument types of the entry points (shared with oracle/ref.py, whose library exports the same
    functions under the prefix ref_)."""
    L.ora_create.restype = C.c_void_p
    L.ora_create.argtypes = [C.POINTER(OraConfig), C.c_int]
    L.ora_destroy.argtypes = [C.c_void_p]
    L.ora_last_error.restype = C.c_char_p
    L.ora_last_error.argtypes = [C.c_void_p]
    for name in ("ora_num_nodes", "ora_num_valid_edges", "ora_num_all_tris", "ora_num_tracks",
                 "ora_num_track_members"):
        getattr(L, name).restype = C.c_int64
        getattr(L, name).argtypes = [C.c_void_p]
    for name in ("ora_compute_epipolar_IoU", "ora_cam_projdepth", "ora_line3d_sensitivity",
                 "ora_line3d_uncertainty", "ora_linker2d_score", "ora_linker3d_score"):
        getattr(L, name).restype = C.c_double
    return L


def lib():
    global _lib
    if _lib is None:
        build()
        L = _prototype(C.CDLL(_LIB_PATH))
        _lib = L
        # tiny parallel regions (<= topk iterations each) on a 256-core host spend all their time in
        # fork/join; the checker runs with a bounded team.  bench.py sets the count it reports.
        L.ora_set_num_threads(min(os.cpu_count() or 1, 16))
    return _lib


def set_num_threads(n):
    lib().ora_set_num_threads(int(n))


def set_one_point_solver(generated):
    """One-point proposal: True (default) = the reference's generated solver, term by term (bit-identical to oracle/_ref);
    False = the restated optimisation problem the device code follows (equal to ~1e-6 relative)."""
    lib().ora_set_one_point_solver(1 if generated else 0)


def get_one_point_solver():
    return bool(lib().ora_get_one_point_solver())


def get_max_threads():
    return int(lib().ora_get_max_threads())


def config_from_dict(d=None):
    """Reference semantics (internal/helpers.h:25-27): missing keys keep the C++ defaults,
    unknown keys are ignored."""
    cfg = OraConfig()
    lib().ora_config_default(C.byref(cfg))
    d = d or {}
    for k in _BASE_KEYS:
        if k in d:
            setattr(cfg, k, type(getattr(cfg, k))(d[k]))
    if "merging_strategy" in d:
        ms = d["merging_strategy"]
        cfg.merging_strategy = {"greedy": 0, "exhaustive": 1, "avg": 2}.get(ms, 99)
    for prefix, keys, sub in (("l2_", _L2_KEYS, "linker2d_config"), ("l3_", _L3_KEYS, "linker3d_config")):
        for k in keys:
            if k in d.get(sub, {}):
                cur = getattr(cfg, prefix + k)
                setattr(cfg, prefix + k, type(cur)(d[sub][k]))
    return cfg


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


def cam11(kvec, qvec, tvec):
    return _f64(np.concatenate([np.asarray(kvec, float), np.asarray(qvec, float), np.asarray(tvec, float)]))


def flatten_bipartites(bpts):
    """dict img_id -> dict(point_ids, xy, point3D_ids, line_points) -> the flat CSR arrays of the C APIs."""
    ids = sorted(int(k) for k in bpts)
    pt_off, line_off, lp_off = [0], [0], [0]
    pt_ids, pt_xy, pt_p3d, lp = [], [], [], []
    for i in ids:
        b = bpts[i]
        pid = np.asarray(b["point_ids"], np.int64).reshape(-1)
        pt_ids.append(pid); pt_xy.append(np.asarray(b["xy"], float).reshape(-1, 2))
        pt_p3d.append(np.asarray(b["point3D_ids"], np.int64).reshape(-1))
        pt_off.append(pt_off[-1] + len(pid))
        for pts in b["line_points"]:
            lp.append(np.asarray(pts, np.int64).reshape(-1))
            lp_off.append(lp_off[-1] + len(lp[-1]))
        line_off.append(line_off[-1] + len(b["line_points"]))

    def cat(parts, dtype, shape_tail=()):
        if parts and sum(len(x) for x in parts):
            return np.ascontiguousarray(np.concatenate(parts, 0), dtype=dtype)
        return np.zeros((1,) + shape_tail, dtype)
    return dict(img_ids=_i32(ids), pt_off=_i64(pt_off), pt_ids=cat(pt_ids, np.int32), pt_xy=cat(pt_xy, np.float64, (2,)),
                pt_p3d=cat(pt_p3d, np.int32), line_off=_i64(line_off), lp_off=_i64(lp_off), lp_ptids=cat(lp, np.int32))


class OracleTriangulator:
    """CPU restatement of limap.triangulation.GlobalLineTriangulator on flat arrays."""

    def __init__(self, cfg_dict=None, faithful=True):
        self.L = lib()
        self.cfg = config_from_dict(cfg_dict)
        self.ctx = C.c_void_p(self.L.ora_create(C.byref(self.cfg), int(faithful)))
        self.n_img = 0

    def __del__(self):
        if getattr(self, "ctx", None):
            self.L.ora_destroy(self.ctx)
            self.ctx = None

    def _chk(self, rc):
        if rc != 0:
            raise RuntimeError(self.L.ora_last_error(self.ctx).decode())

    def SetRanges(self, ranges):
        lo, hi = _f64(ranges[0]), _f64(ranges[1])
        self._chk(self.L.ora_set_ranges(self.ctx, _p(lo, C.c_double), _p(hi, C.c_double)))

    def UnsetRanges(self):
        self.L.ora_unset_ranges(self.ctx)

    def Init(self, img_ids, kvec, qvec, tvec, seg_off, segs):
        img_ids, kvec, qvec, tvec = _i32(img_ids), _f64(kvec), _f64(qvec), _f64(tvec)
        seg_off, segs = _i64(seg_off), _f64(segs)
        self.n_img = len(img_ids)
        self.img_ids_sorted = np.sort(img_ids)
        self._chk(self.L.ora_init(self.ctx, len(img_ids), _p(img_ids, C.c_int32), _p(kvec, C.c_double),
                                  _p(qvec, C.c_double), _p(tvec, C.c_double), _p(seg_off, C.c_int64),
                                  _p(segs, C.c_double)))

    def InitVPResults(self, vpresults):
        """vpresults: dict img_id -> (labels (M,) int, vps (V,3) float) -- vplib.VPResult content."""
        ids = sorted(int(k) for k in vpresults)
        lab_off, vp_off = np.zeros(len(ids) + 1, np.int64), np.zeros(len(ids) + 1, np.int64)
        labs, vps = [], []
        for n, i in enumerate(ids):
            lab, vp = vpresults[i]
            lab, vp = _i32(lab).reshape(-1), _f64(vp).reshape(-1, 3)
            labs.append(lab); vps.append(vp)
            lab_off[n + 1] = lab_off[n] + len(lab)
            vp_off[n + 1] = vp_off[n] + len(vp)
        labs = _i32(np.concatenate(labs)) if labs else np.zeros(1, np.int32)
        vps = _f64(np.concatenate(vps, 0)) if vps else np.zeros((1, 3))
        if labs.size == 0:
            labs = np.zeros(1, np.int32)
        if vps.size == 0:
            vps = np.zeros((1, 3))
        self._chk(self.L.ora_init_vp(self.ctx, len(ids), _p(_i32(ids), C.c_int32), _p(lab_off, C.c_int64),
                                     _p(labs, C.c_int32), _p(vp_off, C.c_int64), _p(vps, C.c_double)))

    def SetBipartites2d(self, bpts):
        """bpts: dict img_id -> dict(point_ids (Np,), xy (Np,2), point3D_ids (Np,), line_points: list over the
        image's lines of lists of point ids) -- the content of structures.PL_Bipartite2d."""
        flat = flatten_bipartites(bpts)
        self._chk(self.L.ora_set_bipartites(self.ctx, len(flat["img_ids"]), _p(flat["img_ids"], C.c_int32),
                                            _p(flat["pt_off"], C.c_int64), _p(flat["pt_ids"], C.c_int32),
                                            _p(flat["pt_xy"], C.c_double), _p(flat["pt_p3d"], C.c_int32),
                                            _p(flat["line_off"], C.c_int64), _p(flat["lp_off"], C.c_int64),
                                            _p(flat["lp_ptids"], C.c_int32)))

    def SetSfMPoints(self, points):
        ids = _i32(sorted(int(k) for k in points))
        xyz = _f64(np.array([points[int(k)] for k in ids], float).reshape(-1, 3)) if len(ids) else np.zeros((1, 3))
        if len(ids) == 0:
            ids = np.zeros(1, np.int32)
            n = 0
        else:
            n = len(ids)
        self._chk(self.L.ora_set_sfm_points(self.ctx, C.c_int64(n), _p(ids, C.c_int32), _p(xyz, C.c_double)))

    def TriangulateImage(self, img_id, matches):
        """matches: dict ng_img_id -> (K,2) int array."""
        nb = _i32(list(matches.keys()))
        off = np.zeros(len(nb) + 1, np.int64)
        rows = []
        for k, key in enumerate(matches.keys()):
            m = np.asarray(matches[key]).reshape(-1, 2)
            rows.append(m)
            off[k + 1] = off[k] + len(m)
        pairs = _i32(np.concatenate(rows, 0) if rows else np.zeros((0, 2)))
        self._chk(self.L.ora_triangulate_image(self.ctx, int(img_id), len(nb), _p(nb, C.c_int32),
                                               _p(off, C.c_int64), _p(pairs, C.c_int32)))

    def TriangulateImageExhaustiveMatch(self, img_id, neighbors):
        nb = _i32(neighbors)
        self._chk(self.L.ora_triangulate_image_exhaustive(self.ctx, int(img_id), len(nb), _p(nb, C.c_int32)))

    def ComputeLineTracks(self):
        self._chk(self.L.ora_compute_tracks(self.ctx))
        return self.get_tracks()

    # ---- getters ----
    def num_nodes(self):
        return int(self.L.ora_num_nodes(self.ctx))

    def get_num_tris(self):
        out = np.zeros(self.num_nodes(), np.int32)
        self.L.ora_get_num_tris(self.ctx, _p(out, C.c_int32))
        return out

    def get_best(self):
        n = self.num_nodes()
        line = np.zeros((n, 10)); score = np.zeros(n); src = np.zeros((n, 2), np.int32)
        has = np.zeros(n, np.uint8)
        self.L.ora_get_best(self.ctx, _p(line, C.c_double), _p(score, C.c_double), _p(src, C.c_int32),
                            _p(has, C.c_uint8))
        return dict(line=line, score=score, src=src, has_best=has)

    def get_valid_edges(self):
        n = self.num_nodes()
        ne = int(self.L.ora_num_valid_edges(self.ctx))
        off = np.zeros(n + 1, np.int64); edges = np.zeros((max(ne, 1), 2), np.int32)
        self.L.ora_get_valid_edges(self.ctx, _p(off, C.c_int64), _p(edges, C.c_int32))
        return off, edges[:ne]

    def get_all_tris(self):
        n = self.num_nodes()
        nt = int(self.L.ora_num_all_tris(self.ctx))
        off = np.zeros(n + 1, np.int64); line = np.zeros((max(nt, 1), 10)); score = np.zeros(max(nt, 1))
        src = np.zeros((max(nt, 1), 2), np.int32)
        self.L.ora_get_all_tris(self.ctx, _p(off, C.c_int64), _p(line, C.c_double), _p(score, C.c_double),
                                _p(src, C.c_int32))
        return dict(off=off, line=line[:nt], score=score[:nt], src=src[:nt])

    def get_tracks(self):
        T = int(self.L.ora_num_tracks(self.ctx)); M = int(self.L.ora_num_track_members(self.ctx))
        line = np.zeros((max(T, 1), 7)); off = np.zeros(T + 1, np.int64)
        img = np.zeros(max(M, 1), np.int32); lid = np.zeros(max(M, 1), np.int32)
        nid = np.zeros(max(M, 1), np.int32); sc = np.zeros(max(M, 1)); l3d = np.zeros((max(M, 1), 10))
        self.L.ora_get_tracks(self.ctx, _p(line, C.c_double), _p(off, C.c_int64), _p(img, C.c_int32),
                              _p(lid, C.c_int32), _p(nid, C.c_int32), _p(sc, C.c_double), _p(l3d, C.c_double))
        return dict(line=line[:T], off=off, image_ids=img[:M], line_ids=lid[:M], node_ids=nid[:M],
                    scores=sc[:M], line3d=l3d[:M])

    def stats(self):
        out = np.zeros(8, np.int64)
        self.L.ora_get_stats(self.ctx, _p(out, C.c_int64))
        keys = ["connections", "candidates", "pairs", "valid_edges", "graph_nodes", "graph_edges", "tracks"]
        return dict(zip(keys, out.tolist()))

    def timers(self):
        out = np.zeros(4)
        self.L.ora_get_timers(self.ctx, _p(out, C.c_double))
        return dict(gen=out[0], score=out[1], tail=out[2])


# ---- free functions --------------------------------------------------------------------------
def _d(a):
    return _p(a, C.c_double)


def get_normal_direction(seg, cam):
    out = np.zeros(3); lib().ora_get_normal_direction(_d(_f64(seg)), _d(_f64(cam)), _d(out)); return out


def compute_essential_matrix(cam1, cam2):
    out = np.zeros(9); lib().ora_compute_essential_matrix(_d(_f64(cam1)), _d(_f64(cam2)), _d(out))
    return out.reshape(3, 3)


def compute_fundamental_matrix(cam1, cam2):
    out = np.zeros(9); lib().ora_compute_fundamental_matrix(_d(_f64(cam1)), _d(_f64(cam2)), _d(out))
    return out.reshape(3, 3)


def compute_epipolar_IoU(seg1, cam1, seg2, cam2):
    return float(lib().ora_compute_epipolar_IoU(_d(_f64(seg1)), _d(_f64(cam1)), _d(_f64(seg2)), _d(_f64(cam2))))


def triangulate_point(p1, cam1, p2, cam2):
    out = np.zeros(3)
    ok = lib().ora_triangulate_point(_d(_f64(p1)), _d(_f64(cam1)), _d(_f64(p2)), _d(_f64(cam2)), _d(out))
    return out, bool(ok)


def get_direction_from_vp(vp, cam):
    out = np.zeros(3); lib().ora_get_direction_from_vp(_d(_f64(vp)), _d(_f64(cam)), _d(out)); return out


def triangulate_line_with_direction(seg1, cam1, seg2, cam2, direction):
    out = np.zeros(10)
    lib().ora_triangulate_line_with_direction(_d(_f64(seg1)), _d(_f64(cam1)), _d(_f64(seg2)), _d(_f64(cam2)),
                                              _d(_f64(direction)), _d(out))
    return out


def triangulate_line_with_one_point(seg1, cam1, seg2, cam2, point):
    out = np.zeros(10)
    lib().ora_triangulate_line_with_one_point(_d(_f64(seg1)), _d(_f64(cam1)), _d(_f64(seg2)), _d(_f64(cam2)),
                                              _d(_f64(point)), _d(out))
    return out


def triangulate_line(seg1, cam1, seg2, cam2):
    out = np.zeros(10); lib().ora_triangulate_line(_d(_f64(seg1)), _d(_f64(cam1)), _d(_f64(seg2)), _d(_f64(cam2)), _d(out))
    return out


def triangulate_line_by_endpoints(seg1, cam1, seg2, cam2):
    out = np.zeros(10)
    lib().ora_triangulate_line_by_endpoints(_d(_f64(seg1)), _d(_f64(cam1)), _d(_f64(seg2)), _d(_f64(cam2)), _d(out))
    return out


def cam_project(cam, p):
    out = np.zeros(2); lib().ora_cam_project(_d(_f64(cam)), _d(_f64(p)), _d(out)); return out


def cam_ray_direction(cam, p2d):
    out = np.zeros(3); lib().ora_cam_ray_direction(_d(_f64(cam)), _d(_f64(p2d)), _d(out)); return out


def cam_projdepth(cam, p):
    return float(lib().ora_cam_projdepth(_d(_f64(cam)), _d(_f64(p))))


def cam_R(cam):
    out = np.zeros(9); lib().ora_cam_R(_d(_f64(cam)), _d(out)); return out.reshape(3, 3)


def cam_center(cam):
    out = np.zeros(3); lib().ora_cam_center(_d(_f64(cam)), _d(out)); return out


def line3d_sensitivity(line10, cam):
    return float(lib().ora_line3d_sensitivity(_d(_f64(line10)), _d(_f64(cam))))


def line3d_uncertainty(line10, cam, var2d):
    lib().ora_line3d_uncertainty.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_double]
    return float(lib().ora_line3d_uncertainty(_d(_f64(line10)), _d(_f64(cam)), float(var2d)))


def linker2d_score(cfg_dict, seg1, seg2):
    cfg = config_from_dict(cfg_dict)
    return float(lib().ora_linker2d_score(C.byref(cfg), _d(_f64(seg1)), _d(_f64(seg2))))


def linker3d_score(cfg_dict, mode3d, line1, line2):
    cfg = config_from_dict(cfg_dict)
    return float(lib().ora_linker3d_score(C.byref(cfg), int(mode3d), _d(_f64(line1)), _d(_f64(line2))))


def track_labels_greedy(node_img, edge_sim, edge_nodes):
    node_img = _i32(node_img); edge_sim = _f64(edge_sim); edge_nodes = _i32(edge_nodes).reshape(-1, 2)
    out = np.zeros(len(node_img), np.int32)
    lib().ora_track_labels_greedy(len(node_img), _p(node_img, C.c_int32), C.c_int64(len(edge_sim)), _d(edge_sim),
                                  _p(edge_nodes, C.c_int32), _p(out, C.c_int32))
    return out


def aggregate_line3d_list(lines10, scores, num_outliers=2):
    lines10 = _f64(lines10).reshape(-1, 10); scores = _f64(scores)
    out = np.zeros(7)
    lib().ora_aggregate_line3d_list(len(scores), _d(lines10), _d(scores), int(num_outliers), _d(out))
    return out


class OracleTrackSet:
    """Copy of an OracleTriangulator's tracks for the post-triangulation steps
    (limap.merging.filter_tracks_by_reprojection / remerge / filter_tracks_by_sensitivity /
    filter_tracks_by_overlap; runners/line_triangulation.py:171-200)."""

    def __init__(self, tri):
        self.L = lib()
        self.tri = tri
        self.L.ora_ts_from_ctx.restype = C.c_void_p
        self.L.ora_ts_num_tracks.restype = C.c_int64
        self.L.ora_ts_num_members.restype = C.c_int64
        self.L.ora_ts_filter_by_reprojection.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_int]
        self.L.ora_ts_filter_by_sensitivity.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_int]
        self.L.ora_ts_filter_by_overlap.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_int]
        self.L.ora_ts_remerge_once.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(OraConfig), C.c_int]
        self.h = C.c_void_p(self.L.ora_ts_from_ctx(tri.ctx))

    def __del__(self):
        if getattr(self, "h", None):
            self.L.ora_ts_destroy(self.h)
            self.h = None

    def num_tracks(self):
        return int(self.L.ora_ts_num_tracks(self.h))

    def filter_by_reprojection(self, th_angular2d, th_perp2d, num_outliers=2):
        self.tri._chk(self.L.ora_ts_filter_by_reprojection(self.tri.ctx, self.h, th_angular2d, th_perp2d, num_outliers))

    def filter_by_sensitivity(self, th_angular3d, min_supports):
        self.tri._chk(self.L.ora_ts_filter_by_sensitivity(self.tri.ctx, self.h, th_angular3d, min_supports))

    def filter_by_overlap(self, th_overlap, min_supports):
        self.tri._chk(self.L.ora_ts_filter_by_overlap(self.tri.ctx, self.h, th_overlap, min_supports))

    def remerge(self, linker3d_dict, num_outliers=2):
        """merging.remerge (merging/merging.py:24-42): repeat until the track count stops changing."""
        cfg = config_from_dict({"linker3d_config": dict(linker3d_dict)})
        if self.num_tracks() == 0:
            return
        n = self.num_tracks()
        while True:
            self.tri._chk(self.L.ora_ts_remerge_once(self.tri.ctx, self.h, C.byref(cfg), num_outliers))
            n_new = self.num_tracks()
            if n_new == n:
                break
            n = n_new

    def get(self):
        T = self.num_tracks(); M = int(self.L.ora_ts_num_members(self.h))
        line = np.zeros((max(T, 1), 7)); active = np.zeros(max(T, 1), np.uint8); off = np.zeros(T + 1, np.int64)
        img = np.zeros(max(M, 1), np.int32); lid = np.zeros(max(M, 1), np.int32); nid = np.zeros(max(M, 1), np.int32)
        sc = np.zeros(max(M, 1)); l2 = np.zeros((max(M, 1), 4)); l3 = np.zeros((max(M, 1), 10))
        self.L.ora_ts_get(self.h, _p(line, C.c_double), _p(active, C.c_uint8), _p(off, C.c_int64), _p(img, C.c_int32),
                          _p(lid, C.c_int32), _p(nid, C.c_int32), _p(sc, C.c_double), _p(l2, C.c_double), _p(l3, C.c_double))
        return dict(line=line[:T], active=active[:T], off=off, image_ids=img[:M], line_ids=lid[:M], node_ids=nid[:M],
                    scores=sc[:M], line2d=l2[:M], line3d=l3[:M])
