"""Host-side mirror of the `limap.merging` functions that follow ComputeLineTracks inside
`limap.runners.line_triangulation` (runners/line_triangulation.py:171-200; python wrappers
merging/merging.py:24-100, C++ merging/merging_utils.cc:27-155 and merging/merging.cc:513-644):

    filter_tracks_by_reprojection, remerge, filter_tracks_by_sensitivity, filter_tracks_by_overlap

Same names and argument meaning.  `TrackSet` is the efficient form: it stays bound to the
triangulator's context (cameras already resident) and keeps the tracks in the native container
between steps; the module-level functions take and return LineTrack lists like the reference.
"""
import ctypes as C

import numpy as np

from . import _capi
from .base import Line2d, Line3d, LineTrack


def _linker_cfg(linker3d):
    """dict (cfg["triangulation"]["remerging"]["linker3d"]) or an object exposing the fields."""
    if isinstance(linker3d, dict):
        d = dict(linker3d)
    else:
        conf = getattr(linker3d, "config", linker3d)
        d = {k: getattr(conf, k) for k in _capi.L3_KEYS if hasattr(conf, k)}
    return _capi.config_from_dict({"linker3d_config": d})


class TrackSet:
    def __init__(self, ctx, handle):
        self.ctx = ctx
        self.L = ctx.L
        self.h = C.c_void_p(handle)

    @classmethod
    def from_triangulator(cls, tri):
        """Tracks of a GlobalLineTriangulator after ComputeLineTracks()."""
        ctx = tri.context()
        return cls(ctx, ctx.L.lt_ts_from_ctx(ctx.h))

    @classmethod
    def from_tracks(cls, ctx, tracks):
        T = len(tracks)
        off = np.zeros(T + 1, np.int64)
        off[1:] = np.cumsum([len(t.image_id_list) for t in tracks])
        M = int(off[-1])
        line7 = np.zeros((max(T, 1), 7)); active = np.ones(max(T, 1), np.uint8)
        img = np.zeros(max(M, 1), np.int32); lid = np.zeros(max(M, 1), np.int32); nid = np.zeros(max(M, 1), np.int32)
        sc = np.zeros(max(M, 1)); l2 = np.zeros((max(M, 1), 4)); l3 = np.zeros((max(M, 1), 10))
        for n, t in enumerate(tracks):
            line7[n, :3], line7[n, 3:6], line7[n, 6] = t.line.start, t.line.end, getattr(t.line, "uncertainty", -1.0)
            active[n] = 1 if getattr(t, "active", True) else 0
            a, b = int(off[n]), int(off[n + 1])
            img[a:b], lid[a:b] = t.image_id_list, t.line_id_list
            if len(t.node_id_list) == b - a:
                nid[a:b] = t.node_id_list
            if len(t.score_list) == b - a:
                sc[a:b] = t.score_list
            for k in range(b - a):
                l2[a + k, :2], l2[a + k, 2:] = t.line2d_list[k].start, t.line2d_list[k].end
                if k < len(t.line3d_list):
                    l = t.line3d_list[k]
                    l3[a + k, :3], l3[a + k, 3:6] = l.start, l.end
                    l3[a + k, 6:8] = getattr(l, "depths", (-1.0, -1.0))
                    l3[a + k, 8], l3[a + k, 9] = getattr(l, "uncertainty", -1.0), getattr(l, "score", -1.0)
        p = _capi.ptr
        h = ctx.L.lt_ts_create(T, p(line7, C.c_double), p(active, C.c_uint8), p(off, C.c_int64), p(img, C.c_int32),
                               p(lid, C.c_int32), p(nid, C.c_int32), p(sc, C.c_double), p(l2, C.c_double),
                               p(l3, C.c_double))
        return cls(ctx, h)

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.L.lt_ts_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def __len__(self):
        return int(self.L.lt_ts_num_tracks(self.h))

    def filter_by_reprojection(self, th_angular2d, th_perp2d, num_outliers=2):
        self.ctx.chk(self.L.lt_ts_filter_by_reprojection(self.ctx.h, self.h, float(th_angular2d), float(th_perp2d),
                                                         int(num_outliers)))
        return self

    def filter_by_sensitivity(self, th_angular3d, min_num_supports):
        self.ctx.chk(self.L.lt_ts_filter_by_sensitivity(self.ctx.h, self.h, float(th_angular3d), int(min_num_supports)))
        return self

    def filter_by_overlap(self, th_overlap, min_num_supports):
        self.ctx.chk(self.L.lt_ts_filter_by_overlap(self.ctx.h, self.h, float(th_overlap), int(min_num_supports)))
        return self

    def remerge(self, linker3d, num_outliers=2):
        """merging.remerge (merging/merging.py:24-42): repeat until the number of tracks is stable."""
        cfg = _linker_cfg(linker3d)
        n = len(self)
        if n == 0:
            return self
        while True:
            self.ctx.chk(self.L.lt_ts_remerge_once(self.ctx.h, self.h, C.byref(cfg), int(num_outliers)))
            n_new = len(self)
            if n_new == n:
                break
            n = n_new
        return self

    def arrays(self):
        T = len(self); M = int(self.L.lt_ts_num_members(self.h))
        line = np.zeros((max(T, 1), 7)); active = np.zeros(max(T, 1), np.uint8); off = np.zeros(T + 1, np.int64)
        img = np.zeros(max(M, 1), np.int32); lid = np.zeros(max(M, 1), np.int32); nid = np.zeros(max(M, 1), np.int32)
        sc = np.zeros(max(M, 1)); l2 = np.zeros((max(M, 1), 4)); l3 = np.zeros((max(M, 1), 10))
        p = _capi.ptr
        self.ctx.chk(self.L.lt_ts_get(self.h, p(line, C.c_double), p(active, C.c_uint8), p(off, C.c_int64),
                                      p(img, C.c_int32), p(lid, C.c_int32), p(nid, C.c_int32), p(sc, C.c_double),
                                      p(l2, C.c_double), p(l3, C.c_double)))
        return dict(line=line[:T], active=active[:T], off=off, image_ids=img[:M], line_ids=lid[:M], node_ids=nid[:M],
                    scores=sc[:M], line2d=l2[:M], line3d=l3[:M])

    def tracks(self):
        a = self.arrays()
        out = []
        for n in range(len(a["off"]) - 1):
            sl = slice(int(a["off"][n]), int(a["off"][n + 1]))
            t = LineTrack()
            t.line = Line3d(a["line"][n, :3], a["line"][n, 3:6], -1.0, -1.0, -1.0, a["line"][n, 6])
            t.image_id_list = a["image_ids"][sl].tolist(); t.line_id_list = a["line_ids"][sl].tolist()
            t.node_id_list = a["node_ids"][sl].tolist(); t.score_list = a["scores"][sl].tolist()
            t.line2d_list = [Line2d(s[:2], s[2:]) for s in a["line2d"][sl]]
            t.line3d_list = [Line3d.from10(s) for s in a["line3d"][sl]]
            t.active = bool(a["active"][n])
            out.append(t)
        return out


# ---- module-level functions with the reference's signatures ------------------------------------
def _ctx_for(imagecols):
    from .triangulation import _view_arrays
    ids = [int(i) for i in imagecols.get_img_ids()]
    k = np.zeros((len(ids), 4)); q = np.zeros((len(ids), 4)); t = np.zeros((len(ids), 3))
    for n, i in enumerate(ids):
        k[n], q[n], t[n] = _view_arrays(imagecols.camview(i))
    ctx = _capi.Context()
    ctx.init(ids, k, q, t, np.zeros(len(ids) + 1, np.int64), np.zeros((0, 4)))
    return ctx


def filter_tracks_by_reprojection(linetracks, imagecols, th_angular2d, th_perp2d, num_outliers=2):
    ts = TrackSet.from_tracks(_ctx_for(imagecols), linetracks)
    return ts.filter_by_reprojection(th_angular2d, th_perp2d, num_outliers).tracks()


def remerge(linker3d, linetracks, num_outliers=2):
    if len(linetracks) == 0:
        return linetracks
    ctx = _capi.Context()
    ctx.init([0], np.array([[1.0, 1, 0, 0]]), np.array([[1.0, 0, 0, 0]]), np.zeros((1, 3)), np.zeros(2, np.int64),
             np.zeros((0, 4)))
    return TrackSet.from_tracks(ctx, linetracks).remerge(linker3d, num_outliers).tracks()


def filter_tracks_by_sensitivity(linetracks, imagecols, th_angular3d, min_num_supports):
    ts = TrackSet.from_tracks(_ctx_for(imagecols), linetracks)
    return ts.filter_by_sensitivity(th_angular3d, min_num_supports).tracks()


def filter_tracks_by_overlap(linetracks, imagecols, th_overlap, min_num_supports):
    ts = TrackSet.from_tracks(_ctx_for(imagecols), linetracks)
    return ts.filter_by_overlap(th_overlap, min_num_supports).tracks()
