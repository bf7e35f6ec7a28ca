"""Host-side mirror of `limap.triangulation` for the MI355X backend.

Same names, argument meaning and error behaviour as the reference's pybind surface
(src/limap/triangulation/bindings.cc:19-32,78-119 and the doc wrappers in
src/limap/triangulation/triangulation.py), implemented over the C ABI of include/limap_amd.h.
Drop-in: ``limap.triangulation.GlobalLineTriangulator = limap_amd.triangulation.GlobalLineTriangulator``
(see INTEGRATION.md); the only production caller is src/limap/runners/line_triangulation.py:102-168.

The reference runs the whole multi-view step inside every ``TriangulateImage`` call.  Here the
per-image calls only buffer their inputs; the batch is executed on the GPU the first time results
are requested (``ComputeLineTracks`` or any getter), which is not observable through the API.
"""
import numpy as np

from . import _capi
from .base import CameraView, ImageCollection, Line2d, Line3d, LineTrack

# the pybind surface of triangulation/bindings.cc:22-31,78-119 (the reference's package does `from _limap._triangulation
# import *`, limap/triangulation/__init__.py:1-2: every free function has to be in __all__)
__all__ = [
    "GlobalLineTriangulator", "GlobalLineTriangulatorConfig", "get_normal_direction", "get_direction_from_VP",
    "compute_essential_matrix", "compute_fundamental_matrix", "compute_epipolar_IoU", "triangulate_point",
    "triangulate_line_by_endpoints", "triangulate_line", "triangulate_line_with_one_point",
    "triangulate_line_with_direction",
]


def _bipartite_as_arrays(b, n_lines):
    """PL_Bipartite2d-like object or plain dict -> dict(point_ids, xy, point3D_ids, line_points)."""
    if isinstance(b, dict) and "point_ids" in b:
        return b
    d = b.as_dict() if hasattr(b, "as_dict") else b
    pts, nl2p = d["points_"], d["nl2p_"]
    ids = sorted(int(k) for k in pts)

    def fields(p):
        if isinstance(p, dict):
            return np.asarray(p["p"], float).reshape(2), int(p["point3D_id"])
        return np.asarray(p.p, float).reshape(2), int(p.point3D_id)
    xy = np.zeros((len(ids), 2)); p3d = np.zeros(len(ids), np.int64)
    for n, i in enumerate(ids):
        xy[n], p3d[n] = fields(pts[i])
    n_lines = max(n_lines, (max((int(k) for k in nl2p), default=-1) + 1))
    return dict(point_ids=np.array(ids, np.int64), xy=xy, point3D_ids=p3d,
                line_points=[sorted(int(x) for x in nl2p.get(l, ())) for l in range(n_lines)])


def flatten_bipartites(bpts):
    """dict img_id -> dict(point_ids, xy, point3D_ids, line_points) -> the CSR arrays of lt_set_bipartites."""
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

    def cat(parts, dtype, tail=()):
        if parts and sum(len(x) for x in parts):
            return np.ascontiguousarray(np.concatenate(parts, 0), dtype=dtype)
        return np.zeros((1,) + tail, dtype)
    return dict(img_ids=np.asarray(ids, np.int32), pt_off=np.asarray(pt_off, np.int64), pt_ids=cat(pt_ids, np.int32),
                pt_xy=cat(pt_xy, np.float64, (2,)), pt_p3d=cat(pt_p3d, np.int32), line_off=np.asarray(line_off, np.int64),
                lp_off=np.asarray(lp_off, np.int64), lp_ptids=cat(lp, np.int32))


class _SegStore:
    """img_id -> (M, 4) segments of a scene, resolved when a 2D line is first looked at.  Holds the caller's arrays only
    -- NOT the triangulator: the tracks a triangulator hands out point here, and a reference back to it would be a cycle
    (triangulator -> tracks -> store -> triangulator) that keeps the context and its device memory alive until Python's
    cyclic collector happens to run (measured: every scene of a loop then pays the first-use allocations again)."""

    __slots__ = ("_src", "_order", "_img_ids", "_table")

    def __init__(self, src=(), order=None, img_ids=()):
        self._src, self._order, self._img_ids, self._table = src, order, img_ids, None

    def table(self):
        if self._table is None:
            pos = range(len(self._img_ids)) if self._order is None else self._order
            self._table = {i: np.asarray(self._src[o], float).reshape(-1, 4) for i, o in zip(self._img_ids, pos)}
        return self._table

    def __getitem__(self, img_id):
        return self.table()[img_id]


class _LazyLineTrack(LineTrack):
    """A LineTrack over the arrays lt_get_tracks returned: every field is materialised on first access."""

    def __init__(self, t, n, segs):  # deliberately no LineTrack.__init__: the fields appear on demand
        d = self.__dict__
        d["_t"], d["_n"], d["_segs"], d["active"] = t, n, segs, True

    def _slice(self):
        off = self._t["off"]
        return slice(int(off[self._n]), int(off[self._n + 1]))

    _LAZY = ("line", "image_id_list", "line_id_list", "node_id_list", "score_list", "line2d_list", "line3d_list")

    def __getattr__(self, name):  # only reached while the field has not been built yet
        # copy / pickle probe attributes on instances whose __dict__ is still empty: never recurse through `_t`
        if name not in _LazyLineTrack._LAZY:
            raise AttributeError(name)
        t = self.__dict__.get("_t")
        if t is None:
            raise AttributeError(name)
        if name == "line":
            r = t["line"][self._n]
            v = Line3d(r[0:3], r[3:6], -1.0, -1.0, -1.0, r[6])
        elif name == "image_id_list":
            v = t["image_ids"][self._slice()].tolist()
        elif name == "line_id_list":
            v = t["line_ids"][self._slice()].tolist()
        elif name == "node_id_list":
            v = t["node_ids"][self._slice()].tolist()
        elif name == "score_list":
            v = t["scores"][self._slice()].tolist()
        elif name == "line2d_list":
            v = [Line2d(self._segs[i][l, 0:2], self._segs[i][l, 2:4])
                 for i, l in zip(self.image_id_list, self.line_id_list)]
        elif name == "line3d_list":
            v = [Line3d.from10(a) for a in t["line3d"][self._slice()]]  # full Line3d: depths, uncertainty, score
        else:
            raise AttributeError(name)
        self.__dict__[name] = v
        return v

    def count_lines(self):
        s = self._slice()
        return s.stop - s.start

    def materialise(self):
        """A plain LineTrack with every field built (what copy / pickle hand on; the reference's LineTrack is
        picklable through its dict form, linetrack.cc:31-74)."""
        out = LineTrack(self.line, self.image_id_list, self.line_id_list, self.line2d_list)
        out.node_id_list, out.line3d_list = list(self.node_id_list), list(self.line3d_list)
        out.score_list, out.active = list(self.score_list), self.active
        return out

    def __reduce__(self):
        return (_track_from_state, (self.materialise().__dict__,))


class _LazyTrackList(list):
    """The list ComputeLineTracks / GetTracks hand back: a `list` of LineTrack whose elements are built when they are first
    looked at (1 367 track objects of the bench scene cost 0.3 ms to create -- a tenth of the whole call sequence -- and a caller
    that goes on with the array form, merging.TrackSet, never touches them).  Unbuilt slots hold None internally; every way
    into the list that could see one goes through `_at` or builds everything first."""

    def __init__(self, t, segs, _raw=None):
        list.__init__(self, [None] * (len(t["off"]) - 1) if _raw is None else _raw)
        self._t, self._segs = t, segs

    def _at(self, i):
        v = list.__getitem__(self, i)
        if v is None:
            v = _LazyLineTrack(self._t, i, self._segs)
            list.__setitem__(self, i, v)
        return v

    def _all(self):
        # (elements appended / inserted by a caller are real objects; a slot is unbuilt only while it is None AND still at
        # its original index, which holds as long as nobody reorders the list before this runs -- every reordering method
        # below runs it first)
        for i in range(list.__len__(self)):
            if list.__getitem__(self, i) is None:
                list.__setitem__(self, i, _LazyLineTrack(self._t, i, self._segs))
        return self

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self._at(k) for k in range(*i.indices(list.__len__(self)))]
        n = list.__len__(self)
        k = i + n if i < 0 else i
        if not 0 <= k < n:
            raise IndexError("list index out of range")
        return self._at(k)

    def __iter__(self):
        for i in range(list.__len__(self)):
            yield self._at(i)

    def __reversed__(self):
        for i in range(list.__len__(self) - 1, -1, -1):
            yield self._at(i)

    def copy(self):  # another lazy list over the same arrays, sharing what is built (a shallow copy, as list.copy is)
        return _LazyTrackList(self._t, self._segs, _raw=list(list.__iter__(self)))

    __copy__ = copy

    def __reduce__(self):
        return (list, (list(self),))

    def _plain(name):  # the rest of list's interface: build everything, then list's own method
        def f(self, *a, **k):
            return getattr(list, name)(self._all(), *a, **k)
        f.__name__ = name
        return f

    for _n in ("__contains__", "__eq__", "__ne__", "__lt__", "__le__", "__gt__", "__ge__", "__add__", "__mul__", "__rmul__", "__iadd__",
               "__imul__", "__repr__", "__delitem__", "__setitem__", "index", "count", "sort", "reverse", "pop", "remove", "insert"):
        locals()[_n] = _plain(_n)
    del _n, _plain
    __hash__ = None


def _track_from_state(state):
    out = LineTrack()
    out.__dict__.update(state)
    return out


try:  # limap's own value types when limap is installed (resolved once: a failed import is not cached by Python)
    import limap.base as _limap_base
except Exception:  # pragma: no cover
    _limap_base = None


# (the shim links liblimap_amd.so, which needs a HIP runtime: torch's bundled copy has to be in the process FIRST, or a
# later `import torch` loads a second runtime that finds no GPU -- _capi._preload_torch_hip_runtime; round 5: the preload
# only ran in load_library(), i.e. after this import, and `limap_amd` before `torch` broke torch.cuda)
_capi._preload_torch_hip_runtime()
try:  # the pybind11 shim over the C ABI (limap_amd/csrc/lt_pybind.cpp): the per-image calls and ComputeLineTracks
    from . import _lt_pybind as _pb
except ImportError:  # pragma: no cover
    _pb = None
try:  # CPython helper (limap_amd/csrc/lt_pymarshal.c), used when the pybind11 module is not built
    from . import _lt_pymarshal as _fast
except ImportError:  # pragma: no cover
    _fast = None


class GlobalLineTriangulatorConfig:
    """Mirror of the pybind config class (bindings.cc:40-74): attribute access to every field the
    reference exposes; constructed empty or from the ``cfg["triangulation"]`` dict."""

    def __init__(self, cfg_dict=None):
        object.__setattr__(self, "_s", _capi.config_from_dict(cfg_dict))
        object.__setattr__(self, "merging_strategy_name", (cfg_dict or {}).get("merging_strategy", "greedy"))

    def __getattr__(self, name):
        s = object.__getattribute__(self, "_s")
        if name == "merging_strategy":
            return object.__getattribute__(self, "merging_strategy_name")
        if name in ("linker2d_config", "linker3d_config"):
            pre = "l2_" if name == "linker2d_config" else "l3_"
            return {k[3:]: getattr(s, k) for k, _ in s._fields_ if k.startswith(pre)}
        if hasattr(s, name):
            return getattr(s, name)
        raise AttributeError(name)

    def __setattr__(self, name, value):
        s = object.__getattribute__(self, "_s")
        if name == "merging_strategy":
            object.__setattr__(self, "merging_strategy_name", value)
            s.merging_strategy = _capi.MERGING.get(value, 99)
        elif name in ("linker2d_config", "linker3d_config"):
            pre = "l2_" if name == "linker2d_config" else "l3_"
            for k, v in dict(value).items():
                if hasattr(s, pre + k):
                    setattr(s, pre + k, type(getattr(s, pre + k))(v))
        elif hasattr(s, name):
            setattr(s, name, type(getattr(s, name))(value))
        else:
            raise AttributeError(name)


def _view_arrays(view):
    """(kvec4, qvec4, tvec3) of a limap CameraView or of limap_amd.base.CameraView."""
    if hasattr(view, "kvec"):
        return np.asarray(view.kvec, float), np.asarray(view.qvec, float), np.asarray(view.tvec, float)
    K = np.asarray(view.K(), float)
    pose = getattr(view, "pose", view)
    q = np.asarray(getattr(pose, "qvec"), float).reshape(4)
    t = np.asarray(getattr(pose, "tvec"), float).reshape(3)
    return np.array([K[0, 0], K[1, 1], K[0, 2], K[1, 2]]), q, t


def _segs_array(lines):
    """list[Line2d-like] or (M, >=4) array -> (M,4) float64 (GetLine2dVectorFromArray, linebase.cc:134-142)."""
    if isinstance(lines, np.ndarray):
        a = np.asarray(lines, float)
        if a.size == 0:
            return np.zeros((0, 4))
        if a.ndim != 2 or a.shape[1] < 4:
            raise ValueError("segments must have shape (M, >=4)")
        return np.ascontiguousarray(a[:, :4])
    out = np.zeros((len(lines), 4))
    for i, l in enumerate(lines):
        if hasattr(l, "start"):
            out[i, :2] = np.asarray(l.start, float)
            out[i, 2:] = np.asarray(l.end, float)
        else:
            out[i] = np.asarray(l, float).reshape(-1)[:4]
    return out


def _cam11(view):
    if isinstance(view, np.ndarray) or isinstance(view, (list, tuple)):
        a = np.asarray(view, float).reshape(-1)
        if a.size != 11:
            raise ValueError("camera must be kvec4|qvec4|tvec3")
        return a
    k, q, t = _view_arrays(view)
    return np.concatenate([k, q, t])


def _make_line3d(a10):
    if _limap_base is not None:  # hand limap's own type back when it is installed
        try:
            return _limap_base.Line3d(np.asarray(a10[0:3]), np.asarray(a10[3:6]), float(a10[9]), float(a10[6]),
                                      float(a10[7]), float(a10[8]))
        except Exception:
            pass
    return Line3d.from10(a10)


class GlobalLineTriangulator:
    """`limap.triangulation.GlobalLineTriangulator` on MI355X (bindings.cc:78-119)."""

    def __init__(self, cfg=None, device=0):
        if isinstance(cfg, GlobalLineTriangulatorConfig):
            self._ctx = _capi.Context(cfg_struct=cfg._s, device=device)
        else:
            self._ctx = _capi.Context(cfg_dict=dict(cfg) if cfg is not None else None, device=device)
        self._img_ids = []
        self._seg_store = _SegStore()
        self._seg_off = None
        self._tracks = []
        self._debug = bool(self._ctx.cfg.debug_mode)
        self._best_cache = self._all_cache = None
        # non-owning pybind11 view of the same lt_ctx: TriangulateImage / ComputeLineTracks go through it
        self._pbv = _pb.GlobalLineTriangulator(self._ctx.h.value) if _pb is not None else None

    # ---- interfaces (bindings.cc:78-95) ----
    def SetRanges(self, ranges):
        self._ctx.set_ranges(ranges[0], ranges[1])

    def UnsetRanges(self):
        self._ctx.unset_ranges()

    def Init(self, all_2d_segs, imagecols):
        """Init(dict[int -> list[Line2d] | ndarray(M,4+)], ImageCollection)."""
        if hasattr(imagecols, "IsUndistorted") and not imagecols.IsUndistorted():
            raise ValueError("Check failed: imagecols->IsUndistorted() == true")  # base_line_triangulator.cc:49
        if isinstance(imagecols, dict):
            imagecols = ImageCollection({int(i): (v if hasattr(v, "K") else CameraView(*v)) for i, v in imagecols.items()})
        ids = [int(i) for i in imagecols.get_img_ids()]
        k = np.zeros((len(ids), 4)); q = np.zeros((len(ids), 4)); t = np.zeros((len(ids), 3))
        seg_list = []
        for n, i in enumerate(ids):
            k[n], q[n], t[n] = _view_arrays(imagecols.camview(i))
            if i not in all_2d_segs:
                raise IndexError(f"map::at: no 2D segments for image {i}")  # all_2d_segs.at(img_id), :56
            seg_list.append(_segs_array(all_2d_segs[i]))
        self.InitArrays(ids, k, q, t, seg_list)

    def InitArrays(self, img_ids, kvec, qvec, tvec, segs_per_image):
        """Flat-array form of Init (what the C ABI takes)."""
        n = len(img_ids)
        seg_off = np.zeros(n + 1, np.int64)
        np.cumsum([len(s) for s in segs_per_image], out=seg_off[1:])
        segs = np.concatenate(segs_per_image, 0) if n else np.zeros((0, 4))
        self._ctx.init(img_ids, kvec, qvec, tvec, seg_off, segs.reshape(-1, 4))
        ids = np.asarray(img_ids).astype(np.int64, copy=False).reshape(-1)
        if n < 2 or bool((ids[1:] > ids[:-1]).all()):  # already in the native order (ascending id): the usual case
            self._img_ids = ids.tolist()
            self._seg_off = seg_off
            self._seg_store = _SegStore(segs_per_image, None, self._img_ids)
        else:
            order = np.argsort(ids, kind="stable")
            self._img_ids = ids[order].tolist()
            so = np.zeros(n + 1, np.int64)
            np.cumsum((seg_off[1:] - seg_off[:-1])[order], out=so[1:])
            self._seg_off = so
            self._seg_store = _SegStore(segs_per_image, order.tolist(), self._img_ids)
        self._idx = dict(zip(self._img_ids, range(n)))
        self._tracks = []
        self._best_cache = self._all_cache = None

    @property
    def _segs(self):
        """img_id -> (M, 4) array of the image's 2D segments, built when a getter first needs the 2D lines."""
        return self._seg_store.table()

    def InitVPResults(self, vpresults):
        """vpresults: dict img_id -> limap.vplib.VPResult (or anything with .labels / .vps, a dict with those
        keys, or a (labels, vps) pair) -- bindings.cc:89, base_line_triangulator.h:47-49."""
        flat = {}
        for img_id, r in dict(vpresults).items():
            if isinstance(r, dict):
                lab, vps = r["labels"], r["vps"]
            elif hasattr(r, "labels"):
                lab, vps = r.labels, r.vps
            else:
                lab, vps = r
            flat[int(img_id)] = (np.asarray(lab, dtype=np.int64).reshape(-1),
                                 np.asarray(vps, dtype=float).reshape(-1, 3))
        self._ctx.init_vp(flat)
        self._vpresults = dict(vpresults)  # handed back as given (base_line_triangulator.h:50-55)

    def GetVPResult(self, image_id):
        """The VPResult passed to InitVPResults for this image (std::map::at -> KeyError if absent)."""
        return getattr(self, "_vpresults", {})[image_id]

    def GetVPResults(self):
        return dict(getattr(self, "_vpresults", {}))

    def SetBipartites2d(self, all_bpt2ds):
        """all_bpt2ds: dict img_id -> limap.structures.PL_Bipartite2d (anything with ``as_dict()`` giving
        ``points_`` / ``nl2p_``), or a plain dict(point_ids, xy, point3D_ids, line_points) -- bindings.cc:90.
        Enables the many-points and one-point proposals (cfg ``disable_*_triangulation``)."""
        n_lines = {i: int(self._seg_off[self._idx[i] + 1] - self._seg_off[self._idx[i]]) for i in self._img_ids}
        self._ctx.set_bipartites(flatten_bipartites({int(k): _bipartite_as_arrays(v, n_lines.get(int(k), 0))
                                                     for k, v in dict(all_bpt2ds).items()}))

    def SetSfMPoints(self, points):
        """points: dict point3D_id -> xyz (bindings.cc:91)."""
        ids = sorted(int(k) for k in points)
        self._ctx.set_sfm_points(ids, np.array([np.asarray(points[k], float).reshape(3) for k in ids], float).reshape(-1, 3))

    def TriangulateImage(self, img_id, matches):
        """matches: dict[int -> ndarray(K,2) int] (the content of matches_{img_id}.npy)."""
        self._best_cache = self._all_cache = None
        if self._pbv is not None and type(matches) is dict:
            self._pbv.TriangulateImage(int(img_id), matches)  # buffer protocol, GIL released around the native call
            return
        if _fast is not None and type(matches) is dict:
            # C-contiguous int32 (K,2) arrays (what matchers write) go straight through the buffer protocol
            rc = _fast.triangulate_image_rows(self._ctx.rows_fn_addr, self._ctx.h.value, int(img_id), matches)
            if rc is not None:
                self._ctx.chk(rc)
                return
        nb, rows = [], []
        for key, m in matches.items():
            m = np.asarray(m)
            if m.size != 0 and (m.ndim != 2 or m.shape[1] != 2):
                raise ValueError("Check failed: match_info.cols() == 2")  # base_line_triangulator.cc:79
            if m.dtype != np.int32 or not m.flags.c_contiguous or m.ndim != 2:
                m = np.ascontiguousarray(m.reshape(-1, 2), dtype=np.int32)  # Eigen::MatrixXi caster: converted by copy
            nb.append(int(key))
            rows.append(m)
        self._ctx.triangulate_image_rows(img_id, nb, rows)

    def TriangulateAll(self, matches_by_image):
        """The caller's loop `for img_id in ...: TriangulateImage(img_id, matches[img_id])`
        (runners/line_triangulation.py:160-167) as ONE call: matches_by_image = {img_id: {ng_img_id: (K,2) int array}},
        processed in the dict's order.  Same results and errors as the loop; the rows of all images are validated and
        buffered in one pass (the per-image form pays ~21 us of fork/join per call).  No reference counterpart."""
        self._best_cache = self._all_cache = None
        if self._pbv is not None and type(matches_by_image) is dict and all(type(m) is dict for m in matches_by_image.values()):
            self._pbv.TriangulateAll(matches_by_image)
            return
        ids, nbs, arrs = [], [], []
        for img_id, matches in matches_by_image.items():
            nb, rows = [], []
            for key, m in matches.items():
                m = np.asarray(m)
                if m.size != 0 and (m.ndim != 2 or m.shape[1] != 2):
                    raise ValueError("Check failed: match_info.cols() == 2")  # base_line_triangulator.cc:79
                if m.dtype != np.int32 or not m.flags.c_contiguous or m.ndim != 2:
                    m = np.ascontiguousarray(m.reshape(-1, 2), dtype=np.int32)
                nb.append(int(key))
                rows.append(m)
            ids.append(int(img_id)); nbs.append(nb); arrs.append(rows)
        self._ctx.triangulate_all_rows(ids, nbs, arrs)

    def TriangulateImageExhaustiveMatch(self, img_id, neighbors):
        self._best_cache = self._all_cache = None
        self._ctx.triangulate_image_exhaustive(img_id, [int(x) for x in neighbors])

    def ComputeLineTracks(self):
        self._best_cache = self._all_cache = None  # the call may run a pending batch: the getters read afresh
        if self._pbv is not None:
            t = self._pbv.ComputeLineTracks()  # one call: run + tail + the track arrays
        else:
            self._ctx.compute_tracks()
            t = self._ctx.get_tracks()
        self._tracks = self._build_tracks(t)
        return self.GetTracks()

    def GetTracks(self):
        return self._tracks.copy() if isinstance(self._tracks, _LazyTrackList) else list(self._tracks)

    def CountImages(self):
        return self._ctx.count_images()

    def CountLines(self, img_id):
        return self._ctx.count_lines(img_id)

    def GetLinker(self):
        c = self._ctx.cfg
        return dict(linker2d={k[3:]: getattr(c, k) for k, _ in c._fields_ if k.startswith("l2_")},
                    linker3d={k[3:]: getattr(c, k) for k, _ in c._fields_ if k.startswith("l3_")})

    # ---- visualisation getters (bindings.cc:100-119) ----
    def _node(self, img_id, line_id):
        return int(self._seg_off[self._idx[int(img_id)]] + int(line_id))

    def _best(self):
        if self._best_cache is None:
            self._best_cache = self._ctx.get_best()
        return self._best_cache

    def _all_tris(self):
        if self._all_cache is None:  # one read-out of tris_ serves the per-node getters until the next batch
            self._all_cache = self._ctx.get_all_tris()
        return self._all_cache

    def GetAllBestTris(self):
        b = self._best()
        return [_make_line3d(b["line"][g]) for g in range(len(b["score"]))]

    def GetAllValidBestTris(self):
        """Best candidates of the nodes that survive filterNodeByNumOuterEdges (valid_flags_)."""
        b = self._best()
        flags = self._valid_flags()
        return [_make_line3d(b["line"][g]) for g in range(len(b["score"])) if flags[g]]

    def GetBestTrisImage(self, img_id):
        b = self._best()
        i = self._idx[int(img_id)]
        return [_make_line3d(b["line"][g]) for g in range(self._seg_off[i], self._seg_off[i + 1])]

    def GetBestTriNode(self, img_id, line_id):
        return _make_line3d(self._best()["line"][self._node(img_id, line_id)])

    def GetBestScoredTriNode(self, img_id, line_id):
        b = self._best()
        g = self._node(img_id, line_id)
        return (_make_line3d(b["line"][g]), float(b["score"][g]), (int(b["src"][g, 0]), int(b["src"][g, 1])))

    def CountAllTris(self):
        return int(self._ctx.get_num_tris().sum()) if self._debug else 0

    def GetScoredTrisNode(self, img_id, line_id):
        if not self._debug:  # tris_ is cleared after scoring unless debug_mode (global_line_triangulator.cc:156-159)
            return []
        a = self._all_tris()
        g = self._node(img_id, line_id)
        return [(_make_line3d(a["line"][t]), float(a["score"][t]), (int(a["src"][t, 0]), int(a["src"][t, 1])))
                for t in range(a["off"][g], a["off"][g + 1])]

    def GetValidScoredTrisNode(self, img_id, line_id):
        """valid_tris_: candidates with score >= fullscore_th among the max_valid_conns best, in
        descending (score, tri_id) order (global_line_triangulator.cc:124-142)."""
        tris = self.GetScoredTrisNode(img_id, line_id)
        order = sorted(range(len(tris)), key=lambda t: (tris[t][1], t), reverse=True)
        order = order[:int(self._ctx.cfg.max_valid_conns)]
        return [tris[t] for t in order if tris[t][1] >= self._ctx.cfg.fullscore_th]

    def GetValidScoredTrisNodeSet(self, img_id, line_id):
        best = {}
        for tri in self.GetValidScoredTrisNode(img_id, line_id):  # first strictly-greater per image wins
            k = tri[2][0]
            if k not in best or tri[1] > best[k][1]:
                best[k] = tri
        return [best[k] for k in sorted(best)]

    def CountAllValidTris(self):
        return int(len(self._ctx.get_valid_edges()[1])) if self._debug else 0

    def GetValidTrisNode(self, img_id, line_id):
        return [t[0] for t in self.GetValidScoredTrisNode(img_id, line_id)]

    def GetValidTrisNodeSet(self, img_id, line_id):
        return [t[0] for t in self.GetValidScoredTrisNodeSet(img_id, line_id)]

    def GetValidTrisImage(self, img_id):
        out = []
        for l in range(self.CountLines(img_id)):
            out += self.GetValidTrisNode(img_id, l)
        return out

    def GetAllValidTris(self):
        out = []
        for i in self._img_ids:
            out += self.GetValidTrisImage(i)
        return out

    def GetSurvivedLinesImage(self, image_id, n_visible_views):
        out = []
        for tr in self._tracks:  # global_line_triangulator.cc:543-558
            if tr.count_images() < n_visible_views:
                continue
            out += [l for i, l in zip(tr.image_id_list, tr.line_id_list) if i == image_id]
        return out

    # ---- extras of this backend ----
    def stats(self):
        return self._ctx.stats()

    def timers(self):
        return self._ctx.timers()

    def context(self):
        """The underlying C-ABI context.  Whoever drives it directly (import_image_results, run_device, ...) changes the
        results behind this object's back: the cached getter arrays are dropped here."""
        self._best_cache = self._all_cache = None
        return self._ctx

    # ---- helpers ----
    def _valid_flags(self):
        """valid_flags_ (filterNodeByNumOuterEdges, global_line_triangulator.cc:168-232).  The reference fills
        it inside run_clustering, i.e. by ComputeLineTracks(); before that its GetAllValidBestTris indexes an
        empty vector -- here that is a RuntimeError."""
        return self._ctx.get_valid_flags()

    def _build_tracks(self, t):
        # The reference hands back pybind wrappers of C++ LineTracks (no per-member Python objects until they are
        # looked at); building ~35 000 Line2d / Line3d objects eagerly here cost 20x the whole triangulation.
        segs = self._seg_store  # (not the triangulator: see _SegStore)
        if _limap_base is not None:  # limap's LineTrack when limap is installed (linetrack.cc:50-74 dict ctor)
            try:
                return [_limap_base.LineTrack(_LazyLineTrack(t, n, segs).as_dict()) for n in range(len(t["off"]) - 1)]
            except Exception:
                pass
        return _LazyTrackList(t, segs)


# ---- free functions (bindings.cc:22-31; doc wrappers triangulation.py:1-138 of the reference) ----
_fn_ctx = None


def _fctx():
    global _fn_ctx
    if _fn_ctx is None:
        _fn_ctx = _capi.Context()
    return _fn_ctx


def get_normal_direction(l2d, view):
    return _fctx().fn_normal_direction(_segs_array([l2d])[0], _cam11(view))


def get_direction_from_VP(vp, view):
    return _fctx().fn_direction_from_vp(np.asarray(vp, float).reshape(3), _cam11(view))


def triangulate_point(p1, view1, p2, view2):
    """-> (point (3,), ok) like the reference's std::pair<V3D, bool> (functions.cc:100-117)."""
    return _fctx().fn_triangulate_point(np.asarray(p1, float).reshape(2), _cam11(view1),
                                        np.asarray(p2, float).reshape(2), _cam11(view2))


def triangulate_line_with_direction(l1, view1, l2, view2, direction):
    return _make_line3d(_fctx().fn_triangulate_line_with_direction(
        _segs_array([l1])[0], _cam11(view1), _segs_array([l2])[0], _cam11(view2), np.asarray(direction, float).reshape(3)))


def triangulate_line_with_one_point(l1, view1, l2, view2, point):
    """functions.cc:325-383; the solver is a restatement of the reference's optimisation problem (agrees to
    rounding, see include/limap_amd.h)."""
    return _make_line3d(_fctx().fn_triangulate_line_with_one_point(
        _segs_array([l1])[0], _cam11(view1), _segs_array([l2])[0], _cam11(view2), np.asarray(point, float).reshape(3)))


def compute_fundamental_matrix(view1, view2):
    return _fctx().fn_fundamental_matrix(_cam11(view1), _cam11(view2))


def compute_essential_matrix(view1, view2):
    """E = K2^T F K1 is not how the reference computes it (functions.cc:44-67 builds E first); the
    device query returns F, and E is recovered here only for API completeness."""
    F = compute_fundamental_matrix(view1, view2)
    k1, k2 = _cam11(view1)[:4], _cam11(view2)[:4]
    K1 = np.array([[k1[0], 0, k1[2]], [0, k1[1], k1[3]], [0, 0, 1.0]])
    K2 = np.array([[k2[0], 0, k2[2]], [0, k2[1], k2[3]], [0, 0, 1.0]])
    return K2.T @ F @ K1


def compute_epipolar_IoU(l1, view1, l2, view2):
    return _fctx().fn_epipolar_iou(_segs_array([l1])[0], _cam11(view1), _segs_array([l2])[0], _cam11(view2))


def triangulate_line(l1, view1, l2, view2):
    return _make_line3d(_fctx().fn_triangulate_line(_segs_array([l1])[0], _cam11(view1), _segs_array([l2])[0],
                                                    _cam11(view2), False))


def triangulate_line_by_endpoints(l1, view1, l2, view2):
    return _make_line3d(_fctx().fn_triangulate_line(_segs_array([l1])[0], _cam11(view1), _segs_array([l2])[0],
                                                    _cam11(view2), True))
