"""-m gpu: the REAL caller, replayed.  The body of the reference's own `limap.runners.line_triangulation.line_triangulation`
(src/limap/runners/line_triangulation.py:18-205: cfg handling -> metainfos -> segments -> matches -> GlobalLineTriangulator
ctor / SetRanges / Init / TriangulateImage loop / ComputeLineTracks -> the four limap.merging calls -> outputs) is loaded
from /root/reference AS IT IS and executed with `limap.*` rebound the way INTEGRATION.md section 2 describes:

    limap.base, limap.merging, limap.triangulation, limap.util.io  ->  limap_amd.base / .merging / .triangulation / .io
    limap.runners (setup, compute_2d_segs, compute_matches)        ->  stand-ins that hand over a synthetic scene's
                                                                        segments and write its matches_*.npy files
    pycolmap, tqdm, limap.optimize / pointsfm / vplib / visualize   ->  inert stubs (refinement and visualisation off)

The tracks it returns are compared with the CPU oracle driven through the same sequence.  Those two tests need the
reference tree AND (the second one) a GPU -- the runner's source cannot travel to the GPU box, and the build container
has no GPU.  What does travel is a RECORD of the real caller: tests/golden/make_caller_trace.py runs the runner body in
the build container against a recording, oracle-backed `limap.triangulation` / `limap.merging` and writes every call
with its arguments, plus the tracks the caller got, to tests/golden/caller_trace.json;
test_recorded_caller_trace_on_this_backend replays that record call by call against the HIP backend on the GPU box.
(Its first run found a real drop-in defect: the LineTrack lists handed back to the caller carried (start, end) only for
their supporting 3D lines, and remerge, which re-aggregates from the supports' uncertainties, merged 96 tracks into 48
where the reference's objects give 96.)"""
import importlib.util
import os
import sys
import types

import numpy as np
import pytest

from limap_amd import synthetic as syn

RUNNER = "/root/reference/src/limap/runners/line_triangulation.py"
DEFAULT_YAML = "/root/reference/cfgs/triangulation/default.yaml"


def _load_runner(scene, matches_topk, triangulation=None, merging=None):
    from limap_amd import base, io as ltio
    if triangulation is None:
        from limap_amd import triangulation
    if merging is None:
        from limap_amd import merging

    calls = []
    log = types.SimpleNamespace(info=lambda *a, **k: None, warning=lambda *a, **k: None, error=lambda *a, **k: None)
    mods = {}

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        mods[name] = m
        return m
    mod("pycolmap", logging=log, Reconstruction=None)
    mod("tqdm", tqdm=lambda it, *a, **k: it)
    limap = mod("limap")
    limap.__path__ = []
    for name, real in (("base", base), ("merging", merging), ("triangulation", triangulation)):
        mods["limap." + name] = real
        setattr(limap, name, real)
    util = mod("limap.util")
    util.__path__ = []
    mods["limap.util.io"] = ltio
    util.io = ltio
    limap.util = util
    for name in ("optimize", "pointsfm", "vplib"):
        setattr(limap, name, mod("limap." + name))

    class _Vis:  # limap.visualize.Open3DTrackVisualizer as far as the runner touches it
        def __init__(self, tracks):
            self.tracks = tracks

        def report(self):
            calls.append(("report", len(self.tracks)))

        def get_lines_np(self, n_visible_views=4):
            return [t.line.as_array() for t in self.tracks if t.count_images() >= n_visible_views]
    limap.visualize = mod("limap.visualize", Open3DTrackVisualizer=_Vis)

    def setup(cfg):  # runners/functions.py: creates the output folders
        os.makedirs(cfg["dir_save"], exist_ok=True)
        return cfg

    def compute_2d_segs(cfg, imagecols, compute_descinfo=True):
        calls.append(("compute_2d_segs", compute_descinfo))
        return {int(i): scene.segs_of(k) for k, i in enumerate(scene.img_ids)}, None

    def compute_matches(cfg, descinfo_folder, image_ids, neighbors):
        folder = os.path.join(cfg["dir_save"], "matches")
        for i in image_ids:
            ltio.save_matches(folder, int(i), scene.matches_of(int(i), matches_topk))
        calls.append(("compute_matches", len(image_ids)))
        return folder
    limap.runners = mod("limap.runners", setup=setup, compute_2d_segs=compute_2d_segs, compute_matches=compute_matches)
    saved = {k: sys.modules.get(k) for k in mods}
    sys.modules.update(mods)
    try:
        spec = importlib.util.spec_from_file_location("_reference_runner_line_triangulation", RUNNER)
        runner = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(runner)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    return runner, calls


def _oracle_backed_modules(oracle, scene):
    """`limap.triangulation` / `limap.merging` stand-ins over the CPU oracle with the reference's call surface -- for the
    CPU form of the replay (no GPU in the build container): what is exercised there is everything ELSE the real caller
    touches of this package (limap_amd.base / limap_amd.io: configs, Line2d dicts, ImageCollection methods, LineLinker3d,
    the writers) in the caller's own order."""
    from limap_amd import base

    def tracks_from(arr):
        out = []
        for k in range(len(arr["off"]) - 1):
            a, b = int(arr["off"][k]), int(arr["off"][k + 1])
            tr = base.LineTrack()
            tr.line = base.Line3d(arr["line"][k, :3], arr["line"][k, 3:6])
            tr.image_id_list = arr["image_ids"][a:b].tolist(); tr.line_id_list = arr["line_ids"][a:b].tolist()
            tr.node_id_list = arr["node_ids"][a:b].tolist(); tr.score_list = arr["scores"][a:b].tolist()
            l2 = arr.get("line2d")
            tr.line2d_list = [base.Line2d(s[:2], s[2:]) for s in l2[a:b]] if l2 is not None else \
                [base.Line2d(scene.segs_of(int(np.searchsorted(scene.img_ids, i)))[l][:2],
                             scene.segs_of(int(np.searchsorted(scene.img_ids, i)))[l][2:]) for i, l in zip(tr.image_id_list, tr.line_id_list)]
            l3 = arr.get("line3d")
            tr.line3d_list = [base.Line3d(s[:3], s[3:6]) for s in l3[a:b]] if l3 is not None else []
            out.append(tr)
        return out
    state = {}

    class GlobalLineTriangulator:
        def __init__(self, cfg):
            self.O = oracle.OracleTriangulator(dict(cfg), faithful=False)
            state["O"] = self.O

        def SetRanges(self, ranges):
            self.O.SetRanges(ranges)

        def Init(self, all_2d_lines, imagecols):
            ids = imagecols.get_img_ids()
            assert ids == [int(i) for i in scene.img_ids] and all(len(all_2d_lines[i]) for i in ids)
            self.O.Init(scene.img_ids, scene.kvec, scene.qvec, scene.tvec, scene.seg_off, scene.segs)

        def TriangulateImage(self, img_id, matches):
            self.O.TriangulateImage(int(img_id), matches)

        def TriangulateImageExhaustiveMatch(self, img_id, neighbors):
            self.O.TriangulateImageExhaustiveMatch(int(img_id), neighbors)

        def ComputeLineTracks(self):
            self.O.ComputeLineTracks()
            state["ts"] = oracle.OracleTrackSet(self.O)
            return tracks_from(state["ts"].get())

    def _ret():
        return tracks_from(state["ts"].get())
    import types
    merging = types.ModuleType("limap.merging")
    merging.filter_tracks_by_reprojection = lambda tr, ic, a, p: (state["ts"].filter_by_reprojection(a, p), _ret())[1]
    merging.filter_tracks_by_sensitivity = lambda tr, ic, a, n: (state["ts"].filter_by_sensitivity(a, n), _ret())[1]
    merging.filter_tracks_by_overlap = lambda tr, ic, o, n: (state["ts"].filter_by_overlap(o, n), _ret())[1]
    merging.remerge = lambda linker, tr: (state["ts"].remerge(vars(linker.config)), _ret())[1]
    triangulation = types.ModuleType("limap.triangulation")
    triangulation.GlobalLineTriangulator = GlobalLineTriangulator
    return triangulation, merging


def test_reference_runner_body_wiring_on_cpu(oracle, tmp_path):
    """CPU form (runs in the build container, where /root/reference is): the real caller's body with this package's
    `base` / `io` and an oracle-backed triangulator + merging module -- every attribute the caller reads of limap.base and
    limap.util.io exists here and behaves (files written, neighbours truncated, var2d resolved)."""
    if not os.path.exists(RUNNER):
        pytest.skip("/root/reference is not present")
    import yaml
    from limap_amd import base, io as ltio
    sc = syn.make_scene(n_views=14, n_segs=90, n_neighbors=6, seed=31)
    cfg = yaml.safe_load(open(DEFAULT_YAML))
    cfg.update(dir_save=str(tmp_path / "out"), visualize=False, n_neighbors=5, n_visible_views=3)
    cfg["line2d"]["detector"]["method"] = "lsd"
    cfg["refinement"]["disable"] = True
    tri_mod, merge_mod = _oracle_backed_modules(oracle, sc)
    runner, calls = _load_runner(sc, 6, triangulation=tri_mod, merging=merge_mod)
    imagecols = base.ImageCollection.from_arrays(sc.img_ids, sc.kvec, sc.qvec, sc.tvec)
    neighbors = {int(i): [int(n) for n in sc.neighbors[int(i)]] for i in sc.img_ids}
    tracks = runner.line_triangulation(cfg, imagecols, neighbors=neighbors, ranges=sc.ranges)
    assert cfg["triangulation"]["var2d"] == 2.0 and len(tracks) > 3 and calls[-1] == ("report", len(tracks))
    nb2, rng2 = ltio.read_txt_metainfos(os.path.join(cfg["dir_save"], "metainfos.txt"))
    assert all(len(v) <= 5 for v in nb2.values()) and np.array_equal(rng2[0], sc.ranges[0])
    back = ltio.read_folder_linetracks(os.path.join(cfg["dir_save"], "finaltracks"))
    assert [t.image_id_list for t in back] == [t.image_id_list for t in tracks]
    ic2 = ltio.read_imagecols(os.path.join(cfg["dir_save"], "imagecols.npy"))
    assert ic2.get_img_ids() == imagecols.get_img_ids()


@pytest.mark.gpu
@pytest.mark.parametrize("exhaustive", [False, True])
def test_reference_runner_body_on_this_backend(gpu_lib, oracle, tmp_path, exhaustive):
    if not os.path.exists(RUNNER):
        pytest.skip("/root/reference is not present (the runner's source cannot travel to the GPU box)")
    import yaml
    from limap_amd import base
    sc = syn.make_scene(n_views=14, n_segs=90, n_neighbors=6, seed=31)
    topk = 6
    cfg = yaml.safe_load(open(DEFAULT_YAML))
    cfg["dir_save"] = str(tmp_path / "out")
    cfg["line2d"]["detector"]["method"] = "lsd"       # var2d = -1 -> cfg["var2d"]["lsd"] = 2.0 (runner :39-40)
    cfg["visualize"] = False
    cfg["refinement"]["disable"] = True
    cfg["n_neighbors"] = 5                             # the runner truncates the neighbour lists (:71-73)
    cfg["n_visible_views"] = 3
    cfg["triangulation"]["use_exhaustive_matcher"] = exhaustive
    runner, calls = _load_runner(sc, topk)
    imagecols = base.ImageCollection.from_arrays(sc.img_ids, sc.kvec, sc.qvec, sc.tvec)
    neighbors = {int(i): [int(n) for n in sc.neighbors[int(i)]] for i in sc.img_ids}
    tracks = runner.line_triangulation(cfg, imagecols, neighbors={k: list(v) for k, v in neighbors.items()}, ranges=sc.ranges)
    assert ("compute_2d_segs", not exhaustive) in calls and calls[-1][0] == "report"
    assert (("compute_matches", sc.n_images) in calls) == (not exhaustive)
    for f in ("imagecols.npy", "metainfos.txt", "image_list.txt", "alltracks.txt", "triangulated_lines_nv3.obj"):
        assert os.path.exists(os.path.join(cfg["dir_save"], f)), f
    assert len(os.listdir(os.path.join(cfg["dir_save"], "finaltracks"))) == len(tracks) + 3  # + config / imagecols / segs

    # ---- the same sequence on the CPU oracle ----
    tcfg = dict(cfg["triangulation"])
    assert tcfg["var2d"] == 2.0
    O = oracle.OracleTriangulator(tcfg, faithful=False)
    O.SetRanges(sc.ranges)
    O.Init(sc.img_ids, sc.kvec, sc.qvec, sc.tvec, sc.seg_off, sc.segs)
    for i in sc.img_ids:
        nb = neighbors[int(i)][:cfg["n_neighbors"]]
        if exhaustive:
            O.TriangulateImageExhaustiveMatch(int(i), nb)
        else:
            m = sc.matches_of(int(i), topk)
            O.TriangulateImage(int(i), m)   # the runner passes the whole matches dict: ids outside `neighbors` are skipped
    O.ComputeLineTracks()
    ts = oracle.OracleTrackSet(O)
    f2d = tcfg["filtering2d"]
    ts.filter_by_reprojection(f2d["th_angular_2d"], f2d["th_perp_2d"])
    ts.remerge(tcfg["remerging"]["linker3d"])
    ts.filter_by_reprojection(f2d["th_angular_2d"], f2d["th_perp_2d"])
    ts.filter_by_sensitivity(f2d["th_sv_angular_3d"], f2d["th_sv_num_supports"])
    ts.filter_by_overlap(f2d["th_overlap"], f2d["th_overlap_num_supports"])
    want = ts.get()
    n = len(want["off"]) - 1
    assert len(tracks) == n > 10
    for k, tr in enumerate(tracks):
        a, b = int(want["off"][k]), int(want["off"][k + 1])
        assert tr.image_id_list == want["image_ids"][a:b].tolist() and tr.line_id_list == want["line_ids"][a:b].tolist()
        assert tr.node_id_list == want["node_ids"][a:b].tolist()
        got = np.concatenate([tr.line.start, tr.line.end])
        scale = max(np.abs(want["line"][k, :6]).max(), 1e-9)
        assert np.abs(got - want["line"][k, :6]).max() / scale <= 1e-5, (k, got, want["line"][k])


@pytest.mark.gpu
@pytest.mark.parametrize("exhaustive", [False, True])
def test_recorded_caller_trace_on_this_backend(gpu_lib, exhaustive):
    """Runs on the GPU box, where the reference tree is absent: tests/golden/caller_trace.json is the record of every
    call the reference's own runner body made on `limap.triangulation` / `limap.merging` (written by
    tests/golden/make_caller_trace.py in the build container, answered there by the CPU oracle) -- replayed here call
    by call, same order, same arguments, against this backend.  The tracks must be the ones the caller got there."""
    import json
    import zlib
    from limap_amd import base, merging, triangulation as tri
    doc = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "caller_trace.json")))
    run = [r for r in doc["runs"] if r["exhaustive"] == exhaustive][0]
    sc = syn.make_scene(**doc["scene"])
    imagecols = base.ImageCollection.from_arrays(sc.img_ids, sc.kvec, sc.qvec, sc.tvec)
    all_2d_segs = base.get_all_lines_2d({int(i): sc.segs_of(k) for k, i in enumerate(sc.img_ids)})
    T, tracks = None, None
    names = []
    for call in run["calls"]:
        name, args = call[0], call[1:]
        names.append(name)
        if name == "GlobalLineTriangulator":
            T = tri.GlobalLineTriangulator(args[0])
        elif name == "SetRanges":
            T.SetRanges((np.asarray(args[0][0]), np.asarray(args[0][1])))
        elif name == "Init":
            assert args[0] == imagecols.get_img_ids() and args[1] == [len(all_2d_segs[i]) for i in args[0]]
            T.Init(all_2d_segs, imagecols)
        elif name == "TriangulateImage":
            img_id, keys, counts, crc = args
            full = sc.matches_of(img_id, doc["topk"])
            m = {k: full[k] for k in keys}
            got = 0
            for k in keys:
                got = zlib.crc32(np.ascontiguousarray(m[k], dtype=np.int32).tobytes(), got)
            assert [len(m[k]) for k in keys] == counts and got == crc, "the scene generator changed: regenerate the trace"
            T.TriangulateImage(img_id, m)
        elif name == "TriangulateImageExhaustiveMatch":
            T.TriangulateImageExhaustiveMatch(args[0], args[1])
        elif name == "ComputeLineTracks":
            tracks = T.ComputeLineTracks()
        elif name == "remerge":
            tracks = merging.remerge(base.LineLinker3d(args[0]), tracks)
        else:
            tracks = getattr(merging, name)(tracks, imagecols, args[0], args[1])
    assert names[:3] == ["GlobalLineTriangulator", "SetRanges", "Init"] and "ComputeLineTracks" in names
    want = run["tracks"]
    assert len(tracks) == len(want) > 10
    for k, (tr, w) in enumerate(zip(tracks, want)):
        assert list(tr.image_id_list) == w["image_ids"] and list(tr.line_id_list) == w["line_ids"], k
        assert list(tr.node_id_list) == w["node_ids"], k
        got = np.concatenate([tr.line.start, tr.line.end])
        scale = max(np.abs(np.asarray(w["line"])).max(), 1e-9)
        assert np.abs(got - np.asarray(w["line"])).max() / scale <= 1e-9, (k, got, w["line"])
