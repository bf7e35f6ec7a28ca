"""Records what THE REAL CALLER does, so that it can be replayed where the reference tree is absent (the GPU box).

Run in the build container (where /root/reference exists): the body of the reference's own
`limap.runners.line_triangulation.line_triangulation` (src/limap/runners/line_triangulation.py:18-205) is loaded from
/root/reference as it is -- the same loader as tests/test_caller_replay.py -- and executed on a seeded synthetic scene
with a `limap.triangulation` / `limap.merging` pair that (a) RECORDS every call the caller makes on them, in order, with
its arguments, and (b) answers from the CPU oracle.  The record and the tracks the caller returned are written to
tests/golden/caller_trace.json; tests/test_caller_replay.py::test_recorded_caller_trace_on_this_backend replays the
record call by call against the HIP backend and compares the outcome.

usage: python tests/golden/make_caller_trace.py [out.json]
"""
import json
import os
import sys
import tempfile
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

SCENE = dict(n_views=14, n_segs=90, n_neighbors=6, seed=31)
TOPK = 6


def rows_digest(matches):
    """(neighbour ids in the order the caller passed them, rows per neighbour, crc32 over all rows)"""
    keys = [int(k) for k in matches.keys()]
    crc = 0
    for k in matches.keys():
        crc = zlib.crc32(np.ascontiguousarray(matches[k], dtype=np.int32).tobytes(), crc)
    return keys, [int(len(matches[k])) for k in matches.keys()], int(crc)


def plain(x):
    if isinstance(x, dict):
        return {str(k): plain(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [plain(v) for v in x]
    if isinstance(x, np.ndarray):
        return x.tolist()
    if isinstance(x, (np.integer,)):
        return int(x)
    if isinstance(x, (np.floating,)):
        return float(x)
    return x


def record(exhaustive):
    import yaml
    import test_caller_replay as tcr
    from limap_amd import base, synthetic as syn
    from oracle import oracle as ora
    ora.build()
    sc = syn.make_scene(**SCENE)
    cfg = yaml.safe_load(open(tcr.DEFAULT_YAML))
    out_dir = tempfile.mkdtemp(prefix="caller_trace_")
    cfg.update(dir_save=os.path.join(out_dir, "out"), visualize=False, n_neighbors=5, n_visible_views=3)
    cfg["line2d"]["detector"]["method"] = "lsd"
    cfg["refinement"]["disable"] = True
    cfg["triangulation"]["use_exhaustive_matcher"] = bool(exhaustive)
    tri_mod, merge_mod = tcr._oracle_backed_modules(ora, sc)
    trace = []
    Inner = tri_mod.GlobalLineTriangulator

    class Recording(Inner):
        def __init__(self, c):
            trace.append(["GlobalLineTriangulator", plain(dict(c))])
            super().__init__(c)

        def SetRanges(self, ranges):
            trace.append(["SetRanges", plain([np.asarray(ranges[0]), np.asarray(ranges[1])])])
            super().SetRanges(ranges)

        def Init(self, all_2d_lines, imagecols):
            trace.append(["Init", [int(i) for i in imagecols.get_img_ids()],
                          [int(len(all_2d_lines[i])) for i in imagecols.get_img_ids()]])
            super().Init(all_2d_lines, imagecols)

        def TriangulateImage(self, img_id, matches):
            trace.append(["TriangulateImage", int(img_id), *rows_digest(matches)])
            super().TriangulateImage(img_id, matches)

        def TriangulateImageExhaustiveMatch(self, img_id, neighbors):
            trace.append(["TriangulateImageExhaustiveMatch", int(img_id), [int(n) for n in neighbors]])
            super().TriangulateImageExhaustiveMatch(img_id, neighbors)

        def ComputeLineTracks(self):
            trace.append(["ComputeLineTracks"])
            return super().ComputeLineTracks()
    tri_mod.GlobalLineTriangulator = Recording
    for name in ("filter_tracks_by_reprojection", "filter_tracks_by_sensitivity", "filter_tracks_by_overlap"):
        def wrap(fn, name=name):
            def f(tracks, imagecols, a, b):
                trace.append([name, float(a), plain(b)])
                return fn(tracks, imagecols, a, b)
            return f
        setattr(merge_mod, name, wrap(getattr(merge_mod, name)))
    inner_remerge = merge_mod.remerge

    def remerge(linker, tracks):
        trace.append(["remerge", plain(vars(linker.config))])
        return inner_remerge(linker, tracks)
    merge_mod.remerge = remerge
    runner, calls = tcr._load_runner(sc, TOPK, triangulation=tri_mod, merging=merge_mod)
    imagecols = base.ImageCollection.from_arrays(sc.img_ids, sc.kvec, sc.qvec, sc.tvec)
    neighbors = {int(i): [int(n) for n in sc.neighbors[int(i)]] for i in sc.img_ids}
    tracks = runner.line_triangulation(cfg, imagecols, neighbors=neighbors, ranges=sc.ranges)
    result = [dict(image_ids=[int(v) for v in t.image_id_list], line_ids=[int(v) for v in t.line_id_list],
                   node_ids=[int(v) for v in t.node_id_list],
                   line=[float(v) for v in np.concatenate([t.line.start, t.line.end])]) for t in tracks]
    return dict(exhaustive=bool(exhaustive), calls=trace, tracks=result)


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(HERE, "caller_trace.json")
    doc = dict(source="src/limap/runners/line_triangulation.py:18-205, executed from /root/reference",
               scene=SCENE, topk=TOPK, runs=[record(False), record(True)])
    with open(out, "w") as f:
        json.dump(doc, f, separators=(",", ":"))
        f.write("\n")
    for r in doc["runs"]:
        print("exhaustive" if r["exhaustive"] else "matched", len(r["calls"]), "calls,", len(r["tracks"]), "tracks")


if __name__ == "__main__":
    main()
