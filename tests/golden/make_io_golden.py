"""Golden files for the on-disk formats, written by THE REFERENCE'S OWN CODE (run in the build container, where
/root/reference exists; the outputs are committed under tests/golden/io/ and compared byte for byte with what
limap_amd/io.py writes -- tests/test_io_formats.py):

  * /root/reference/src/limap/util/io.py, imported as it is under three stub modules (pycolmap.logging, tqdm and a
    limap.base whose LineTrack writes / reads through limap::LineTrack::Write / Read of oracle/_ref, i.e. the
    reference's base/linetrack.cc compiled unmodified): save_txt_metainfos, save_txt_segments, save_npy (matches_*.npy
    as line2d/base_matcher.py:86-100 writes them), save_txt_linetracks, save_folder_linetracks;
  * limap::ImageCollection::as_dict() of oracle/_ref for imagecols.npy.

usage: python tests/golden/make_io_golden.py [out_dir]     (default tests/golden/io)
"""
import importlib.util
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
REF_IO = "/root/reference/src/limap/util/io.py"


def fixed_inputs():
    """The inputs of every golden file (shared with the tests)."""
    rng = np.random.default_rng(20240924)
    neighbors = {0: [3, 1], 7: [], 3: [0, 7, 1], 1: [0]}
    ranges = (np.array([-1.5, 0.25, 1e-3]), np.array([2.0, 3.5, 9.0]))
    segs = rng.uniform(0, 800, (37, 4))
    segs[3] = [1.0, 2.5, 1e-7, 123456789.125]
    matches = {4: np.array([[0, 1], [0, 5], [2, 2]], np.int32), 9: np.zeros((0, 2), np.int32),
               1: rng.integers(0, 50, (11, 2)).astype(np.int32)}
    tracks = []
    for t in range(3):
        n = [3, 1, 6][t]
        tr = dict(line=rng.normal(size=6) * [1, 10, 100, 1e-3, 1, 1e4],
                  image_ids=rng.integers(0, 4 + 3 * t, n).astype(np.int32), line_ids=rng.integers(0, 500, n).astype(np.int32),
                  line2d=rng.uniform(0, 800, (n, 4)), node_ids=rng.integers(0, 5000, n).astype(np.int32),
                  scores=rng.uniform(0, 12, n), line3d=rng.normal(size=(n, 6)))
        tracks.append(tr)
    tracks[1]["line"][2] = np.nan  # LineTrack::Write replaces a NaN coordinate of the track line by 0
    cams = dict(img_ids=np.array([3, 7, 12], np.int32),
                kvec=np.array([[500.0, 501.5, 320.0, 240.0], [400.0, 400.0, 300.25, 200.0], [886.81, 886.81, 512.0, 384.0]]),
                qvec=np.array([[1.0, 0, 0, 0], [0.5, 0.5, 0.5, 0.5], [0.9, 0.1, -0.3, 0.2]]),
                tvec=np.array([[0.0, 0, 1], [1, 2, 3], [-0.25, 7.5, 1e-3]]))
    return dict(neighbors=neighbors, ranges=ranges, segs=segs, matches=matches, tracks=tracks, cams=cams)


def reference_io():
    """The reference's util/io.py under stub modules."""
    from oracle import ref

    class _Line:
        def __init__(self, v):
            self.start, self.end = np.array(v[:len(v) // 2], float), np.array(v[len(v) // 2:], float)

    class LineTrack:  # what util/io.py touches of limap.base.LineTrack
        def __init__(self, d=None):
            self.d = d

        @property
        def line(self):
            return _Line(self.d["line"])

        @property
        def image_id_list(self):
            return [int(i) for i in self.d["image_ids"]]

        @property
        def line_id_list(self):
            return [int(i) for i in self.d["line_ids"]]

        def count_lines(self):
            return len(self.d["image_ids"])

        def count_images(self):
            return len(set(int(i) for i in self.d["image_ids"]))

        def Write(self, fname):  # limap::LineTrack::Write
            d = self.d
            ref.track_write(fname, d["line"], d["image_ids"], d["line_ids"], d["line2d"], d.get("node_ids"),
                            d.get("scores"), d.get("line3d"))

        def Read(self, fname):  # limap::LineTrack::Read
            self.d = ref.track_read(fname)

    stubs = {}
    pycolmap = types.ModuleType("pycolmap")
    pycolmap.logging = types.SimpleNamespace(info=lambda *a, **k: None, warning=lambda *a, **k: None)
    stubs["pycolmap"] = pycolmap
    tqdm = types.ModuleType("tqdm")
    tqdm.tqdm = lambda it, *a, **k: it
    stubs["tqdm"] = tqdm
    limap = types.ModuleType("limap")
    limap.base = types.ModuleType("limap.base")
    limap.base.LineTrack = LineTrack
    stubs["limap"] = limap
    stubs["limap.base"] = limap.base
    saved = {k: sys.modules.get(k) for k in stubs}
    sys.modules.update(stubs)
    try:
        spec = importlib.util.spec_from_file_location("_reference_limap_util_io", REF_IO)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    return mod, LineTrack


def write_all(out):
    from oracle import ref
    rio, LineTrack = reference_io()
    x = fixed_inputs()
    os.makedirs(out, exist_ok=True)
    rio.save_txt_metainfos(os.path.join(out, "metainfos.txt"), x["neighbors"], x["ranges"])
    rio.save_txt_segments(out, 12, x["segs"])
    rio.save_txt_segments(out, 13, np.zeros((0, 4)))
    rio.save_npy(os.path.join(out, "matches_3.npy"), x["matches"])  # base_matcher.save_match
    tracks = [LineTrack(t) for t in x["tracks"]]
    rio.save_txt_linetracks(os.path.join(out, "alltracks_nv1.txt"), tracks, n_visible_views=1)
    rio.save_txt_linetracks(os.path.join(out, "alltracks_nv4.txt"), tracks, n_visible_views=4)
    rio.save_folder_linetracks(os.path.join(out, "finaltracks"), tracks)
    # a track file without the auxiliary lists
    t0 = dict(x["tracks"][0])
    for k in ("node_ids", "scores", "line3d"):
        t0.pop(k)
    LineTrack(t0).Write(os.path.join(out, "track_noaux.txt"))
    c = x["cams"]
    rio.save_npy(os.path.join(out, "imagecols.npy"), ref.imagecols_as_dict(c["img_ids"], c["kvec"], c["qvec"], c["tvec"]))


if __name__ == "__main__":
    write_all(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tests", "golden", "io"))
    print("golden io files written")
