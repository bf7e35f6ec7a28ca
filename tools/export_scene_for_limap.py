"""Upstream kit, step 1 (any machine, no GPU): write the synthetic scenes this backend is tested on as limap OUTPUT FOLDERS --
what `limap.runners.line_triangulation` leaves behind (runners/line_triangulation.py:56-97: imagecols.npy, metainfos.txt,
segments/segments_<id>.txt, matches/matches_<id>.npy) -- plus cfg.json (the `triangulation` section) and, with --expected, the
tracks the CPU checker of this repository computes for the scene (expected/track_*.txt, LineTrack::Write format).

    python tools/export_scene_for_limap.py OUT [--cases config2 matched_s11 ...] [--expected]

Step 2, where limap is installed:   python tools/run_upstream.py OUT/<case>       -> OUT/<case>/upstream/track_*.txt
Step 3, on the MI355X box:          python tools/diff_limap_dump.py --imagecols OUT/<case>/imagecols.npy
                                        --metainfos OUT/<case>/metainfos.txt --segments OUT/<case>/segments
                                        --matches OUT/<case>/matches --tracks OUT/<case>/upstream --cfg OUT/<case>/cfg.json
        (or, without a GPU, compare OUT/<case>/upstream with OUT/<case>/expected: `--ours-folder OUT/<case>/expected`).
The folders are deterministic: tests/golden/export_digests.json holds the SHA-256 of every case's input files
(tests/test_io_formats.py regenerates the small cases and compares)."""
import argparse
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from limap_amd import io as ltio, synthetic as syn  # noqa: E402
from limap_amd.base import ImageCollection  # noqa: E402

# name -> (mode, seed, n_views, n_segs, n_neighbors, topk, cfg overrides): config2 = BASELINE.json configs[1]; the rest are the
# scenes of tests/golden/make_golden.py (the VP scene needs limap's VP detector results and is not exported)
CASES = {
    "config2": ("matched", 0, 100, 500, 20, 10, {}),
    "matched_s11": ("matched", 11, 16, 100, 8, 6, {}),
    "exhaustive_s12": ("exhaustive", 12, 14, 60, 8, 0, {}),
    "matched_outer2_halfpix_s13": ("matched", 13, 14, 80, 8, 6, dict(add_halfpix=True, min_num_outer_edges=2)),
    "matched_endpoints_s14": ("matched", 14, 10, 60, 6, 5, dict(use_endpoints_triangulation=True)),
}


def export_case(out, name):
    mode, seed, n_views, n_segs, nn, topk, over = CASES[name]
    sc = syn.make_scene(n_views=n_views, n_segs=n_segs, n_neighbors=nn, seed=seed, topk=topk or None)
    d = os.path.join(out, name)
    os.makedirs(d, exist_ok=True)
    ltio.save_imagecols(os.path.join(d, "imagecols.npy"), ImageCollection.from_arrays(sc.img_ids, sc.kvec, sc.qvec, sc.tvec))
    ltio.save_txt_metainfos(os.path.join(d, "metainfos.txt"), {int(i): sc.neighbors[int(i)] for i in sc.img_ids}, sc.ranges)
    for n, i in enumerate(sc.img_ids):
        ltio.save_txt_segments(os.path.join(d, "segments"), int(i), sc.segs_of(n))
        if mode == "matched":
            ltio.save_matches(os.path.join(d, "matches"), int(i), sc.matches_of(int(i), topk))
    cfg = syn.default_triangulation_cfg(**over)
    cfg["use_exhaustive_matcher"] = mode == "exhaustive"
    json.dump({"triangulation": cfg}, open(os.path.join(d, "cfg.json"), "w"), indent=1, sort_keys=True)
    return sc, cfg, d


def _canon(o):
    """pickled .npy content in a form that does not depend on the pickle / numpy version that wrote it"""
    if isinstance(o, dict):
        return {str(k): _canon(v) for k, v in sorted(o.items(), key=lambda kv: str(kv[0]))}
    if isinstance(o, np.ndarray) and o.dtype == object:
        return _canon(o.item() if o.ndim == 0 else o.tolist())
    if isinstance(o, np.ndarray):
        return [str(o.dtype), list(o.shape), hashlib.sha256(np.ascontiguousarray(o).tobytes()).hexdigest()]
    if isinstance(o, (list, tuple)):
        return [_canon(v) for v in o]
    return o


def folder_digest(d):
    """SHA-256 over the case's INPUT files in sorted relative-path order: text files by their bytes, .npy files (pickled
    dicts) by their canonical content."""
    h = hashlib.sha256()
    files = []
    for root, _, names in os.walk(d):
        if os.path.basename(root) in ("expected", "upstream"):
            continue
        files += [os.path.join(root, n) for n in names]
    for f in sorted(files, key=lambda f: os.path.relpath(f, d)):
        if f.endswith(".npy"):
            body = json.dumps(_canon(ltio.read_npy(f)), sort_keys=True).encode()
        else:
            body = open(f, "rb").read()
        h.update(os.path.relpath(f, d).encode() + b"\0" + body + b"\0")
    return h.hexdigest()


def write_expected(sc, cfg, d, exhaustive):
    """the CPU checker's tracks for the scene (test infrastructure: tools/ and tests/ may use it, the product never does)"""
    from oracle import oracle as ora
    ora.build()
    O = ora.OracleTriangulator(cfg, faithful=False)
    O.SetRanges(sc.ranges)
    O.Init(sc.img_ids, sc.kvec, sc.qvec, sc.tvec, sc.seg_off, sc.segs)
    for i in sc.img_ids:
        if exhaustive:
            O.TriangulateImageExhaustiveMatch(int(i), sc.neighbors[int(i)])
        else:
            O.TriangulateImage(int(i), sc.matches_of(int(i), CASES_TOPK[0]))
    t = O.ComputeLineTracks()
    from limap_amd.base import Line2d, Line3d, LineTrack
    tracks = []
    for n in range(len(t["off"]) - 1):
        s = slice(int(t["off"][n]), int(t["off"][n + 1]))
        r = t["line"][n]
        imgs, lids = t["image_ids"][s].tolist(), t["line_ids"][s].tolist()
        segs = [sc.segs_of(int(np.searchsorted(sc.img_ids, i)))[l] for i, l in zip(imgs, lids)]
        tracks.append(LineTrack(Line3d(r[0:3], r[3:6]), imgs, lids, [Line2d(g[0:2], g[2:4]) for g in segs]))
    ltio.save_folder_linetracks(os.path.join(d, "expected"), tracks)
    return len(tracks)


CASES_TOPK = [None]

if __name__ == "__main__":
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("out")
    ap.add_argument("--cases", nargs="*", default=list(CASES))
    ap.add_argument("--expected", action="store_true")
    args = ap.parse_args()
    rep = {}
    for name in args.cases:
        sc, cfg, d = export_case(args.out, name)
        rep[name] = {"sha256_inputs": folder_digest(d), "images": int(sc.n_images), "segments": int(sc.seg_off[-1])}
        if args.expected:
            CASES_TOPK[0] = CASES[name][5] or None
            rep[name]["expected_tracks"] = write_expected(sc, cfg, d, CASES[name][0] == "exhaustive")
        print(name, rep[name], flush=True)
    json.dump(rep, open(os.path.join(args.out, "digests.json"), "w"), indent=1, sort_keys=True)
