"""Offline cross-check against a REAL limap installation.

The GPU box has no limap, and the build container has neither a GPU nor Eigen / COLMAP / ceres -- so the comparison
with an upstream build is split in two:

  1. where limap runs (any machine, CPU): `runners/line_triangulation.py` already leaves its intermediate artefacts in
     its output folder (runners/line_triangulation.py:56-97: imagecols.npy, metainfos.txt, segments/, the matches
     folder) and its result (`alltracks.txt` via limapio.save_txt_linetracks, `finaltracks/` via
     limapio.save_folder_linetracks_with_info).  Optionally dump the per-node best candidates BEFORE the
     post-triangulation filters (they isolate the hot path from merging / remerge):
         np.save("best_tris.npy", {i: [l.as_array() if l is not None else None for l in Triangulator.GetAllBestTris()[i]] ...})
     or simply the tracks returned by ComputeLineTracks() with limapio.save_folder_linetracks(folder, tracks).
  2. on the MI355X box: this tool re-runs the triangulation on the same artefacts with limap_amd and diffs.

    python tools/diff_limap_dump.py --imagecols out/imagecols.npy --metainfos out/metainfos.txt \
        --segments out/segments --matches out/matches --tracks out/tracks_before_filters [--cfg cfg.yaml] [--exhaustive]

`--tracks` is a folder of track_*.txt files (limapio.save_folder_linetracks / LineTrack::Write) holding what upstream's
ComputeLineTracks returned.  What is compared, per track (tracks are matched by their member sets, the numbering of
the union-find is not part of the contract): members (image_id, line_id) EXACTLY; the endpoints of `LineTrack.line` to
1e-5 relative (north_star's tolerance); a track whose endpoints agree only after exchanging start and end is counted
SEPARATELY as `swapped` -- the direction of an aggregated line comes from `JacobiSVD::matrixV().col(0)`, whose sign this
backend reproduces by Eigen 3.4's published procedure (limap_amd/csrc/lt_svd.h) but cannot check without Eigen on disk.
Exit code 0: every member set found and every endpoint pair within tolerance (swaps allowed, reported); 1 otherwise.
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def member_key(track):
    return tuple(sorted(zip((int(i) for i in track.image_id_list), (int(i) for i in track.line_id_list))))


def endpoints(track):
    ln = track.line
    return np.asarray(ln.start, float), np.asarray(ln.end, float)


def compare(ours, theirs, rtol=1e-5):
    """-> dict report; `ours` / `theirs`: lists of LineTrack-like objects (image_id_list, line_id_list, line.start/.end)."""
    mine = {member_key(t): t for t in ours}
    rep = {"n_ours": len(ours), "n_theirs": len(theirs), "missing_member_sets": 0, "extra_member_sets": 0,
           "matched": 0, "swapped": 0, "endpoint_mismatch": 0, "max_rel_err": 0.0, "examples": []}
    seen = set()
    for t in theirs:
        k = member_key(t)
        if k not in mine:
            rep["missing_member_sets"] += 1
            if len(rep["examples"]) < 5:
                rep["examples"].append({"missing": list(k)[:6]})
            continue
        seen.add(k)
        s0, e0 = endpoints(mine[k])
        s1, e1 = endpoints(t)
        scale = max(np.abs(s1).max(), np.abs(e1).max(), 1e-12)
        direct = max(np.abs(s0 - s1).max(), np.abs(e0 - e1).max()) / scale
        swapped = max(np.abs(s0 - e1).max(), np.abs(e0 - s1).max()) / scale
        if direct <= rtol:
            rep["matched"] += 1
            rep["max_rel_err"] = max(rep["max_rel_err"], float(direct))
        elif swapped <= rtol:
            rep["swapped"] += 1
            rep["max_rel_err"] = max(rep["max_rel_err"], float(swapped))
        else:
            rep["endpoint_mismatch"] += 1
            if len(rep["examples"]) < 5:
                rep["examples"].append({"members": list(k)[:6], "direct": float(direct), "swapped": float(swapped)})
    rep["extra_member_sets"] = len(set(mine) - seen)
    rep["ok"] = rep["missing_member_sets"] == 0 and rep["extra_member_sets"] == 0 and rep["endpoint_mismatch"] == 0
    return rep


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--ours-folder", default=None,
                    help="compare --tracks with THIS folder of track_*.txt instead of running the scene here (no GPU needed: "
                         "e.g. the expected/ folder tools/export_scene_for_limap.py --expected wrote)")
    ap.add_argument("--imagecols", default=None)
    ap.add_argument("--metainfos", default=None)
    ap.add_argument("--segments", default=None)
    ap.add_argument("--matches", default=None, help="folder of matches_<img_id>.npy (not needed with --exhaustive)")
    ap.add_argument("--tracks", required=True, help="folder of track_*.txt written by upstream (limapio.save_folder_linetracks)")
    ap.add_argument("--cfg", default=None, help="yaml / json with the `triangulation` section upstream ran with (default: limap's defaults)")
    ap.add_argument("--exhaustive", action="store_true")
    ap.add_argument("--rtol", type=float, default=1e-5)
    args = ap.parse_args()

    from limap_amd import io as ltio
    from limap_amd import synthetic as syn
    cfg = syn.default_triangulation_cfg()
    if args.cfg:
        if args.cfg.endswith((".yaml", ".yml")):
            import yaml
            user = yaml.safe_load(open(args.cfg))
        else:
            user = json.load(open(args.cfg))
        cfg.update(user.get("triangulation", user))
    theirs = ltio.read_folder_linetracks(args.tracks)
    if args.ours_folder:
        rep = compare(ltio.read_folder_linetracks(args.ours_folder), theirs, args.rtol)
        print(json.dumps(rep, indent=1))
        return 0 if rep["ok"] else 1
    if not (args.imagecols and args.metainfos and args.segments):
        ap.error("--imagecols, --metainfos and --segments are needed unless --ours-folder is given")
    _, ours = ltio.triangulate_scene_folder(args.imagecols, args.metainfos, args.segments, args.matches, cfg,
                                            exhaustive=args.exhaustive)
    rep = compare(ours, theirs, args.rtol)
    print(json.dumps(rep, indent=1))
    return 0 if rep["ok"] else 1


if __name__ == "__main__":
    sys.exit(main())
