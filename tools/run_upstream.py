"""Upstream kit, step 2 -- run where cvg/limap is installed (CPU is enough):  python tools/run_upstream.py OUT/<case>
Reads the folder tools/export_scene_for_limap.py wrote, runs limap's own GlobalLineTriangulator through the call sequence of
limap.runners.line_triangulation (:102-168, no detection / matching / post-filters) and writes what ComputeLineTracks()
returned to OUT/<case>/upstream/track_*.txt.  Compare with tools/diff_limap_dump.py (see export_scene_for_limap.py)."""
import json, os, sys
import limap.base as _base, limap.triangulation as _tri, limap.util.io as limapio

d = sys.argv[1]
cfg = json.load(open(os.path.join(d, "cfg.json")))["triangulation"]
imagecols = _base.ImageCollection(limapio.read_npy(os.path.join(d, "imagecols.npy")).item())
neighbors, ranges = limapio.read_txt_metainfos(os.path.join(d, "metainfos.txt"))
all_2d_segs = {i: limapio.read_txt_segments(os.path.join(d, "segments"), i) for i in imagecols.get_img_ids()}
all_2d_lines = _base.get_all_lines_2d(all_2d_segs)
T = _tri.GlobalLineTriangulator(cfg)
T.SetRanges(ranges)
T.Init(all_2d_lines, imagecols)
for img_id in imagecols.get_img_ids():
    if cfg.get("use_exhaustive_matcher"):
        T.TriangulateImageExhaustiveMatch(img_id, neighbors[img_id])
    else:
        T.TriangulateImage(img_id, limapio.read_npy(os.path.join(d, "matches", f"matches_{img_id}.npy")).item())
tracks = T.ComputeLineTracks()
limapio.save_folder_linetracks(os.path.join(d, "upstream"), tracks)
print(f"{len(tracks)} tracks -> {os.path.join(d, 'upstream')}")
