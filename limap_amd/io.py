"""On-disk formats around the triangulation boundary, so that folders written by a real limap run
(configs 4-5 of BASELINE.json) can be fed to this backend without limap installed, and its output
read back by limap's tools.  Formats follow the reference (paths relative to src/limap/):

* ``metainfos.txt``   neighbours + ranges                      util/io.py:87-131
* ``segments_{id}.txt``  2D segments of one image              util/io.py:441-465
* ``matches_{id}.npy``   pickled dict ng_img_id -> (K,2) int   line2d/base_matcher.py:77-115, util/io.py:39-48
* ``imagecols.npy``      pickled ``ImageCollection.as_dict()``  base/image_collection.cc:158-171,
                                                                base/camera.cc:288-296, base/camera_view.cc:17-23
* ``track_{i}.txt``      one LineTrack                          base/linetrack.cc:133-260
* ``alltracks.txt``      all tracks with >= n_visible_views     util/io.py:259-292
"""
import os

import numpy as np

from .base import CameraView, ImageCollection, Line2d, Line3d, LineTrack


# ---- metainfos.txt ---------------------------------------------------------------------------
def save_txt_metainfos(fname, neighbors, ranges):
    os.makedirs(os.path.dirname(os.path.abspath(fname)), exist_ok=True)
    lo, hi = np.asarray(ranges[0], float), np.asarray(ranges[1], float)
    with open(fname, "w") as f:
        f.write(f"number of images, {len(neighbors)}\n")
        for axis, name in enumerate("xyz"):
            f.write(f"{name}-range, {lo[axis]}, {hi[axis]}\n")
        for img_id, nbs in neighbors.items():
            f.write(", ".join([f"image {img_id}"] + [str(int(n)) for n in nbs]) + "\n")


def read_txt_metainfos(fname):
    with open(fname) as f:
        rows = [r.strip() for r in f.readlines()]
    n_images = int(rows[0].split(",")[1])
    lo, hi = np.zeros(3), np.zeros(3)
    for axis in range(3):
        parts = rows[1 + axis].split(",")
        lo[axis], hi[axis] = float(parts[1]), float(parts[2])
    neighbors = {}
    for r in rows[4:4 + n_images]:
        parts = r.split(",")
        neighbors[int(parts[0][len("image"):])] = [int(p) for p in parts[1:] if p.strip() != ""]
    return neighbors, (lo, hi)


# ---- segments_{id}.txt -----------------------------------------------------------------------
def save_txt_segments(folder, img_id, segs):
    os.makedirs(folder, exist_ok=True)
    segs = np.asarray(segs, float)
    with open(os.path.join(folder, f"segments_{img_id}.txt"), "w") as f:
        f.write(f"{segs.shape[0]}\n")
        for s in segs:
            f.write(f"{s[0]} {s[1]} {s[2]} {s[3]}\n")


def read_txt_segments(folder, img_id):
    with open(os.path.join(folder, f"segments_{img_id}.txt")) as f:
        rows = f.readlines()
    n = int(rows[0].strip())
    if n + 1 != len(rows):
        raise ValueError(f"segments_{img_id}.txt: header says {n} segments, file has {len(rows) - 1}")
    if n == 0:
        return np.zeros((0, 4))
    return np.array([[float(v) for v in r.strip().split(" ")] for r in rows[1:]])


def read_all_segments_from_folder(folder):
    out = {}
    for fname in os.listdir(folder):
        if fname.startswith("segments_") and fname.endswith(".txt"):
            img_id = int(fname[len("segments_"):-4])
            out[img_id] = read_txt_segments(folder, img_id)
    return out


# ---- npy with pickled objects ------------------------------------------------------------------
def save_npy(fname, obj):
    os.makedirs(os.path.dirname(os.path.abspath(fname)), exist_ok=True)
    with open(fname, "wb") as f:
        np.save(f, np.array(obj, dtype=object))


def read_npy(fname):
    with open(fname, "rb") as f:
        return np.load(f, allow_pickle=True)


def save_matches(folder, img_id, matches):
    save_npy(os.path.join(folder, f"matches_{img_id}.npy"), {int(k): np.asarray(v) for k, v in matches.items()})


def read_matches(folder, img_id):
    return read_npy(os.path.join(folder, f"matches_{img_id}.npy")).item()


# ---- imagecols.npy -----------------------------------------------------------------------------
_PINHOLE_LIKE = {  # colmap model id -> (index of fx, fy, cx, cy; indices that must be zero)
    0: ((0, 0, 1, 2), ()),                 # SIMPLE_PINHOLE f, cx, cy
    1: ((0, 1, 2, 3), ()),                 # PINHOLE fx, fy, cx, cy
    2: ((0, 0, 1, 2), (3,)),               # SIMPLE_RADIAL with k == 0
    3: ((0, 0, 1, 2), (3, 4)),             # RADIAL with k1 == k2 == 0
    4: ((0, 1, 2, 3), (4, 5, 6, 7)),       # OPENCV with zero distortion
}


def kvec_from_camera_dict(cam):
    """(fx, fy, cx, cy) of an undistorted camera as stored by ``Camera::as_dict`` (camera.cc:288-296).
    The triangulator requires undistorted views (base_line_triangulator.cc:49)."""
    model = int(cam["model_id"])
    params = [float(p) for p in cam["params"]]
    if model not in _PINHOLE_LIKE:
        raise ValueError(f"camera model {model} is not supported (needs an undistorted pinhole camera)")
    idx, zeros = _PINHOLE_LIKE[model]
    if any(params[z] != 0.0 for z in zeros):
        raise ValueError("Check failed: imagecols->IsUndistorted() == true")
    return np.array([params[i] for i in idx])


def imagecols_from_dict(d):
    cams = {int(k): kvec_from_camera_dict(v) for k, v in d["cameras"].items()}
    views = {}
    for img_id, im in d["images"].items():
        pose = im["pose"]
        views[int(img_id)] = CameraView(cams[int(im["cam_id"])], np.asarray(pose["qvec"], float),
                                        np.asarray(pose["tvec"], float), im.get("image_name", "none"))
    return ImageCollection(views)


def imagecols_to_dict(imagecols, hw=(0, 0)):
    """``ImageCollection::as_dict()`` (base/image_collection.cc:158-171) with one PINHOLE camera per image, as
    ``Camera(model, params, cam_id)`` leaves it (camera.cc:65-75, 265-274: height / width of a camera built without
    them are colmap's defaults 0 / 0, its ``initialized`` list is empty)."""
    cameras, images = {}, {}
    for n, img_id in enumerate(imagecols.get_img_ids()):
        v = imagecols.camview(img_id)
        cameras[n] = dict(model_id=1, params=[float(x) for x in v.kvec], cam_id=n, height=int(hw[0]), width=int(hw[1]),
                          initialized=[])
        images[int(img_id)] = dict(cam_id=n, pose=dict(qvec=np.asarray(v.qvec, float), tvec=np.asarray(v.tvec, float),
                                                       initialized=True), image_name=v.image_name())
    return dict(cameras=cameras, images=images)


def read_imagecols(fname):
    return imagecols_from_dict(read_npy(fname).item())


def save_imagecols(fname, imagecols):
    save_npy(fname, imagecols_to_dict(imagecols))


# ---- line tracks -------------------------------------------------------------------------------
def write_track(fname, track):
    """LineTrack::Write (linetrack.cc:133-209)."""
    def f10(v):
        return f"{0.0 if np.isnan(v) else float(v):.10f}"
    n = track.count_lines()
    with open(fname, "w") as f:
        f.write(" ".join(f10(v) for v in list(track.line.start) + list(track.line.end)) + " \n")
        f.write(f"{n} {track.count_images()}\n")
        f.write("image_id_list " + "".join(f"{i} " for i in track.image_id_list) + "\n")
        f.write("line_id_list " + "".join(f"{i} " for i in track.line_id_list) + "\n")
        f.write("line2d_list\n")
        for l in track.line2d_list:
            f.write(f"{f10(l.start[0])} {f10(l.start[1])} {f10(l.end[0])} {f10(l.end[1])} \n")
        if track.node_id_list:
            f.write("node_id_list " + "".join(f"{i} " for i in track.node_id_list) + "\n")
        if track.score_list:
            f.write("score_list " + "".join(f"{f10(s)} " for s in track.score_list) + "\n")
        if track.line3d_list:
            f.write("line3d_list\n")
            for l in track.line3d_list:
                f.write(" ".join(f10(v) for v in list(l.start) + list(l.end)) + " \n")
        f.write("END\n")


def read_track(fname):
    """LineTrack::Read (linetrack.cc:211-260 and on)."""
    with open(fname) as f:
        tok = f.read().split()
    pos = 0

    def take(k, conv):
        nonlocal pos
        out = [conv(t) for t in tok[pos:pos + k]]
        pos += k
        return out
    tr = LineTrack()
    v = take(6, float)
    tr.line = Line3d(v[:3], v[3:])
    n, _n_images = take(2, int)

    def expect(word):
        nonlocal pos
        if tok[pos] != word:
            raise ValueError(f"{fname}: expected '{word}', found '{tok[pos]}'")
        pos += 1
    expect("image_id_list")
    tr.image_id_list = take(n, int)
    expect("line_id_list")
    tr.line_id_list = take(n, int)
    if pos >= len(tok) or tok[pos] != "line2d_list":
        return tr
    pos += 1
    for _ in range(n):
        s = take(4, float)
        tr.line2d_list.append(Line2d(s[:2], s[2:]))
    if tok[pos] == "END":
        return tr
    expect("node_id_list")
    tr.node_id_list = take(n, int)
    expect("score_list")
    tr.score_list = take(n, float)
    if tok[pos] == "line3d_list":
        pos += 1
        for _ in range(n):
            s = take(6, float)
            tr.line3d_list.append(Line3d(s[:3], s[3:]))
    return tr


def save_folder_linetracks(folder, tracks):
    os.makedirs(folder, exist_ok=True)
    for old in os.listdir(folder):
        if old.startswith("track_") and old.endswith(".txt"):
            os.remove(os.path.join(folder, old))
    for i, tr in enumerate(tracks):
        write_track(os.path.join(folder, f"track_{i}.txt"), tr)


def read_folder_linetracks(folder):
    n = sum(1 for f in os.listdir(folder) if f.startswith("track") and f.endswith(".txt"))
    return [read_track(os.path.join(folder, f"track_{i}.txt")) for i in range(n)]


_ALLTRACKS_SEP = " " + " " * 18


def save_txt_linetracks(fname, tracks, n_visible_views=4):
    """alltracks.txt (util/io.py:259-292): only tracks seen in >= n_visible_views images."""
    os.makedirs(os.path.dirname(os.path.abspath(fname)), exist_ok=True)
    keep = [t for t in tracks if t.count_images() >= n_visible_views]
    with open(fname, "w") as f:
        f.write(f"{len(keep)}\n")
        for i, t in enumerate(keep):
            f.write(f"{i} {t.count_lines()} {t.count_images()}\n")
            # the reference's f-strings continue over source lines with a backslash: the three coordinates are
            # separated by one blank plus the 18 blanks of the next line's indentation (util/io.py:277-286)
            f.write(_ALLTRACKS_SEP.join(f"{v:.10f}" for v in t.line.start) + "\n")
            f.write(_ALLTRACKS_SEP.join(f"{v:.10f}" for v in t.line.end) + "\n")
            f.write("".join(f"{v} " for v in t.image_id_list) + "\n")
            f.write("".join(f"{v} " for v in t.line_id_list) + "\n")


def save_folder_linetracks_with_info(folder, linetracks, config=None, imagecols=None, all_2d_segs=None):
    """util/io.py:323-333."""
    save_folder_linetracks(folder, linetracks)
    if config is not None:
        save_npy(os.path.join(folder, "config.npy"), config)
    if imagecols is not None:
        save_npy(os.path.join(folder, "imagecols.npy"), imagecols.as_dict())
    if all_2d_segs is not None:
        save_npy(os.path.join(folder, "all_2d_segs.npy"), all_2d_segs)


def save_txt_imname_dict(fname, imname_dict):
    """util/io.py:157-163: `number of images, N` then one `img_id, name` row per image."""
    os.makedirs(os.path.dirname(os.path.abspath(fname)), exist_ok=True)
    with open(fname, "w") as f:
        f.write(f"number of images, {len(imname_dict)}\n")
        for img_id, name in imname_dict.items():
            f.write(f"{img_id}, {name}\n")


def save_obj(fname, lines):
    """util/io.py:181-199: every 3D segment as two vertices and one `l` element."""
    if isinstance(lines, list) and len(lines) > 0:
        # a list of (2, 3) arrays or of Line3d objects (anything with as_array()), as the reference accepts
        if isinstance(lines[0], np.ndarray):
            lines = np.array(lines)
        else:
            lines = np.array([line.as_array() for line in lines])
    n = 0 if lines is None or len(lines) == 0 else lines.shape[0]
    os.makedirs(os.path.dirname(os.path.abspath(fname)), exist_ok=True)
    with open(fname, "w") as f:
        for k in range(n):
            for p in lines[k]:
                f.write(f"v {p[0]} {p[1]} {p[2]}\n")
        for k in range(n):
            f.write(f"l {2 * k + 1} {2 * k + 2}\n")


# ---- a scene folder as written by limap.runners.line_triangulation ---------------------------
def triangulate_scene_folder(imagecols_npy, metainfos_txt, segments_folder, matches_folder, cfg, device=0,
                             exhaustive=False):
    """Run the MI355X triangulator on the intermediate artefacts of a limap run
    (runners/line_triangulation.py:56-97 writes them): returns (triangulator, tracks)."""
    from . import triangulation as tri
    imagecols = read_imagecols(imagecols_npy)
    neighbors, ranges = read_txt_metainfos(metainfos_txt)
    all_2d_segs = {i: read_txt_segments(segments_folder, i) for i in imagecols.get_img_ids()}
    T = tri.GlobalLineTriangulator(cfg, device=device)
    T.SetRanges(ranges)
    T.Init(all_2d_segs, imagecols)
    if exhaustive:
        for i in imagecols.get_img_ids():
            T.TriangulateImageExhaustiveMatch(i, neighbors[i])
    else:  # the whole loop of runners/line_triangulation.py:160-167 in one native call
        T.TriangulateAll({i: read_matches(matches_folder, i) for i in imagecols.get_img_ids()})
    return T, T.ComputeLineTracks()
