"""Light stand-ins for the `limap.base` value types the triangulation boundary exchanges.

When a real limap install is present the triangulator accepts and returns limap's own objects
(duck-typed, see ``triangulation.py``); these classes carry the same attribute / method names
(reference: src/limap/base/linebase.h:16-60, base/linetrack.h:21-50, base/camera_view.h:56-88,
base/image_collection.h:24-115) so the same caller code runs without limap.
"""
import numpy as np


class Line2d:
    """linebase.h:16-35"""

    def __init__(self, start=None, end=None, score=-1.0):
        if start is not None and end is None:
            a = np.asarray(start, float)
            if a.shape == (2, 2):
                start, end = a[0], a[1]
            else:
                a = a.reshape(-1)
                start, end = a[0:2], a[2:4]
        self.start = np.zeros(2) if start is None else np.asarray(start, float).copy()
        self.end = np.zeros(2) if end is None else np.asarray(end, float).copy()
        self.score = float(score)

    def length(self):
        return float(np.linalg.norm(self.start - self.end))

    def midpoint(self):
        return 0.5 * (self.start + self.end)

    def direction(self):
        d = self.end - self.start
        n = np.linalg.norm(d)
        return d / n if n > 0 else d

    def as_array(self):
        return np.stack([self.start, self.end], 0)


class Line3d:
    """linebase.h:37-60"""

    def __init__(self, start=None, end=None, score=-1.0, depth_start=-1.0, depth_end=-1.0, uncertainty=-1.0):
        if start is not None and end is None:
            a = np.asarray(start, float).reshape(2, 3)
            start, end = a[0], a[1]
        self.start = np.zeros(3) if start is None else np.asarray(start, float).copy()
        self.end = np.zeros(3) if end is None else np.asarray(end, float).copy()
        self.score = float(score)
        self.uncertainty = float(uncertainty)
        self.depths = np.array([depth_start, depth_end], float)

    @classmethod
    def from10(cls, a):
        """line10 layout of the C ABI: start3, end3, depths2, uncertainty, score."""
        return cls(a[0:3], a[3:6], a[9], a[6], a[7], a[8])

    def set_uncertainty(self, val):
        self.uncertainty = float(val)

    def length(self):
        return float(np.linalg.norm(self.start - self.end))

    def midpoint(self):
        return 0.5 * (self.start + self.end)

    def direction(self):
        d = self.end - self.start
        n = np.linalg.norm(d)
        return d / n if n > 0 else d

    def as_array(self):
        return np.stack([self.start, self.end], 0)


class LineTrack:
    """linetrack.h:21-50 (fields filled by GlobalLineTriangulator::build_tracks_from_clusters)."""

    def __init__(self, line=None, image_id_list=None, line_id_list=None, line2d_list=None):
        self.line = line if line is not None else Line3d()
        self.image_id_list = list(image_id_list or [])
        self.line_id_list = list(line_id_list or [])
        self.line2d_list = list(line2d_list or [])
        self.node_id_list = []
        self.line3d_list = []
        self.score_list = []
        self.active = True

    def count_lines(self):
        return len(self.line2d_list)

    def GetSortedImageIds(self):
        return sorted(set(self.image_id_list))

    def count_images(self):
        return len(self.GetSortedImageIds())

    def HasImage(self, image_id):
        return image_id in self.image_id_list

    def as_dict(self):  # linetrack.cc:31-48
        return dict(line=self.line.as_array(), image_id_list=list(self.image_id_list),
                    line_id_list=list(self.line_id_list), node_id_list=list(self.node_id_list),
                    score_list=list(self.score_list), line2d_list=[l.as_array() for l in self.line2d_list],
                    line3d_list=[l.as_array() for l in self.line3d_list], active=self.active)

    def Write(self, filename):  # linetrack.cc:133-209
        from . import io as _io
        _io.write_track(filename, self)

    def Read(self, filename):  # linetrack.cc:211-270
        from . import io as _io
        other = _io.read_track(filename)
        self.__dict__.update(other.__dict__)


class CameraView:
    """camera_view.h:56-88 reduced to the undistorted pinhole the hot path requires
    (base_line_triangulator.cc:49)."""

    def __init__(self, kvec, qvec, tvec, image_name="none"):
        self.kvec = np.asarray(kvec, float).reshape(4)
        q = np.asarray(qvec, float).reshape(4)
        n = np.linalg.norm(q)
        self.qvec = q / n if n > 0 else q
        self.tvec = np.asarray(tvec, float).reshape(3)
        self._name = image_name

    def image_name(self):
        return self._name

    def K(self):
        fx, fy, cx, cy = self.kvec
        return np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1.0]])

    def R(self):
        w, x, y, z = self.qvec / np.linalg.norm(self.qvec)
        return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                         [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                         [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])

    def T(self):
        return self.tvec

    def as_cam11(self):
        return np.concatenate([self.kvec, self.qvec, self.tvec])


class ImageCollection:
    """image_collection.h:24-115 reduced to ids + per-image pinhole views."""

    def __init__(self, views=None):
        self._views = dict(views or {})

    @classmethod
    def from_arrays(cls, img_ids, kvec, qvec, tvec):
        return cls({int(i): CameraView(kvec[k], qvec[k], tvec[k]) for k, i in enumerate(img_ids)})

    def get_img_ids(self):
        return sorted(self._views.keys())

    def NumImages(self):
        return len(self._views)

    def exist_image(self, img_id):
        return img_id in self._views

    def camview(self, img_id):
        return self._views[img_id]

    def IsUndistorted(self):
        return True

    def get_image_name_dict(self):  # image_collection.cc: img_id -> image name
        return {i: self._views[i].image_name() for i in self.get_img_ids()}

    def as_dict(self):  # image_collection.cc:158-171
        from . import io as _io
        return _io.imagecols_to_dict(self)

    def set_max_image_dim(self, val):
        """image_collection.cc:399-403 / camera.cc:216-226: cameras larger than `val` are rescaled.  The views here carry
        no image size (height = width = 0, like a reference Camera built without one): max(h, w) = 0 makes the ratio
        infinite, nothing is rescaled -- the same as the reference does for such cameras."""
        if val <= 0:
            raise ValueError("Check failed: val > 0")

    def update_neighbors(self, neighbors):  # image_collection.cc:322-341
        if len(neighbors) == self.NumImages():
            return neighbors
        out = {}
        for img_id in self.get_img_ids():
            if img_id not in neighbors:
                raise RuntimeError("Error! The image id is not found in the input neighbors.")
            out[img_id] = [n for n in neighbors[img_id] if self.exist_image(n)]
        return out

    def get_map_camviews(self):
        return {i: self._views[i] for i in self.get_img_ids()}


def get_all_lines_2d(all_2d_segs):
    """base/functions.py:4-24: dict img_id -> (N, 4+) array  ->  dict img_id -> list of Line2d."""
    return {int(i): [Line2d(s[0:2], s[2:4]) for s in np.asarray(segs, float).reshape(-1, np.asarray(segs).shape[-1] if np.asarray(segs).ndim == 2 else 4)]
            for i, segs in all_2d_segs.items()}


class LineLinker3d:
    """base/line_linker.h:168-185 as far as the path needs it: a holder of the 3D linker configuration (dict of
    cfg["triangulation"]["remerging"]["linker3d"]); `limap_amd.merging.remerge` reads the fields."""

    def __init__(self, cfg=None):
        from types import SimpleNamespace
        self.config = SimpleNamespace(**dict(cfg or {}))


def track_report(off, image_ids):
    """The numbers of limap's track report (visualize/trackvis/base.py:25-50) from a CSR track set (`off`,
    `image_ids` as returned by Context.get_tracks()): tracks with >= 2, 4, 6, 8, 10, 20, 50 supporting IMAGES, and the
    average number of supporting images / lines over the tracks seen from >= 3 and >= 4 images."""
    import numpy as np
    off = np.asarray(off, np.int64)
    image_ids = np.asarray(image_ids)
    n = len(off) - 1
    counts = np.array([len(np.unique(image_ids[off[t]:off[t + 1]])) for t in range(n)], np.int64)
    lines = np.diff(off)
    rep = {f"N{k}": int((counts >= k).sum()) for k in (2, 4, 6, 8, 10, 20, 50)}
    for k in (3, 4):
        m = counts >= k
        rep[f"avg_supporting_images_ge{k}"] = float(counts[m].mean()) if m.any() else 0.0
        rep[f"avg_supporting_lines_ge{k}"] = float(lines[m].mean()) if m.any() else 0.0
    return rep

