"""oracle/_ref: the REFERENCE ITSELF as a checker.

`oracle/_ref/liblimap_ref.so` holds the unmodified hot-path sources of /root/reference/src/limap (list:
oracle/Makefile, REF_SRCS) compiled where they lie against the stand-in headers of oracle/ref_shim/ (Eigen,
COLMAP and PoseLib are not on disk) plus the C entry points of oracle/ref_driver.cpp, which mirror the oracle's
(`ora_*` -> `ref_*`).  `module()` returns a second instance of oracle/oracle.py bound to that library, so every
wrapper class and free function of the oracle is available with the reference underneath:

    ref = oracle.ref.module()
    R = ref.OracleTriangulator(cfg)        # limap::triangulation::GlobalLineTriangulator inside
    ref.compute_epipolar_IoU(seg1, cam1, seg2, cam2)

TEST INFRASTRUCTURE ONLY (tests/test_oracle_vs_ref.py pins the oracle with it; bench.py may time it as the
`cpu_baseline` of kind "reference").  /root/reference does not exist on the GPU box: there only a prebuilt
library (it travels with the snapshot) can be loaded -- nothing here reads the reference at run time.
"""
import ctypes as C
import importlib.util
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_ref", "liblimap_ref.so")
REFERENCE_SRC = "/root/reference/src/limap"

_mod = None


def build(force=False):
    """`make -C oracle ref` when the reference sources are present; returns the library path or None."""
    if os.path.isdir(REFERENCE_SRC):
        cmd = ["make", "-C", _HERE, "-j8", "ref"] + (["-B"] if force else [])
        res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if res.returncode != 0:
            raise RuntimeError("building oracle/_ref failed:\n" + res.stdout[-4000:])
    return LIB_PATH if os.path.exists(LIB_PATH) else None


def available():
    return os.path.exists(LIB_PATH) or os.path.isdir(REFERENCE_SRC)


def tree_source_hash():
    """SHA-256 (16 hex digits) over the tree files oracle/_ref is compiled against besides the reference's own sources:
    every stand-in header under ref_shim/, eigen_svd_ref.h, lt_oracle.h, ref_driver.cpp -- in sorted relative-path order,
    each as `path NUL bytes NUL`.  oracle/Makefile bakes it into the library (`ref_source_hash()`)."""
    import hashlib
    files = [os.path.join(_HERE, f) for f in ("eigen_svd_ref.h", "lt_oracle.h", "ref_driver.cpp")]
    for root, _, names in os.walk(os.path.join(_HERE, "ref_shim")):
        files += [os.path.join(root, n) for n in names]
    h = hashlib.sha256()
    for f in sorted(files, key=lambda f: os.path.relpath(f, _HERE)):
        h.update(os.path.relpath(f, _HERE).encode() + b"\0")
        with open(f, "rb") as fh:
            h.update(fh.read() + b"\0")
    return h.hexdigest()[:16]


def library_source_hash():
    """The hash baked into the built oracle/_ref library, or None when it is absent / predates the hash."""
    if not os.path.exists(LIB_PATH):
        return None
    dll = C.PyDLL(LIB_PATH)
    try:
        fn = dll.ref_source_hash
    except AttributeError:
        return None
    fn.restype = C.c_char_p
    return fn().decode()


class _Renamed:
    """ora_* names resolved as ref_* in the reference-backed library."""

    def __init__(self, dll):
        self._dll = dll

    def __getattr__(self, name):
        if not name.startswith("ora_"):
            raise AttributeError(name)
        fn = getattr(self._dll, "ref_" + name[4:])
        setattr(self, name, fn)
        return fn


def module():
    """oracle/oracle.py instantiated a second time over oracle/_ref (built first if the sources are here)."""
    global _mod
    if _mod is None:
        path = build()
        if path is None:
            raise RuntimeError("oracle/_ref is not built and /root/reference is not present")
        spec = importlib.util.spec_from_file_location("oracle._ref_binding", os.path.join(_HERE, "oracle.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        # PyDLL: the reference's constructors take py::dict -- the GIL stays held during the calls
        L = mod._prototype(_Renamed(C.PyDLL(path)))
        L.ora_set_num_threads(min(os.cpu_count() or 1, 16))
        mod._lib = L
        mod.KIND = "reference"
        _mod = mod
    return _mod


# ---- file formats through the reference's own code (tests/test_io_formats.py, tests/golden/make_io_golden.py) -------
_raw = None


def _dll():
    """The reference-backed library itself (entry points that have no ora_* counterpart)."""
    global _raw
    if _raw is None:
        path = build()
        if path is None:
            raise RuntimeError("oracle/_ref is not built and /root/reference is not present")
        _raw = C.PyDLL(path)
        _raw.ref_imagecols_as_dict.restype = C.py_object
        _raw.ref_imagecols_from_dict.argtypes = [C.py_object, C.c_int] + [C.c_void_p] * 4
        _raw.ref_track_write.restype = C.c_int
        _raw.ref_track_read.restype = C.c_int
    return _raw


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def track_write(fname, line6, image_ids, line_ids, line2d, node_ids=None, scores=None, line3d=None):
    """limap::LineTrack::Write (base/linetrack.cc:133-209) on a track given as arrays."""
    import numpy as np
    n = len(image_ids)
    line6 = np.ascontiguousarray(line6, np.float64)
    img = np.ascontiguousarray(image_ids, np.int32); lid = np.ascontiguousarray(line_ids, np.int32)
    l2 = np.ascontiguousarray(line2d, np.float64).reshape(n, 4)
    flags = (1 if node_ids is not None else 0) | (2 if scores is not None else 0) | (4 if line3d is not None else 0)
    nid = np.ascontiguousarray(node_ids if node_ids is not None else np.zeros(n), np.int32)
    sc = np.ascontiguousarray(scores if scores is not None else np.zeros(n), np.float64)
    l3 = np.ascontiguousarray(line3d if line3d is not None else np.zeros((n, 6)), np.float64).reshape(n, 6)
    rc = _dll().ref_track_write(os.fsencode(fname), _ptr(line6), C.c_int(n), _ptr(img), _ptr(lid), _ptr(l2), C.c_int(flags),
                                _ptr(nid), _ptr(sc), _ptr(l3))
    if rc != 0:
        raise RuntimeError("LineTrack::Write failed")


def track_read(fname, cap=1 << 16):
    """limap::LineTrack::Read (base/linetrack.cc:211-270): dict of arrays (aux lists zero where the file has none)."""
    import numpy as np
    line6 = np.zeros(6); img = np.zeros(cap, np.int32); lid = np.zeros(cap, np.int32); l2 = np.zeros((cap, 4))
    nid = np.zeros(cap, np.int32); sc = np.zeros(cap); l3 = np.zeros((cap, 6))
    n = _dll().ref_track_read(os.fsencode(fname), _ptr(line6), C.c_int(cap), _ptr(img), _ptr(lid), _ptr(l2), _ptr(nid),
                              _ptr(sc), _ptr(l3))
    if n < 0:
        raise RuntimeError("LineTrack::Read failed (%d)" % n)
    return dict(line=line6, image_ids=img[:n].copy(), line_ids=lid[:n].copy(), line2d=l2[:n].copy(), node_ids=nid[:n].copy(),
                scores=sc[:n].copy(), line3d=l3[:n].copy())


def imagecols_as_dict(img_ids, kvec, qvec, tvec):
    """limap::ImageCollection::as_dict() (base/image_collection.cc:158-171) for PINHOLE cameras, one per image."""
    import numpy as np
    ids = np.ascontiguousarray(img_ids, np.int32)
    k = np.ascontiguousarray(kvec, np.float64); q = np.ascontiguousarray(qvec, np.float64)
    t = np.ascontiguousarray(tvec, np.float64)
    return _dll().ref_imagecols_as_dict(C.c_int(len(ids)), _ptr(ids), _ptr(k), _ptr(q), _ptr(t))


def imagecols_from_dict(d):
    """limap::ImageCollection(py::dict) -> (img_ids, kvec, qvec, tvec) in ascending image id."""
    import numpy as np
    cap = len(d["images"]) + 1
    ids = np.zeros(cap, np.int32); k = np.zeros((cap, 4)); q = np.zeros((cap, 4)); t = np.zeros((cap, 3))
    n = _dll().ref_imagecols_from_dict(d, C.c_int(cap), _ptr(ids), _ptr(k), _ptr(q), _ptr(t))
    if n < 0:
        raise RuntimeError("ImageCollection(dict) failed (%d)" % n)
    return ids[:n], k[:n], q[:n], t[:n]


if __name__ == "__main__":
    import sys
    if "--hash" in sys.argv:
        print(tree_source_hash())
