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
