"""The pybind11 shim (limap_amd/_lt_pybind, csrc/lt_pybind.cpp): importable, exposes the reference's method names
(triangulation/bindings.cc:78-95), and -- on a GPU -- gives the results of the ctypes path."""
import numpy as np
import pytest

from limap_amd import synthetic as syn


def test_pybind_module_surface():
    from limap_amd import _lt_pybind as pb
    from limap_amd import _capi
    assert pb.abi_version() == _capi.load_library().lt_abi_version()
    names = set(dir(pb.GlobalLineTriangulator))
    for n in ("SetRanges", "UnsetRanges", "TriangulateImage", "TriangulateImageExhaustiveMatch", "ComputeLineTracks",
              "GetTracks", "CountImages", "CountLines", "InitArrays"):
        assert n in names, n
    with pytest.raises(Exception):  # no GPU here / no such handle: construction must not silently succeed
        pb.GlobalLineTriangulator(0)


@pytest.mark.gpu
def test_pybind_class_end_to_end(gpu_lib, oracle):
    """The pybind class on its own (owning context, array-level Init) against the oracle."""
    from limap_amd import _lt_pybind as pb
    from helpers import compare_tracks, compare_best, run_oracle, small_scene
    sc = small_scene(seed=14, n_views=12, n_segs=90, n_neighbors=6)
    cfg = syn.default_triangulation_cfg()
    T = pb.GlobalLineTriangulator(cfg, 0)
    T.SetRanges((sc.ranges[0], sc.ranges[1]))
    T.InitArrays(sc.img_ids, sc.kvec, sc.qvec, sc.tvec, sc.seg_off, sc.segs)
    assert T.CountImages() == sc.n_images and T.CountLines(int(sc.img_ids[2])) == sc.seg_off[3] - sc.seg_off[2]
    for i in sc.img_ids:
        m = sc.matches_of(int(i))
        if int(i) % 3 == 0:  # other integer dtypes / non-contiguous arrays are converted by copy
            m = {k: np.asarray(v, np.int64) for k, v in m.items()}
        T.TriangulateImage(int(i), m)
    tracks = T.ComputeLineTracks()
    O = run_oracle(oracle, sc, cfg)
    compare_tracks(tracks, O.ComputeLineTracks())
    compare_best(T.GetBest(), O.get_best())
    assert T.Stats()["tracks"] == O.stats()["tracks"]
    with pytest.raises(RuntimeError, match="IndexError"):
        T2 = pb.GlobalLineTriangulator(cfg, 0)
        T2.InitArrays(sc.img_ids, sc.kvec, sc.qvec, sc.tvec, sc.seg_off, sc.segs)
        T2.TriangulateImage(int(sc.img_ids[0]), {int(sc.img_ids[1]): np.array([[100000, 0]], np.int32)})
    with pytest.raises(ValueError):
        T.TriangulateImage(int(sc.img_ids[0]), {int(sc.img_ids[1]): np.zeros((3, 3), np.int32)})
    with pytest.raises(IndexError):
        T.CountLines(987654)
