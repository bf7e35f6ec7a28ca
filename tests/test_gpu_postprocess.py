"""-m gpu: the steps after ComputeLineTracks (limap.merging.filter_tracks_by_reprojection, remerge,
filter_tracks_by_sensitivity, filter_tracks_by_overlap; runners/line_triangulation.py:171-200) --
HIP/host backend through the C ABI vs the CPU oracle, in the order and with the thresholds of
cfgs/triangulation/default.yaml:102-115."""
import numpy as np
import pytest

from limap_amd import synthetic as syn

from helpers import run_oracle, run_product

pytestmark = pytest.mark.gpu

REMERGE_LINKER = dict(score_th=0.5, th_angle=5.0, th_overlap=0.001, th_smartoverlap=0.1, th_smartangle=1.0,
                      th_perp=1.0, th_innerseg=1.0)
F2D = dict(th_angular_2d=8.0, th_perp_2d=5.0, th_sv_angular_3d=75.0, th_sv_num_supports=3, th_overlap=0.5,
           th_overlap_num_supports=3)


def compare_sets(g, o, stage):
    assert np.array_equal(g["off"], o["off"]), f"{stage}: track sizes differ"
    for k in ("image_ids", "line_ids", "node_ids"):
        assert np.array_equal(g[k], o[k]), f"{stage}: {k} differ"
    assert np.array_equal(g["active"], o["active"]), f"{stage}: active flags differ"
    np.testing.assert_allclose(g["scores"], o["scores"], rtol=1e-12)
    assert np.array_equal(g["line2d"], o["line2d"]) and np.array_equal(g["line3d"][:, :9], o["line3d"][:, :9])
    gl, ol = g["line"], o["line"]
    if len(ol):
        scale = np.maximum(np.abs(ol[:, :6]).max(1), 1e-9)
        err = np.abs(gl[:, :6] - ol[:, :6]).max(1) / scale  # same orientation: no start / end swap is accepted
        assert err.max() <= 1e-5, f"{stage}: track line differs {err.max()}"
        np.testing.assert_allclose(gl[:, 6], ol[:, 6], rtol=1e-12)


@pytest.mark.parametrize("seed,views,segs,nn", [(0, 30, 200, 10), (3, 20, 150, 8)])
def test_postprocess_chain(gpu_lib, oracle, seed, views, segs, nn):
    from limap_amd import merging
    sc = syn.make_scene(n_views=views, n_segs=segs, n_neighbors=nn, seed=seed)
    cfg = syn.default_triangulation_cfg()
    T = run_product(sc, cfg)
    O = run_oracle(oracle, sc, cfg)
    T.ComputeLineTracks()
    O.ComputeLineTracks()
    gs = merging.TrackSet.from_triangulator(T)
    os_ = oracle.OracleTrackSet(O)
    assert len(gs) == os_.num_tracks() > 50
    compare_sets(gs.arrays(), os_.get(), "initial")
    gs.filter_by_reprojection(F2D["th_angular_2d"], F2D["th_perp_2d"]); os_.filter_by_reprojection(F2D["th_angular_2d"], F2D["th_perp_2d"])
    compare_sets(gs.arrays(), os_.get(), "reprojection")
    gs.remerge(REMERGE_LINKER); os_.remerge(REMERGE_LINKER)
    compare_sets(gs.arrays(), os_.get(), "remerge")
    gs.filter_by_reprojection(F2D["th_angular_2d"], F2D["th_perp_2d"]); os_.filter_by_reprojection(F2D["th_angular_2d"], F2D["th_perp_2d"])
    compare_sets(gs.arrays(), os_.get(), "reprojection 2")
    gs.filter_by_sensitivity(F2D["th_sv_angular_3d"], F2D["th_sv_num_supports"]); os_.filter_by_sensitivity(F2D["th_sv_angular_3d"], F2D["th_sv_num_supports"])
    compare_sets(gs.arrays(), os_.get(), "sensitivity")
    gs.filter_by_overlap(F2D["th_overlap"], F2D["th_overlap_num_supports"]); os_.filter_by_overlap(F2D["th_overlap"], F2D["th_overlap_num_supports"])
    compare_sets(gs.arrays(), os_.get(), "overlap")
    assert 0 < len(gs) < os_.get()["off"].shape[0] + 1


def test_module_level_functions_match_trackset(gpu_lib):
    """limap.merging-style functions on LineTrack lists give the same tracks as the bound TrackSet."""
    from limap_amd import base, merging
    sc = syn.make_scene(n_views=16, n_segs=120, n_neighbors=8, seed=1)
    cfg = syn.default_triangulation_cfg()
    T = run_product(sc, cfg)
    tracks = T.ComputeLineTracks()
    ic = base.ImageCollection.from_arrays(sc.img_ids, sc.kvec, sc.qvec, sc.tvec)
    a = merging.TrackSet.from_triangulator(T).filter_by_reprojection(8.0, 5.0).remerge(REMERGE_LINKER).arrays()
    # LineTrack lists drop nothing we need except the per-support 3D uncertainties (as_array has no slot
    # for them): feed the TrackSet's own LineTracks instead of the dict round trip
    lst = merging.TrackSet.from_triangulator(T).tracks()
    lst = merging.filter_tracks_by_reprojection(lst, ic, 8.0, 5.0)
    lst = merging.remerge(REMERGE_LINKER, lst)
    assert len(lst) == len(a["off"]) - 1
    assert [t.image_id_list for t in lst] == [a["image_ids"][a["off"][n]:a["off"][n + 1]].tolist() for n in range(len(lst))]
    lst2 = merging.filter_tracks_by_overlap(merging.filter_tracks_by_sensitivity(lst, ic, 75.0, 3), ic, 0.5, 3)
    assert len(lst2) <= len(lst) and len(tracks) > 0
