"""On-disk formats: round trips and agreement with the reference's layout
(util/io.py:87-131,259-292,441-465; base/linetrack.cc:133-260; line2d/base_matcher.py:77-115)."""
import os

import numpy as np
import pytest

from limap_amd import base, io as ltio, synthetic as syn


def test_metainfos_roundtrip_and_layout(tmp_path):
    nb = {0: [3, 1], 7: [], 3: [0]}
    rng = (np.array([-1.5, 0.25, 1e-3]), np.array([2.0, 3.5, 9.0]))
    f = tmp_path / "metainfos.txt"
    ltio.save_txt_metainfos(str(f), nb, rng)
    rows = f.read_text().splitlines()
    assert rows[0] == "number of images, 3" and rows[1] == "x-range, -1.5, 2.0" and rows[4] == "image 0, 3, 1"
    assert rows[5] == "image 7"
    nb2, rng2 = ltio.read_txt_metainfos(str(f))
    assert nb2 == nb and np.array_equal(rng2[0], rng[0]) and np.array_equal(rng2[1], rng[1])


def test_segments_roundtrip_is_exact(tmp_path):
    segs = np.random.default_rng(0).uniform(0, 800, (37, 4))
    ltio.save_txt_segments(str(tmp_path), 12, segs)
    rows = (tmp_path / "segments_12.txt").read_text().splitlines()
    assert rows[0] == "37" and len(rows) == 38
    back = ltio.read_txt_segments(str(tmp_path), 12)
    assert np.array_equal(back, segs), "repr() of a float64 round-trips exactly"
    ltio.save_txt_segments(str(tmp_path), 13, np.zeros((0, 4)))
    assert ltio.read_txt_segments(str(tmp_path), 13).shape == (0, 4)
    assert sorted(ltio.read_all_segments_from_folder(str(tmp_path))) == [12, 13]


def test_matches_npy_is_a_pickled_dict(tmp_path):
    m = {4: np.array([[0, 1], [0, 5], [2, 2]], np.int32), 9: np.zeros((0, 2), np.int32)}
    ltio.save_matches(str(tmp_path), 3, m)
    raw = np.load(tmp_path / "matches_3.npy", allow_pickle=True)
    assert raw.dtype == object and raw.shape == ()  # what limapio.read_npy(...).item() expects
    back = ltio.read_matches(str(tmp_path), 3)
    assert sorted(back) == [4, 9] and np.array_equal(back[4], m[4]) and back[9].shape == (0, 2)


def test_imagecols_dict_roundtrip(tmp_path):
    sc = syn.make_scene(n_views=4, n_segs=5, n_neighbors=2, seed=1)
    ic = base.ImageCollection.from_arrays(sc.img_ids, sc.kvec, sc.qvec, sc.tvec)
    f = tmp_path / "imagecols.npy"
    ltio.save_imagecols(str(f), ic)
    d = ltio.read_npy(str(f)).item()
    assert set(d) == {"cameras", "images"} and set(d["images"][0]) == {"cam_id", "pose", "image_name"}
    ic2 = ltio.read_imagecols(str(f))
    for i in ic.get_img_ids():
        assert np.array_equal(ic2.camview(i).kvec, ic.camview(i).kvec)
        np.testing.assert_allclose(ic2.camview(i).qvec, ic.camview(i).qvec, atol=1e-16)
        assert np.array_equal(ic2.camview(i).tvec, ic.camview(i).tvec)
    # SIMPLE_PINHOLE params (f, cx, cy) and a distorted camera
    assert np.array_equal(ltio.kvec_from_camera_dict(dict(model_id=0, params=[500.0, 320, 240])), [500, 500, 320, 240])
    assert np.array_equal(ltio.kvec_from_camera_dict(dict(model_id=2, params=[500.0, 320, 240, 0.0])), [500, 500, 320, 240])
    with pytest.raises(ValueError, match="IsUndistorted"):
        ltio.kvec_from_camera_dict(dict(model_id=2, params=[500.0, 320, 240, 0.1]))


def _track():
    tr = base.LineTrack()
    tr.line = base.Line3d([0.1, 0.2, 0.3], [1.0, 2.0, 3.0])
    tr.image_id_list, tr.line_id_list, tr.node_id_list = [2, 5, 5], [10, 3, 4], [0, 1, 2]
    tr.score_list = [1.5, 2.25, 0.0]
    tr.line2d_list = [base.Line2d([1, 2], [3, 4]), base.Line2d([5, 6], [7, 8]), base.Line2d([9, 10], [11, 12.5])]
    tr.line3d_list = [base.Line3d([0, 0, 0], [1, 1, 1]), base.Line3d([0, 0, 1], [1, 1, 2]), base.Line3d([0, 1, 0], [2, 1, 1])]
    return tr


def test_track_file_layout_and_roundtrip(tmp_path):
    tr = _track()
    f = tmp_path / "track_0.txt"
    ltio.write_track(str(f), tr)
    rows = f.read_text().splitlines()
    assert rows[0].split() == ["0.1000000000", "0.2000000000", "0.3000000000", "1.0000000000", "2.0000000000", "3.0000000000"]
    assert rows[1] == "3 2" and rows[2].split() == ["image_id_list", "2", "5", "5"] and rows[4] == "line2d_list"
    assert rows[8].split()[0] == "node_id_list" and rows[9].split()[0] == "score_list" and rows[10] == "line3d_list"
    assert rows[-1] == "END"
    back = ltio.read_track(str(f))
    assert back.image_id_list == tr.image_id_list and back.line_id_list == tr.line_id_list
    assert back.node_id_list == tr.node_id_list and back.score_list == tr.score_list
    np.testing.assert_allclose(back.line.as_array(), tr.line.as_array(), atol=1e-10)
    np.testing.assert_allclose(back.line2d_list[2].as_array(), tr.line2d_list[2].as_array())
    np.testing.assert_allclose(back.line3d_list[1].as_array(), tr.line3d_list[1].as_array())
    ltio.save_folder_linetracks(str(tmp_path / "finaltracks"), [tr, tr])
    assert len(ltio.read_folder_linetracks(str(tmp_path / "finaltracks"))) == 2


def test_alltracks_filters_by_visible_views(tmp_path):
    tr = _track()
    f = tmp_path / "alltracks.txt"
    ltio.save_txt_linetracks(str(f), [tr], n_visible_views=4)
    assert f.read_text().splitlines()[0] == "0"
    ltio.save_txt_linetracks(str(f), [tr], n_visible_views=2)
    rows = f.read_text().splitlines()
    assert rows[0] == "1" and rows[1] == "0 3 2" and rows[4].split() == ["2", "5", "5"]


@pytest.mark.gpu
def test_scene_folder_end_to_end(gpu_lib, oracle, tmp_path):
    """Write the artefacts of a limap run for a synthetic scene, triangulate from the folder, compare
    with the oracle fed from memory."""
    sc = syn.make_scene(n_views=10, n_segs=80, n_neighbors=5, seed=8)
    cfg = syn.default_triangulation_cfg()
    ic = base.ImageCollection.from_arrays(sc.img_ids, sc.kvec, sc.qvec, sc.tvec)
    ltio.save_imagecols(str(tmp_path / "imagecols.npy"), ic)
    ltio.save_txt_metainfos(str(tmp_path / "metainfos.txt"), sc.neighbors, sc.ranges)
    for n, i in enumerate(sc.img_ids):
        ltio.save_txt_segments(str(tmp_path / "segs"), int(i), sc.segs_of(n))
        ltio.save_matches(str(tmp_path / "matches"), int(i), sc.matches_of(int(i)))
    T, tracks = ltio.triangulate_scene_folder(str(tmp_path / "imagecols.npy"), str(tmp_path / "metainfos.txt"),
                                              str(tmp_path / "segs"), str(tmp_path / "matches"), cfg)
    from helpers import compare_tracks, run_oracle
    O = run_oracle(oracle, sc, cfg)
    # qvec passes through one extra numpy normalisation in ImageCollection: geometry agrees to 1e-9, indices exactly
    gt, ot = T.context().get_tracks(), O.ComputeLineTracks()
    assert np.array_equal(gt["off"], ot["off"]) and np.array_equal(gt["image_ids"], ot["image_ids"])
    assert np.array_equal(gt["line_ids"], ot["line_ids"])
    ltio.save_folder_linetracks(str(tmp_path / "finaltracks"), tracks)
    back = ltio.read_folder_linetracks(str(tmp_path / "finaltracks"))
    assert len(back) == len(tracks) and back[0].image_id_list == tracks[0].image_id_list
