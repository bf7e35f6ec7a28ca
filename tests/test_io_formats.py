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


# ---- against the reference's own writers / readers ---------------------------------------------
# tests/golden/io/ was written by /root/reference/src/limap/util/io.py (imported under stub modules) and by
# limap::LineTrack::Write / limap::ImageCollection::as_dict of oracle/_ref = the reference's sources compiled unmodified
# (tests/golden/make_io_golden.py).  limap_amd/io.py must write the same BYTES and read the same values.
import importlib.util
import sys

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "io")


def _golden_module():
    spec = importlib.util.spec_from_file_location("make_io_golden", os.path.join(os.path.dirname(GOLD), "make_io_golden.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _tracks_from(inputs):
    out = []
    for t in inputs["tracks"]:
        tr = base.LineTrack()
        tr.line = base.Line3d(t["line"][:3], t["line"][3:])
        tr.image_id_list = [int(i) for i in t["image_ids"]]
        tr.line_id_list = [int(i) for i in t["line_ids"]]
        tr.line2d_list = [base.Line2d(s[:2], s[2:]) for s in t["line2d"]]
        tr.node_id_list = [int(i) for i in t["node_ids"]]
        tr.score_list = [float(s) for s in t["scores"]]
        tr.line3d_list = [base.Line3d(s[:3], s[3:]) for s in t["line3d"]]
        out.append(tr)
    return out


def _bytes(path):
    with open(path, "rb") as f:
        return f.read()


def test_writers_produce_the_references_bytes(tmp_path):
    x = _golden_module().fixed_inputs()
    out = str(tmp_path)
    ltio.save_txt_metainfos(os.path.join(out, "metainfos.txt"), x["neighbors"], x["ranges"])
    ltio.save_txt_segments(out, 12, x["segs"])
    ltio.save_txt_segments(out, 13, np.zeros((0, 4)))
    ltio.save_matches(out, 3, x["matches"])
    tracks = _tracks_from(x)
    ltio.save_txt_linetracks(os.path.join(out, "alltracks_nv1.txt"), tracks, n_visible_views=1)
    ltio.save_txt_linetracks(os.path.join(out, "alltracks_nv4.txt"), tracks, n_visible_views=4)
    ltio.save_folder_linetracks(os.path.join(out, "finaltracks"), tracks)
    t0 = _tracks_from(x)[0]
    t0.node_id_list, t0.score_list, t0.line3d_list = [], [], []
    ltio.write_track(os.path.join(out, "track_noaux.txt"), t0)
    c = x["cams"]
    ltio.save_imagecols(os.path.join(out, "imagecols.npy"), base.ImageCollection.from_arrays(c["img_ids"], c["kvec"], c["qvec"], c["tvec"]))
    names = ["metainfos.txt", "segments_12.txt", "segments_13.txt", "alltracks_nv1.txt", "alltracks_nv4.txt",
             "track_noaux.txt"] + [os.path.join("finaltracks", f"track_{i}.txt") for i in range(3)]
    for n in names:
        assert _bytes(os.path.join(out, n)) == _bytes(os.path.join(GOLD, n)), f"{n}: bytes differ from the reference's file"
    # pickled objects: same structure, types and values (the pickle stream itself depends on numpy's version)
    got, ref = ltio.read_npy(os.path.join(out, "matches_3.npy")).item(), np.load(os.path.join(GOLD, "matches_3.npy"), allow_pickle=True).item()
    assert list(got) == list(ref)
    for k in ref:
        assert got[k].dtype == ref[k].dtype and np.array_equal(got[k], ref[k])
    got, ref = ltio.read_npy(os.path.join(out, "imagecols.npy")).item(), np.load(os.path.join(GOLD, "imagecols.npy"), allow_pickle=True).item()
    assert list(got) == list(ref) == ["cameras", "images"]
    assert list(got["cameras"]) == list(ref["cameras"]) and list(got["images"]) == list(ref["images"])
    for k in ref["cameras"]:
        assert list(got["cameras"][k]) == list(ref["cameras"][k])
        for f in ref["cameras"][k]:
            assert got["cameras"][k][f] == ref["cameras"][k][f], (k, f)
    for k in ref["images"]:
        g, r = got["images"][k], ref["images"][k]
        assert list(g) == list(r) and g["cam_id"] == r["cam_id"] and g["image_name"] == r["image_name"]
        assert list(g["pose"]) == list(r["pose"]) and g["pose"]["initialized"] == r["pose"]["initialized"]
        np.testing.assert_allclose(g["pose"]["qvec"], r["pose"]["qvec"], rtol=0, atol=1e-16)  # CameraPose normalises
        assert np.array_equal(g["pose"]["tvec"], r["pose"]["tvec"])


def test_readers_parse_the_references_files():
    x = _golden_module().fixed_inputs()
    nb, rng = ltio.read_txt_metainfos(os.path.join(GOLD, "metainfos.txt"))
    assert nb == x["neighbors"] and list(nb) == list(x["neighbors"])
    assert np.array_equal(rng[0], x["ranges"][0]) and np.array_equal(rng[1], x["ranges"][1])
    assert np.array_equal(ltio.read_txt_segments(GOLD, 12), x["segs"])
    assert ltio.read_txt_segments(GOLD, 13).shape == (0, 4)
    m = ltio.read_matches(GOLD, 3)
    assert sorted(m) == sorted(x["matches"]) and all(np.array_equal(m[k], x["matches"][k]) for k in m)
    tracks = ltio.read_folder_linetracks(os.path.join(GOLD, "finaltracks"))
    assert len(tracks) == 3
    for tr, t in zip(tracks, x["tracks"]):
        assert tr.image_id_list == [int(i) for i in t["image_ids"]] and tr.line_id_list == [int(i) for i in t["line_ids"]]
        assert tr.node_id_list == [int(i) for i in t["node_ids"]]
        want = np.where(np.isnan(t["line"]), 0.0, t["line"])
        np.testing.assert_allclose(tr.line.as_array().ravel(), want, rtol=0, atol=0.5e-10 + 1e-16)
        np.testing.assert_allclose(np.array([l.as_array().ravel() for l in tr.line2d_list]), t["line2d"], rtol=0, atol=0.51e-10)
        np.testing.assert_allclose(tr.score_list, t["scores"], rtol=0, atol=0.51e-10)
        np.testing.assert_allclose(np.array([l.as_array().ravel() for l in tr.line3d_list]), t["line3d"], rtol=0, atol=0.51e-10)
    t0 = ltio.read_track(os.path.join(GOLD, "track_noaux.txt"))
    assert t0.node_id_list == [] and t0.score_list == [] and t0.line3d_list == [] and len(t0.line2d_list) == 3
    ic = ltio.read_imagecols(os.path.join(GOLD, "imagecols.npy"))
    c = x["cams"]
    assert ic.get_img_ids() == [int(i) for i in c["img_ids"]]
    for n, i in enumerate(ic.get_img_ids()):
        assert np.array_equal(ic.camview(i).kvec, c["kvec"][n]) and np.array_equal(ic.camview(i).tvec, c["tvec"][n])
        np.testing.assert_allclose(ic.camview(i).qvec, c["qvec"][n] / np.linalg.norm(c["qvec"][n]), rtol=0, atol=1e-15)


def test_track_files_both_ways_through_the_reference(tmp_path):
    """limap::LineTrack::Read (oracle/_ref) on files written here, and read_track on files written by LineTrack::Write."""
    from oracle import ref as oref
    if not oref.available():
        pytest.skip("oracle/_ref is not built and /root/reference is not present")
    x = _golden_module().fixed_inputs()
    for n, (tr, t) in enumerate(zip(_tracks_from(x), x["tracks"])):
        mine, theirs = str(tmp_path / f"mine_{n}.txt"), str(tmp_path / f"theirs_{n}.txt")
        ltio.write_track(mine, tr)
        oref.track_write(theirs, t["line"], t["image_ids"], t["line_ids"], t["line2d"], t["node_ids"], t["scores"], t["line3d"])
        assert _bytes(mine) == _bytes(theirs)
        back = oref.track_read(mine)       # the reference reads our file
        again = ltio.read_track(theirs)    # we read the reference's file
        assert list(back["image_ids"]) == again.image_id_list == tr.image_id_list
        assert list(back["line_ids"]) == again.line_id_list and list(back["node_ids"]) == again.node_id_list
        assert np.array_equal(back["line"], again.line.as_array().ravel())
        assert np.array_equal(back["line2d"], np.array([l.as_array().ravel() for l in again.line2d_list]))
        assert np.array_equal(back["scores"], np.array(again.score_list))
        assert np.array_equal(back["line3d"], np.array([l.as_array().ravel() for l in again.line3d_list]))
    # ImageCollection: our dict through the reference's constructor, the reference's dict through ours
    c = x["cams"]
    ic = base.ImageCollection.from_arrays(c["img_ids"], c["kvec"], c["qvec"], c["tvec"])
    ids, k, q, t = oref.imagecols_from_dict(ltio.imagecols_to_dict(ic))
    assert np.array_equal(ids, c["img_ids"]) and np.array_equal(k, c["kvec"]) and np.array_equal(t, c["tvec"])
    np.testing.assert_allclose(q, c["qvec"] / np.linalg.norm(c["qvec"], axis=1, keepdims=True), rtol=0, atol=1e-15)
    ic2 = ltio.imagecols_from_dict(oref.imagecols_as_dict(c["img_ids"], c["kvec"], c["qvec"], c["tvec"]))
    assert ic2.get_img_ids() == [int(i) for i in c["img_ids"]]


def test_golden_io_files_are_what_the_reference_writes_today(tmp_path):
    """Where /root/reference exists (the build container): regenerate the golden files and compare with the committed ones."""
    mod = _golden_module()
    if not os.path.exists(mod.REF_IO):
        pytest.skip("/root/reference is not present")
    mod.write_all(str(tmp_path))
    for root, _, files in os.walk(GOLD):
        for f in files:
            rel = os.path.relpath(os.path.join(root, f), GOLD)
            if rel.endswith(".npy"):
                a = np.load(os.path.join(GOLD, rel), allow_pickle=True).item()
                b = np.load(os.path.join(str(tmp_path), rel), allow_pickle=True).item()
                assert repr(a) == repr(b), rel
            else:
                assert _bytes(os.path.join(GOLD, rel)) == _bytes(os.path.join(str(tmp_path), rel)), rel


def test_save_obj_accepts_arrays_lists_and_line3d_objects(tmp_path):
    """util/io.py:181-199: a (N, 2, 3) array, a list of (2, 3) arrays and a list of Line3d objects (as_array()) give the
    same file."""
    from limap_amd import base, io as lio
    rng = np.random.default_rng(5)
    arr = rng.normal(size=(4, 2, 3))
    outs = []
    for k, lines in enumerate((arr, [a for a in arr], [base.Line3d(a[0], a[1]) for a in arr])):
        f = tmp_path / f"l{k}.obj"
        lio.save_obj(str(f), lines)
        outs.append(f.read_text())
    assert outs[0] == outs[1] == outs[2]
    assert outs[0].count("\nl ") + outs[0].startswith("l ") == 4 and outs[0].count("v ") == 8
    lio.save_obj(str(tmp_path / "empty.obj"), [])
    assert (tmp_path / "empty.obj").read_text() == ""


# ---- tools/diff_limap_dump.py: the offline cross-check against a dump of a real limap run -----------------------
def _load_diff_tool():
    spec = importlib.util.spec_from_file_location(
        "diff_limap_dump", os.path.join(os.path.dirname(GOLD), "..", "..", "tools", "diff_limap_dump.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_diff_tool_counts_swaps_separately():
    tool = _load_diff_tool()
    a = _track()
    same = _track()
    swapped = _track()
    swapped.image_id_list = [7, 8, 9]
    swapped.line = base.Line3d(np.asarray(a.line.end, float), np.asarray(a.line.start, float))
    other = _track()
    other.image_id_list = [7, 8, 9]
    rep = tool.compare([a, other], [same, swapped])
    assert rep["ok"] and rep["matched"] == 1 and rep["swapped"] == 1 and rep["endpoint_mismatch"] == 0
    moved = _track()
    moved.line = base.Line3d(np.asarray(a.line.start, float) + 1e-3, np.asarray(a.line.end, float))
    rep = tool.compare([a], [moved])
    assert not rep["ok"] and rep["endpoint_mismatch"] == 1
    lost = _track()
    lost.line_id_list = [99] * len(lost.line_id_list)
    rep = tool.compare([a], [lost])
    assert not rep["ok"] and rep["missing_member_sets"] == 1 and rep["extra_member_sets"] == 1


def _load_tool(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(os.path.dirname(GOLD), "..", "..", "tools", name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_upstream_kit_folders_are_deterministic_and_self_consistent(tmp_path):
    """tools/export_scene_for_limap.py (the scenes of this repository as limap output folders, for a machine WITH limap):
    the small cases regenerate to the committed digests (tests/golden/export_digests.json), read back through this
    package's readers as the scene they came from, and `diff_limap_dump.py --ours-folder` (the no-GPU form: upstream's
    tracks against the folder of expected tracks) accepts the expected tracks against themselves."""
    import json as _json
    import subprocess
    kit = _load_tool("export_scene_for_limap")
    gold = _json.load(open(os.path.join(os.path.dirname(GOLD), "export_digests.json")))
    for name in ("matched_s11", "exhaustive_s12", "matched_outer2_halfpix_s13"):
        sc, cfg, d = kit.export_case(str(tmp_path), name)
        assert kit.folder_digest(d) == gold[name]["sha256_inputs"], name
        ic = ltio.read_imagecols(os.path.join(d, "imagecols.npy"))
        assert ic.get_img_ids() == [int(i) for i in sc.img_ids]
        nb, rg = ltio.read_txt_metainfos(os.path.join(d, "metainfos.txt"))
        assert nb == {int(i): [int(x) for x in sc.neighbors[int(i)]] for i in sc.img_ids} and np.array_equal(rg[0], sc.ranges[0])
        assert np.array_equal(ltio.read_txt_segments(os.path.join(d, "segments"), int(sc.img_ids[3])), sc.segs_of(3))
        if name.startswith("matched"):
            m = ltio.read_matches(os.path.join(d, "matches"), int(sc.img_ids[2]))
            want = sc.matches_of(int(sc.img_ids[2]), kit.CASES[name][5])
            assert sorted(m) == sorted(want) and all(np.array_equal(m[k], want[k]) for k in want)
    kit.CASES_TOPK[0] = kit.CASES["matched_s11"][5]
    sc, cfg, d = kit.export_case(str(tmp_path), "matched_s11")
    assert kit.write_expected(sc, cfg, d, False) == gold["matched_s11"]["expected_tracks"] > 0
    tool = os.path.join(os.path.dirname(GOLD), "..", "..", "tools", "diff_limap_dump.py")
    p = subprocess.run([sys.executable, tool, "--ours-folder", os.path.join(d, "expected"), "--tracks", os.path.join(d, "expected")],
                       capture_output=True, text=True)
    assert p.returncode == 0, p.stdout + p.stderr
    rep = _json.loads(p.stdout[p.stdout.index("{"):])
    assert rep["ok"] and rep["matched"] == gold["matched_s11"]["expected_tracks"] and rep["swapped"] == 0


@pytest.mark.gpu
def test_upstream_kit_expected_tracks_are_what_this_backend_computes(gpu_lib, tmp_path):
    """the kit's folder through the GPU path of the diff tool, the kit's expected/ folder standing in for upstream's"""
    import json as _json
    import subprocess
    kit = _load_tool("export_scene_for_limap")
    for name in ("matched_s11", "exhaustive_s12"):
        kit.CASES_TOPK[0] = kit.CASES[name][5] or None
        sc, cfg, d = kit.export_case(str(tmp_path), name)
        n = kit.write_expected(sc, cfg, d, name.startswith("exhaustive"))
        tool = os.path.join(os.path.dirname(GOLD), "..", "..", "tools", "diff_limap_dump.py")
        cmd = [sys.executable, tool, "--imagecols", os.path.join(d, "imagecols.npy"), "--metainfos", os.path.join(d, "metainfos.txt"),
               "--segments", os.path.join(d, "segments"), "--tracks", os.path.join(d, "expected"), "--cfg", os.path.join(d, "cfg.json")]
        cmd += ["--exhaustive"] if name.startswith("exhaustive") else ["--matches", os.path.join(d, "matches")]
        p = subprocess.run(cmd, capture_output=True, text=True)
        assert p.returncode == 0, p.stdout + p.stderr
        rep = _json.loads(p.stdout[p.stdout.index("{"):])
        assert rep["ok"] and rep["matched"] == n > 0 and rep["swapped"] == 0


@pytest.mark.gpu
def test_diff_tool_on_a_scene_folder(gpu_lib, oracle, tmp_path):
    """The whole flow of tools/diff_limap_dump.py with the oracle standing in for the upstream run: its tracks are
    written in limap's track format, the tool re-runs the scene folder on the GPU and must find every member set with
    endpoints inside 1e-5 and no start / end swap (the oracle and the product share the SVD procedure)."""
    import subprocess
    sc = syn.make_scene(n_views=12, n_segs=90, n_neighbors=5, seed=18)
    cfg = syn.default_triangulation_cfg()
    ic = base.ImageCollection.from_arrays(sc.img_ids, sc.kvec, sc.qvec, sc.tvec)
    ltio.save_imagecols(str(tmp_path / "imagecols.npy"), ic)
    ltio.save_txt_metainfos(str(tmp_path / "metainfos.txt"), sc.neighbors, sc.ranges)
    for n, i in enumerate(sc.img_ids):
        ltio.save_txt_segments(str(tmp_path / "segs"), int(i), sc.segs_of(n))
        ltio.save_matches(str(tmp_path / "matches"), int(i), sc.matches_of(int(i)))
    from helpers import run_oracle
    O = run_oracle(oracle, sc, cfg)
    ot = O.ComputeLineTracks()
    ups = []
    for k in range(len(ot["off"]) - 1):
        a, b = int(ot["off"][k]), int(ot["off"][k + 1])
        tr = base.LineTrack()
        tr.line = base.Line3d(ot["line"][k][:3], ot["line"][k][3:6])
        tr.image_id_list = [int(x) for x in ot["image_ids"][a:b]]
        tr.line_id_list = [int(x) for x in ot["line_ids"][a:b]]
        tr.node_id_list = [int(x) for x in ot["node_ids"][a:b]]
        tr.line2d_list = [base.Line2d(np.zeros(2), np.ones(2)) for _ in range(a, b)]
        tr.line3d_list = [base.Line3d(np.zeros(3), np.ones(3)) for _ in range(a, b)]
        tr.score_list = [0.0] * (b - a)
        ups.append(tr)
    assert len(ups) > 5
    ltio.save_folder_linetracks(str(tmp_path / "upstream_tracks"), ups)
    tool = os.path.join(os.path.dirname(GOLD), "..", "..", "tools", "diff_limap_dump.py")
    p = subprocess.run([sys.executable, tool, "--imagecols", str(tmp_path / "imagecols.npy"), "--metainfos",
                        str(tmp_path / "metainfos.txt"), "--segments", str(tmp_path / "segs"), "--matches",
                        str(tmp_path / "matches"), "--tracks", str(tmp_path / "upstream_tracks")],
                       capture_output=True, text=True)
    assert p.returncode == 0, p.stdout + p.stderr
    import json as _json
    rep = _json.loads(p.stdout[p.stdout.index("{"):])
    assert rep["ok"] and rep["swapped"] == 0 and rep["matched"] == len(ups)
