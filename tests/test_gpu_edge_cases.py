"""-m gpu: empty / ragged inputs and the error conventions of the boundary
(base_line_triangulator.cc:49,79,87-94; global_line_triangulator.cc:314-316)."""
import numpy as np
import pytest

from limap_amd import synthetic as syn

from helpers import compare_best, compare_tracks, compare_valid_edges

pytestmark = pytest.mark.gpu


def _ragged_scene():
    """Image 2 has no segments, image 5 has no neighbours, image 7 has empty match arrays, the
    segment counts differ per image."""
    sc = syn.make_scene(n_views=10, n_segs=60, n_neighbors=5, seed=9)
    keep = [60, 45, 0, 60, 31, 60, 17, 60, 60, 52]
    segs = [sc.segs_of(i)[:keep[i]] for i in range(10)]
    matches = {}
    for i in sc.img_ids:
        m = sc.matches_of(int(i))
        out = {}
        for nb, rows in m.items():
            ok = (rows[:, 0] < keep[int(i)]) & (rows[:, 1] < keep[int(nb)])
            out[nb] = rows[ok]
        if int(i) == 5:
            out = {}
        if int(i) == 7:
            out = {nb: np.zeros((0, 2), np.int32) for nb in out}
        matches[int(i)] = out
    return sc, segs, matches


def test_ragged_and_empty_inputs(gpu_lib, oracle):
    from limap_amd import triangulation as tri
    sc, segs, matches = _ragged_scene()
    cfg = syn.default_triangulation_cfg(debug_mode=True)
    T = tri.GlobalLineTriangulator(cfg)
    O = oracle.OracleTriangulator(cfg, faithful=False)
    T.SetRanges(sc.ranges); O.SetRanges(sc.ranges)
    T.InitArrays(sc.img_ids, sc.kvec, sc.qvec, sc.tvec, segs)
    off = np.zeros(11, np.int64); off[1:] = np.cumsum([len(s) for s in segs])
    O.Init(sc.img_ids, sc.kvec, sc.qvec, sc.tvec, off, np.concatenate(segs, 0))
    for i in sc.img_ids:
        T.TriangulateImage(int(i), matches[int(i)])
        O.TriangulateImage(int(i), matches[int(i)])
    assert T.CountLines(2) == 0 and T.CountImages() == 10
    assert np.array_equal(T.context().get_num_tris(), O.get_num_tris())
    compare_best(T.context().get_best(), O.get_best())
    compare_valid_edges(T.context().get_valid_edges(), O.get_valid_edges())
    T.ComputeLineTracks()
    compare_tracks(T.context().get_tracks(), O.ComputeLineTracks())


def test_nothing_to_triangulate(gpu_lib):
    from limap_amd import triangulation as tri
    sc = syn.make_scene(n_views=3, n_segs=10, n_neighbors=2, seed=0)
    T = tri.GlobalLineTriangulator(syn.default_triangulation_cfg())
    T.InitArrays(sc.img_ids, sc.kvec, sc.qvec, sc.tvec, [sc.segs_of(i) for i in range(3)])
    assert T.ComputeLineTracks() == [] and T.GetTracks() == []
    T.TriangulateImage(0, {})
    T.TriangulateImageExhaustiveMatch(1, [])
    assert T.ComputeLineTracks() == []
    b = T.context().get_best()
    assert not b["has_best"].any() and np.all(b["line"] == 0)
    # exhaustive with empty neighbour images
    T2 = tri.GlobalLineTriangulator(syn.default_triangulation_cfg())
    T2.InitArrays([0, 1], sc.kvec[:2], sc.qvec[:2], sc.tvec[:2], [sc.segs_of(0), np.zeros((0, 4))])
    T2.TriangulateImageExhaustiveMatch(0, [1])
    T2.TriangulateImageExhaustiveMatch(1, [0])
    assert T2.ComputeLineTracks() == []


def test_error_conventions(gpu_lib):
    from limap_amd import triangulation as tri
    sc = syn.make_scene(n_views=4, n_segs=20, n_neighbors=2, seed=1)
    cfg = syn.default_triangulation_cfg()
    segs = [sc.segs_of(i) for i in range(4)]
    T = tri.GlobalLineTriangulator(cfg)
    with pytest.raises(RuntimeError):
        T.TriangulateImage(0, {})  # before Init
    T.InitArrays(sc.img_ids, sc.kvec, sc.qvec, sc.tvec, segs)
    with pytest.raises(RuntimeError, match="IndexError! Out-of-index matches"):
        T.TriangulateImage(0, {1: np.array([[25, 0]])})  # line id >= M
    with pytest.raises(RuntimeError, match="IndexError"):
        T.TriangulateImage(0, {1: np.array([[0, -1]])})
    with pytest.raises(ValueError, match="cols"):
        T.TriangulateImage(0, {1: np.zeros((3, 3), np.int32)})
    with pytest.raises((IndexError, ValueError)):
        T.TriangulateImage(0, {99: np.array([[0, 0]])})  # unknown neighbour image
    with pytest.raises((IndexError, ValueError)):
        T.TriangulateImage(77, {})
    with pytest.raises(IndexError):
        T.CountLines(77)
    # a failed call leaves nothing behind: the image can still be triangulated
    T.TriangulateImage(0, {1: np.array([[0, 0], [3, 4]], np.int64)})  # other int dtypes are converted
    T.ComputeLineTracks()
    with pytest.raises(RuntimeError, match="merging strategy"):
        T3 = tri.GlobalLineTriangulator(dict(cfg, merging_strategy="spectral"))
        T3.InitArrays(sc.img_ids, sc.kvec, sc.qvec, sc.tvec, segs)
        T3.ComputeLineTracks()
    with pytest.raises(RuntimeError, match="InitVPResults"):  # use_vp needs the VP detections
        T4 = tri.GlobalLineTriangulator(dict(cfg, use_vp=True))
        T4.InitArrays(sc.img_ids, sc.kvec, sc.qvec, sc.tvec, segs)
        T4.TriangulateImage(0, {1: np.array([[0, 0]], np.int32)})
        T4.ComputeLineTracks()
    # degenerate one-point query (the point is the origin, nowhere near the lines): a Line3d comes back,
    # valid or the failure sentinel, never an exception
    l = tri.triangulate_line_with_one_point(segs[0][0], sc.cam11(0), segs[1][0], sc.cam11(1), np.zeros(3))
    assert l.score in (1.0, -1.0)
    with pytest.raises((ValueError, RuntimeError), match="255"):
        T5 = tri.GlobalLineTriangulator(cfg)
        T5.InitArrays(sc.img_ids, sc.kvec, sc.qvec, sc.tvec, segs)
        T5.TriangulateImageExhaustiveMatch(0, list(range(300)))


def test_free_functions_match_oracle(gpu_lib, oracle):
    from limap_amd import triangulation as tri
    sc = syn.make_scene(n_views=6, n_segs=40, n_neighbors=3, seed=2)
    c1, c2 = sc.cam11(0), sc.cam11(sc.neighbors[0][0])
    j = int(np.searchsorted(sc.img_ids, sc.neighbors[0][0]))
    g1, g2 = sc.gt_ids[sc.seg_off[0]:sc.seg_off[1]], sc.gt_ids[sc.seg_off[j]:sc.seg_off[j + 1]]
    common = [g for g in g1 if g >= 0 and g in set(g2.tolist())]
    s1 = sc.segs_of(0)[list(g1).index(common[0])]
    s2 = sc.segs_of(j)[list(g2).index(common[0])]
    assert np.array_equal(tri.get_normal_direction(s1, c1), oracle.get_normal_direction(s1, c1))
    assert np.array_equal(tri.compute_fundamental_matrix(c1, c2), oracle.compute_fundamental_matrix(c1, c2))
    assert tri.compute_epipolar_IoU(s1, c1, s2, c2) == oracle.compute_epipolar_IoU(s1, c1, s2, c2)
    for fn_g, fn_o in ((tri.triangulate_line, oracle.triangulate_line),
                       (tri.triangulate_line_by_endpoints, oracle.triangulate_line_by_endpoints)):
        l, o = fn_g(s1, c1, s2, c2), fn_o(s1, c1, s2, c2)
        assert np.array_equal(np.concatenate([l.start, l.end]), o[:6]) and np.array_equal(l.depths, o[6:8])
        assert l.score == o[9]
    # VP / direction / point functions
    K = np.array([[c1[0], 0, c1[2]], [0, c1[1], c1[3]], [0, 0, 1.0]])
    R = syn.quat_to_rot(c1[4:8])
    gt = sc.gt_lines[common[0]]
    d = (gt[3:] - gt[:3]) / np.linalg.norm(gt[3:] - gt[:3])
    vp = K @ R @ d
    assert np.array_equal(tri.get_direction_from_VP(vp, c1), oracle.get_direction_from_vp(vp, c1))
    for direction in (d, oracle.get_direction_from_vp(vp, c1), np.array([0.3, -0.2, 0.93])):
        l, o = tri.triangulate_line_with_direction(s1, c1, s2, c2, direction), \
            oracle.triangulate_line_with_direction(s1, c1, s2, c2, direction)
        assert np.array_equal(np.concatenate([l.start, l.end]), o[:6]) and np.array_equal(l.depths, o[6:8])
        assert l.score == o[9]
    (pg, okg), (po, oko) = tri.triangulate_point(s1[:2], c1, s2[:2], c2), oracle.triangulate_point(s1[:2], c1, s2[:2], c2)
    assert okg == oko and (not oko or np.array_equal(pg, po))
    (pg, okg), (po, oko) = tri.triangulate_point(s1[2:], c1, s2[2:], c2), oracle.triangulate_point(s1[2:], c1, s2[2:], c2)
    assert okg == oko and (not oko or np.array_equal(pg, po))
    # one-point proposal: product and oracle run the same restated solver with different libm's -> 1e-9
    from helpers import restated_one_point
    with restated_one_point(oracle):
        for pt in (0.5 * (gt[:3] + gt[3:]) + 0.003, gt[:3] * 0.7 + gt[3:] * 0.3 - 0.002, gt[3:] + 0.01):
            l, o = tri.triangulate_line_with_one_point(s1, c1, s2, c2, pt), \
                oracle.triangulate_line_with_one_point(s1, c1, s2, c2, pt)
            assert l.score == o[9]
            if o[9] > 0:
                np.testing.assert_allclose(np.concatenate([l.start, l.end, l.depths]), o[:8], rtol=1e-9, atol=1e-12)
    E = tri.compute_essential_matrix(c1, c2)
    np.testing.assert_allclose(E / np.linalg.norm(E), oracle.compute_essential_matrix(c1, c2) /
                               np.linalg.norm(oracle.compute_essential_matrix(c1, c2)), atol=1e-9)


def test_triangulate_all_equals_the_per_image_loop(gpu_lib):
    """TriangulateAll(matches_by_image) = the caller's TriangulateImage loop as one call: identical candidates, best
    lines, edges and tracks (through the pybind shim and through the ctypes path); an out-of-index row raises the
    reference's IndexError text and keeps NOTHING of the call; images that are already triangulated are skipped."""
    import numpy as np
    from limap_amd import synthetic as syn, triangulation as tri
    sc = syn.make_scene(n_views=10, n_segs=90, n_neighbors=5, seed=17)
    cfg = syn.default_triangulation_cfg(debug_mode=True)
    matches = {int(i): sc.matches_of(int(i)) for i in sc.img_ids}

    def new():
        T = tri.GlobalLineTriangulator(cfg)
        T.SetRanges(sc.ranges)
        T.InitArrays(sc.img_ids, sc.kvec, sc.qvec, sc.tvec, [sc.segs_of(i) for i in range(sc.n_images)])
        return T

    def results(T):
        T.ComputeLineTracks()
        c = T.context()
        return c.get_all_tris(), c.get_best(), c.get_valid_edges(), c.get_tracks()

    A = new()
    for i in sc.img_ids:
        A.TriangulateImage(int(i), matches[int(i)])
    ra = results(A)
    for use_shim in (True, False):
        B = new()
        if not use_shim:
            B._pbv = None  # the ctypes path
        B.TriangulateAll(matches)
        B.TriangulateAll({int(sc.img_ids[0]): matches[int(sc.img_ids[0])]})  # already scored: nothing happens
        rb = results(B)
        for k in ("off", "src", "line", "score"):
            assert np.array_equal(ra[0][k], rb[0][k]), k
        assert np.array_equal(ra[1]["line"], rb[1]["line"]) and np.array_equal(ra[1]["src"], rb[1]["src"])
        assert np.array_equal(ra[2][0], rb[2][0]) and np.array_equal(ra[2][1], rb[2][1])
        for k in ("off", "image_ids", "line_ids", "node_ids", "line"):
            assert np.array_equal(ra[3][k], rb[3][k]), k
    # an out-of-index line id in the third image: the reference's message, and nothing of the call is buffered
    bad = {k: dict(v) for k, v in matches.items()}
    i3 = int(sc.img_ids[2])
    nb0 = next(iter(bad[i3]))
    rows = bad[i3][nb0].copy()
    rows[0, 0] = 10_000
    bad[i3][nb0] = rows
    Cx = new()
    with pytest.raises(RuntimeError, match="Out-of-index matches exist between image"):
        Cx.TriangulateAll(bad)
    Cx.TriangulateAll(matches)  # the failed call left no image marked as triangulated
    rc = results(Cx)
    assert np.array_equal(ra[3]["off"], rc[3]["off"]) and np.array_equal(ra[3]["line"], rc[3]["line"])
