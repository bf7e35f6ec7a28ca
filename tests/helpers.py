"""Shared helpers of the parity tests: run the HIP backend (through the C ABI) and the CPU oracle on
the same seeded inputs and compare stage by stage."""
import numpy as np

from limap_amd import synthetic as syn


def run_product(scene, cfg, exhaustive=False, images=None, topk=None, vps=None):
    from limap_amd import triangulation as tri
    T = tri.GlobalLineTriangulator(cfg)
    if scene.ranges is not None:
        T.SetRanges(scene.ranges)
    T.InitArrays(scene.img_ids, scene.kvec, scene.qvec, scene.tvec, [scene.segs_of(i) for i in range(scene.n_images)])
    if vps is not None:
        T.InitVPResults(vps)
    for i in (scene.img_ids if images is None else images):
        if exhaustive:
            T.TriangulateImageExhaustiveMatch(int(i), scene.neighbors[int(i)])
        else:
            T.TriangulateImage(int(i), scene.matches_of(i, topk))
    return T


def run_oracle(ora, scene, cfg, exhaustive=False, images=None, topk=None, faithful=False):
    O = ora.OracleTriangulator(cfg, faithful=faithful)
    if scene.ranges is not None:
        O.SetRanges(scene.ranges)
    O.Init(scene.img_ids, scene.kvec, scene.qvec, scene.tvec, scene.seg_off, scene.segs)
    for i in (scene.img_ids if images is None else images):
        if exhaustive:
            O.TriangulateImageExhaustiveMatch(int(i), scene.neighbors[int(i)])
        else:
            O.TriangulateImage(int(i), scene.matches_of(i, topk))
    return O


def ulp_diff(a, b):
    """max distance in units of the last place between two float64 arrays (same sign assumed)."""
    a = np.ascontiguousarray(a, np.float64).view(np.int64)
    b = np.ascontiguousarray(b, np.float64).view(np.int64)
    return int(np.max(np.abs(a - b))) if a.size else 0


def compare_candidates(g_all, o_all):
    """All generated candidates: identical CSR, identical source ids, coordinates bit-exact
    (no transcendental function touches them), scores within a few ulp (acos/exp differ between
    the device libm and glibc)."""
    assert np.array_equal(g_all["off"], o_all["off"]), "candidate counts per node differ"
    assert np.array_equal(g_all["src"], o_all["src"]), "candidate source (ng_img, ng_line) differ"
    assert np.array_equal(g_all["line"], o_all["line"]), (
        "candidate geometry not bit-exact: max ulp %d" % ulp_diff(g_all["line"], o_all["line"]))
    zero_g, zero_o = g_all["score"] == 0, o_all["score"] == 0
    assert np.array_equal(zero_g, zero_o), "support gating differs (score == 0 sets differ)"
    np.testing.assert_allclose(g_all["score"], o_all["score"], rtol=1e-12, atol=0)


def compare_best(gb, ob):
    assert np.array_equal(gb["has_best"], ob["has_best"])
    assert np.array_equal(gb["src"], ob["src"]), "best candidate (arg-max) differs"
    assert np.array_equal(gb["line"], ob["line"]), "best candidate geometry not bit-exact"
    np.testing.assert_allclose(gb["score"], ob["score"], rtol=1e-12, atol=0)


def edge_sets(off, edges):
    return [set(map(tuple, edges[off[g]:off[g + 1]].tolist())) for g in range(len(off) - 1)]


def compare_valid_edges(g, o):
    (goff, ge), (ooff, oe) = g, o
    assert np.array_equal(goff, ooff), "valid edge counts differ"
    # the reference stores them in descending (score, tri_id) order; the order has no observable
    # effect (they feed a std::set), so compare as sets
    assert edge_sets(goff, ge) == edge_sets(ooff, oe)


def compare_tracks(gt, ot, rtol=1e-5, score_rtol=1e-12, exact_supports=True):
    """Track MEMBERSHIP always bit-exact (images, lines, node ids of every track: the north-star bar), endpoints within
    1e-5 relative IN THE SAME ORIENTATION (start to start, end to end: the product's principal axis follows the oracle's
    / the Eigen stand-in's sign rule for the SVD of merging/aggregator.cc:76-78, no start / end swap is tolerated).
    `exact_supports=False` / a looser `score_rtol` only relax the COORDINATES of the supporting 3D lines and the scores
    to 1e-9 / score_rtol, for candidates whose coordinates are themselves only equal to rounding (point proposals).
    Returns (max relative endpoint error, number of tracks that would only match swapped)."""
    assert np.array_equal(gt["off"], ot["off"]), "track sizes differ"
    assert np.array_equal(gt["image_ids"], ot["image_ids"])
    assert np.array_equal(gt["line_ids"], ot["line_ids"])
    assert np.array_equal(gt["node_ids"], ot["node_ids"])
    np.testing.assert_allclose(gt["scores"], ot["scores"], rtol=score_rtol)
    if exact_supports:
        assert np.array_equal(gt["line3d"], ot["line3d"])
    else:
        np.testing.assert_allclose(gt["line3d"], ot["line3d"], rtol=1e-9, atol=1e-12)
    gl, ol = gt["line"], ot["line"]
    scale = np.maximum(np.abs(ol[:, :6]).max(axis=1, keepdims=True), 1e-9)
    d_same = (np.abs(gl[:, :6] - ol[:, :6]) / scale).max(axis=1)
    swapped = np.concatenate([ol[:, 3:6], ol[:, 0:3]], 1)
    d_swap = (np.abs(gl[:, :6] - swapped) / scale).max(axis=1)
    n_swapped = int(np.count_nonzero((d_swap < d_same) & (d_same > rtol)))
    assert n_swapped == 0, "%d of %d tracks have start and end exchanged" % (n_swapped, len(d_same))
    assert d_same.max() <= rtol if len(d_same) else True, "track endpoints differ: max rel err %g" % d_same.max()
    np.testing.assert_allclose(gl[:, 6], ol[:, 6], rtol=score_rtol)
    return (float(d_same.max()) if len(d_same) else 0.0), n_swapped


def small_scene(seed=0, n_views=16, n_segs=120, n_neighbors=8, **kw):
    return syn.make_scene(n_views=n_views, n_segs=n_segs, n_neighbors=n_neighbors, seed=seed, **kw)


import contextlib


@contextlib.contextmanager
def restated_one_point(oracle):
    """The oracle's one-point proposal on the restated optimisation problem -- the form the device code follows, bit
    for bit -- instead of the reference's generated polynomials (the default, bit-identical to oracle/_ref; the two forms
    are tied together on the CPU by tests/test_oracle_kat.py::test_one_point_restated_problem_equals_generated_solver)."""
    prev = oracle.get_one_point_solver()
    oracle.set_one_point_solver(False)
    try:
        yield
    finally:
        oracle.set_one_point_solver(prev)

