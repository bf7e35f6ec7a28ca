"""-m gpu: VP-guided proposals (step 2 of triangulateOneNode, base_line_triangulator.cc:250-281):
per connection up to three candidates -- vp(l1), vp(l2), algebraic -- in the reference's order."""
import numpy as np
import pytest

from limap_amd import synthetic as syn

from helpers import compare_best, compare_candidates, compare_tracks, compare_valid_edges

pytestmark = pytest.mark.gpu


def _run_both(oracle, sc, cfg, vps, sorted_rows=True):
    from limap_amd import triangulation as tri
    T = tri.GlobalLineTriangulator(cfg)
    O = oracle.OracleTriangulator(cfg, faithful=False)
    T.SetRanges(sc.ranges); O.SetRanges(sc.ranges)
    T.InitArrays(sc.img_ids, sc.kvec, sc.qvec, sc.tvec, [sc.segs_of(i) for i in range(sc.n_images)])
    O.Init(sc.img_ids, sc.kvec, sc.qvec, sc.tvec, sc.seg_off, sc.segs)
    T.InitVPResults(vps); O.InitVPResults(vps)
    assert set(T.GetVPResults()) == set(vps) and T.GetVPResult(int(sc.img_ids[0])) is vps[int(sc.img_ids[0])]
    rng = np.random.default_rng(3)
    for i in sc.img_ids:
        m = sc.matches_of(int(i))
        if not sorted_rows:  # generic (radix-sort) grouping path
            m = {k: v[rng.permutation(len(v))] for k, v in m.items()}
        T.TriangulateImage(int(i), m)
        O.TriangulateImage(int(i), m)
    return T, O


@pytest.mark.parametrize("sorted_rows", [True, False])
def test_vp_proposals_match_oracle(gpu_lib, oracle, sorted_rows):
    sc = syn.make_scene(n_views=14, n_segs=100, n_neighbors=5, seed=41)
    vps = syn.make_vp_results(sc, seed=1)
    cfg = syn.default_triangulation_cfg(debug_mode=True)
    cfg.update(use_vp=True)
    T, O = _run_both(oracle, sc, cfg, vps, sorted_rows)
    g, o = T.context().get_all_tris(), O.get_all_tris()
    # the VP branch really contributes: more candidates than connections that pass the algebraic gates
    cfg0 = dict(cfg, use_vp=False)
    from helpers import run_product
    n_alg = run_product(sc, cfg0).context().stats()["candidates"]
    assert g["off"][-1] > 1.3 * n_alg
    compare_candidates(g, o)
    compare_best(T.context().get_best(), O.get_best())
    compare_valid_edges(T.context().get_valid_edges(), O.get_valid_edges())
    T.context().compute_tracks()
    compare_tracks(T.context().get_tracks(), O.ComputeLineTracks())


def test_vp_proposals_exhaustive_match_oracle(gpu_lib, oracle):
    """TriangulateImageExhaustiveMatch with VP proposals: three survivor ballots per work item."""
    from limap_amd import triangulation as tri
    sc = syn.make_scene(n_views=8, n_segs=70, n_neighbors=4, seed=47)  # 70 segments: a ragged last chunk of 64
    vps = syn.make_vp_results(sc, seed=4)
    cfg = syn.default_triangulation_cfg(debug_mode=True)
    cfg.update(use_vp=True)
    T = tri.GlobalLineTriangulator(cfg)
    O = oracle.OracleTriangulator(cfg, faithful=False)
    T.SetRanges(sc.ranges); O.SetRanges(sc.ranges)
    T.InitArrays(sc.img_ids, sc.kvec, sc.qvec, sc.tvec, [sc.segs_of(i) for i in range(sc.n_images)])
    O.Init(sc.img_ids, sc.kvec, sc.qvec, sc.tvec, sc.seg_off, sc.segs)
    T.InitVPResults(vps); O.InitVPResults(vps)
    for i in sc.img_ids:
        T.TriangulateImageExhaustiveMatch(int(i), sc.neighbors[int(i)])
        O.TriangulateImageExhaustiveMatch(int(i), sc.neighbors[int(i)])
    g, o = T.context().get_all_tris(), O.get_all_tris()
    assert g["off"][-1] > 0
    compare_candidates(g, o)
    compare_best(T.context().get_best(), O.get_best())
    compare_valid_edges(T.context().get_valid_edges(), O.get_valid_edges())
    T.context().compute_tracks()
    compare_tracks(T.context().get_tracks(), O.ComputeLineTracks())


def test_vp_only(gpu_lib, oracle):
    """disable_algebraic_triangulation: the VP candidates alone."""
    sc = syn.make_scene(n_views=10, n_segs=80, n_neighbors=4, seed=43)
    vps = syn.make_vp_results(sc, seed=2)
    cfg = syn.default_triangulation_cfg(debug_mode=True)
    cfg.update(use_vp=True, disable_algebraic_triangulation=True)
    T, O = _run_both(oracle, sc, cfg, vps)
    compare_candidates(T.context().get_all_tris(), O.get_all_tris())
    assert T.context().stats()["candidates"] > 0
    T.context().compute_tracks()
    compare_tracks(T.context().get_tracks(), O.ComputeLineTracks())


def test_vp_switches_and_errors(gpu_lib, oracle):
    from limap_amd import triangulation as tri
    sc = syn.make_scene(n_views=6, n_segs=40, n_neighbors=3, seed=44)
    vps = syn.make_vp_results(sc, seed=3)
    # disable_vp_triangulation turns the branch off even with use_vp
    cfg = syn.default_triangulation_cfg(debug_mode=True)
    cfg.update(use_vp=True, disable_vp_triangulation=True)
    T, O = _run_both(oracle, sc, cfg, vps)
    compare_candidates(T.context().get_all_tris(), O.get_all_tris())
    # use_vp without InitVPResults is an error, exhaustive matching with VPs is not implemented
    cfg.update(disable_vp_triangulation=False)
    T = tri.GlobalLineTriangulator(cfg)
    T.InitArrays(sc.img_ids, sc.kvec, sc.qvec, sc.tvec, [sc.segs_of(i) for i in range(sc.n_images)])
    T.TriangulateImage(int(sc.img_ids[0]), sc.matches_of(int(sc.img_ids[0])))
    with pytest.raises(RuntimeError, match="InitVPResults"):
        T.ComputeLineTracks()
    T = tri.GlobalLineTriangulator(cfg)
    T.InitArrays(sc.img_ids, sc.kvec, sc.qvec, sc.tvec, [sc.segs_of(i) for i in range(sc.n_images)])
    # wrong label count
    bad = dict(vps)
    k = int(sc.img_ids[1])
    bad[k] = (bad[k][0][:-1], bad[k][1])
    with pytest.raises((RuntimeError, ValueError), match="labels"):
        T.InitVPResults(bad)
