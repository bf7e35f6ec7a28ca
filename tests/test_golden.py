"""Committed golden fixtures (tests/golden/*.npz, made by tests/golden/make_golden.py).
CPU: the oracle reproduces them (pins the checker across compilers / libm versions).
GPU (-m gpu): the HIP backend reproduces them through the C ABI."""
import ast
import glob
import os

import numpy as np
import pytest

from limap_amd import synthetic as syn

HERE = os.path.dirname(os.path.abspath(__file__))
FIXTURES = sorted(glob.glob(os.path.join(HERE, "golden", "*.npz")))


def _load(path):
    d = dict(np.load(path, allow_pickle=False))
    d["mode"] = str(d["mode"])
    d["cfg"] = syn.default_triangulation_cfg(debug_mode=True, **ast.literal_eval(str(d["cfg_over"])))
    return d


def _feed(T, d, init):
    T.SetRanges((d["ranges"][0], d["ranges"][1]))
    init(T)
    if "vp_labels" in d:
        so = d["seg_off"]
        T.InitVPResults({int(i): (d["vp_labels"][so[n]:so[n + 1]], d["vp_vps"][n]) for n, i in enumerate(d["img_ids"])})
    for n, i in enumerate(d["img_ids"]):
        nbs = d["nb_flat"][d["nb_off"][n]:d["nb_off"][n + 1]].tolist()
        if d["mode"] == "matched":
            sel = np.nonzero(d["m_img"] == i)[0]
            m = {int(d["m_nb"][b]): d["m_rows"][d["m_off"][b]:d["m_off"][b + 1]] for b in sel}
            T.TriangulateImage(int(i), m)
        else:
            T.TriangulateImageExhaustiveMatch(int(i), nbs)


def _check(d, n_tris, best, edges, tracks, stats, exact_scores):
    assert np.array_equal(n_tris, d["n_tris"])
    assert np.array_equal(best["has_best"], d["has_best"])
    assert np.array_equal(best["src"], d["best_src"])
    assert np.array_equal(best["line"], d["best_line"]), "best candidate geometry must be bit-exact"
    if exact_scores:
        assert np.array_equal(best["score"], d["best_score"])
    else:
        np.testing.assert_allclose(best["score"], d["best_score"], rtol=1e-12)
    eoff, e = edges
    assert np.array_equal(eoff, d["edge_off"])
    es = np.concatenate([np.array(sorted(map(tuple, e[eoff[g]:eoff[g + 1]].tolist())), np.int32).reshape(-1, 2)
                         for g in range(len(eoff) - 1)], 0) if len(e) else e
    assert np.array_equal(es.reshape(-1, 2), d["edges"].reshape(-1, 2))
    assert np.array_equal(tracks["off"], d["track_off"])
    assert np.array_equal(tracks["image_ids"], d["track_img"]) and np.array_equal(tracks["line_ids"], d["track_lid"])
    assert np.array_equal(tracks["node_ids"], d["track_node"])
    gl, ol = tracks["line"], d["track_line"]
    if len(ol):
        sw = np.concatenate([ol[:, 3:6], ol[:, :3]], 1)
        scale = np.maximum(np.abs(ol[:, :6]).max(1), 1e-9)
        err = np.minimum(np.abs(gl[:, :6] - ol[:, :6]).max(1), np.abs(gl[:, :6] - sw).max(1)) / scale
        assert err.max() <= 1e-5
    for k, v in zip(("connections", "candidates", "pairs", "valid_edges", "graph_nodes", "graph_edges", "tracks"), d["stats"]):
        assert stats[k] == v, k


@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p)[:-4] for p in FIXTURES])
def test_oracle_reproduces_golden(oracle, path):
    d = _load(path)
    O = oracle.OracleTriangulator(d["cfg"], faithful=False)
    _feed(O, d, lambda T: T.Init(d["img_ids"], d["kvec"], d["qvec"], d["tvec"], d["seg_off"], d["segs"]))
    best, edges, n_tris = O.get_best(), O.get_valid_edges(), O.get_num_tris()
    tracks = O.ComputeLineTracks()
    _check(d, n_tris, best, edges, tracks, O.stats(), exact_scores=True)


@pytest.mark.gpu
@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p)[:-4] for p in FIXTURES])
def test_hip_reproduces_golden(gpu_lib, path):
    from limap_amd import triangulation as tri
    d = _load(path)
    T = tri.GlobalLineTriangulator(d["cfg"])
    segs = [d["segs"][d["seg_off"][n]:d["seg_off"][n + 1]] for n in range(len(d["img_ids"]))]
    _feed(T, d, lambda T_: T_.InitArrays(d["img_ids"], d["kvec"], d["qvec"], d["tvec"], segs))
    ctx = T.context()
    best, edges, n_tris = ctx.get_best(), ctx.get_valid_edges(), ctx.get_num_tris()
    T.ComputeLineTracks()
    _check(d, n_tris, best, edges, ctx.get_tracks(), ctx.stats(), exact_scores=False)
