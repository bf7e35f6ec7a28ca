"""Pins the ORACLE (oracle/lt_oracle.cpp, the checker of every GPU parity test) against the REFERENCE ITSELF:
oracle/_ref = the unmodified hot-path sources of /root/reference/src/limap compiled where they lie
(oracle/Makefile `ref`; Eigen / COLMAP / PoseLib replaced by the stand-in headers of oracle/ref_shim/).
"Bit for bit" below therefore means: against the reference's sources AS COMPILED AGAINST THAT SHIM -- Eigen's evaluation
orders and its SVD sign rule are assumptions shared by shim and oracle, not facts checked against a real Eigen build
(DESIGN.md section 5).

CPU only.  What is compared is what the GPU tests compare: candidate lists in order, candidate geometry bit for
bit, scores, arg-max, valid edges IN ORDER, graph sizes, track membership and order, aggregated lines, the
post-triangulation chain, the free functions -- on every golden fixture, on randomised scenes / configurations in
both matching modes, and with every optional proposal branch.

Skipped (with the reason) only where neither /root/reference nor a prebuilt oracle/_ref exists."""
import ast
import glob
import os

import numpy as np
import pytest

from limap_amd import synthetic as syn

from helpers import compare_best, compare_candidates, compare_tracks, run_oracle, small_scene
from test_golden import _check, _feed, _load

HERE = os.path.dirname(os.path.abspath(__file__))
FIXTURES = sorted(glob.glob(os.path.join(HERE, "golden", "*.npz")))


@pytest.fixture(scope="module")
def ref():
    from oracle import ref as oref
    if not oref.available():
        pytest.skip("oracle/_ref: neither /root/reference nor a prebuilt oracle/_ref/liblimap_ref.so is present")
    return oref.module()


def test_ref_library_is_built_from_this_tree(ref):
    """The prebuilt oracle/_ref that travels to the GPU box embeds the hash of the stand-in headers, eigen_svd_ref.h,
    lt_oracle.h and ref_driver.cpp it was compiled against (oracle/Makefile): a library older than the tree fails here
    instead of silently checking the oracle against yesterday's shim."""
    from oracle import ref as oref
    assert oref.library_source_hash() == oref.tree_source_hash(), \
        "oracle/_ref/liblimap_ref.so was built from other shim / driver sources than the tree holds: make -C oracle ref"


def _same_edges_in_order(a, b):
    (aoff, ae), (boff, be) = a, b
    assert np.array_equal(aoff, boff), "valid edge counts differ"
    assert np.array_equal(ae, be), "valid edges differ (order = descending (score, tri_id), :118-142)"


def _stage_by_stage(R, O, exact_supports=True, score_rtol=1e-12):
    compare_candidates(R.get_all_tris(), O.get_all_tris())
    assert np.array_equal(R.get_num_tris(), O.get_num_tris())
    compare_best(R.get_best(), O.get_best())
    _same_edges_in_order(R.get_valid_edges(), O.get_valid_edges())
    rt, ot = R.ComputeLineTracks(), O.ComputeLineTracks()
    compare_tracks(rt, ot, exact_supports=exact_supports, score_rtol=score_rtol)
    sr, so = R.stats(), O.stats()
    for k in ("connections", "candidates", "pairs", "valid_edges", "graph_nodes", "graph_edges", "tracks"):
        assert sr[k] == so[k], (k, sr[k], so[k])
    return so


@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p)[:-4] for p in FIXTURES])
def test_reference_reproduces_golden(ref, path):
    """The committed fixtures (written by the oracle) are what the reference's own code computes."""
    d = _load(path)
    R = ref.OracleTriangulator(d["cfg"])
    _feed(R, d, lambda T: T.Init(d["img_ids"], d["kvec"], d["qvec"], d["tvec"], d["seg_off"], d["segs"]))
    best, edges, n_tris = R.get_best(), R.get_valid_edges(), R.get_num_tris()
    tracks = R.ComputeLineTracks()
    _check(d, n_tris, best, edges, tracks, R.stats(), exact_scores=False)


@pytest.mark.parametrize("k", range(12))
def test_oracle_equals_reference_on_random_scenes(ref, oracle, k):
    """tools/fuzz_parity.py's draw (scene size, linker thresholds, selection knobs, half-pixel offset; every third
    case exhaustive), reference against oracle."""
    seed = 4000 + k
    rng = np.random.default_rng(seed)
    nv, ns, nn = int(rng.integers(6, 18)), int(rng.integers(30, 150)), int(rng.integers(3, 8))
    sc = syn.make_scene(n_views=nv, n_segs=ns, n_neighbors=min(nn, nv - 1), seed=seed)
    cfg = syn.default_triangulation_cfg(debug_mode=True)
    cfg["linker3d_config"]["th_angle"] = float(rng.choice([5.0, 10.0, 20.0]))
    cfg["linker3d_config"]["th_scaleinv"] = float(rng.choice([0.005, 0.015, 0.05]))
    cfg["linker2d_config"]["th_perp"] = float(rng.choice([1.0, 2.0, 4.0]))
    cfg["IoU_threshold"] = float(rng.choice([0.05, 0.1, 0.3]))
    cfg["line_tri_angle_threshold"] = float(rng.choice([1.0, 5.0]))
    cfg["sensitivity_threshold"] = float(rng.choice([70.0, 40.0]))
    cfg["min_length_2d"] = float(rng.choice([0.0, 20.0]))
    cfg["fullscore_th"] = float(rng.choice([1.0, 2.0]))
    cfg["max_valid_conns"] = int(rng.choice([1000, 4]))
    cfg["min_num_outer_edges"] = int(rng.choice([0, 1, 2]))
    cfg["add_halfpix"] = bool(rng.integers(0, 2))
    cfg["use_endpoints_triangulation"] = bool(k % 5 == 4)
    ex = bool(k % 3 == 2)
    st = _stage_by_stage(run_oracle(ref, sc, cfg, exhaustive=ex), run_oracle(oracle, sc, cfg, exhaustive=ex))
    assert st["candidates"] > 0


def test_without_ranges_and_with_unsorted_neighbours(ref, oracle):
    sc = small_scene(seed=21, n_views=10, n_segs=70, n_neighbors=5)
    sc.ranges = None
    cfg = syn.default_triangulation_cfg(debug_mode=True)
    _stage_by_stage(run_oracle(ref, sc, cfg), run_oracle(oracle, sc, cfg))
    # exhaustive mode keeps the caller's neighbour order (base_line_triangulator.cc:113-114): reverse it
    sc2 = small_scene(seed=22, n_views=8, n_segs=50, n_neighbors=4)
    for i in sc2.neighbors:
        sc2.neighbors[i] = list(reversed(sc2.neighbors[i]))
    _stage_by_stage(run_oracle(ref, sc2, cfg, exhaustive=True), run_oracle(oracle, sc2, cfg, exhaustive=True))


@pytest.mark.parametrize("strategy", ["greedy", "exhaustive", "avg"])
def test_merging_strategies(ref, oracle, strategy):
    sc = small_scene(seed=5, n_views=20, n_segs=150, n_neighbors=8)
    cfg = syn.default_triangulation_cfg(debug_mode=True, merging_strategy=strategy)
    _stage_by_stage(run_oracle(ref, sc, cfg), run_oracle(oracle, sc, cfg))
    with pytest.raises(RuntimeError, match="merging strategy"):
        bad = run_oracle(ref, small_scene(seed=5, n_views=6, n_segs=30, n_neighbors=3),
                         dict(cfg, merging_strategy="spectral"))
        bad.ComputeLineTracks()


def _vp_run(mod, sc, cfg, vps, exhaustive):
    T = mod.OracleTriangulator(cfg)
    T.SetRanges(sc.ranges)
    T.Init(sc.img_ids, sc.kvec, sc.qvec, sc.tvec, sc.seg_off, sc.segs)
    T.InitVPResults(vps)
    for i in sc.img_ids:
        if exhaustive:
            T.TriangulateImageExhaustiveMatch(int(i), sc.neighbors[int(i)])
        else:
            T.TriangulateImage(int(i), sc.matches_of(int(i)))
    return T


@pytest.mark.parametrize("exhaustive", [False, True])
def test_vp_proposals(ref, oracle, exhaustive):
    sc = small_scene(seed=31, n_views=9, n_segs=50, n_neighbors=4)
    vps = syn.make_vp_results(sc, seed=31)
    for over in (dict(use_vp=True), dict(use_vp=True, disable_algebraic_triangulation=True)):
        cfg = syn.default_triangulation_cfg(debug_mode=True, **over)
        _stage_by_stage(_vp_run(ref, sc, cfg, vps, exhaustive), _vp_run(oracle, sc, cfg, vps, exhaustive))


def _pts_run(mod, sc, cfg, bpts, sfm, exhaustive=False):
    T = mod.OracleTriangulator(cfg)
    T.SetRanges(sc.ranges)
    T.Init(sc.img_ids, sc.kvec, sc.qvec, sc.tvec, sc.seg_off, sc.segs)
    T.SetBipartites2d(bpts)
    if sfm is not None:
        T.SetSfMPoints(sfm)
    for i in sc.img_ids:
        if exhaustive:
            T.TriangulateImageExhaustiveMatch(int(i), sc.neighbors[int(i)])
        else:
            T.TriangulateImage(int(i), sc.matches_of(int(i)))
    return T


@pytest.mark.parametrize("with_sfm", [True, False])
def test_many_points_proposal(ref, oracle, with_sfm):
    """base_line_triangulator.cc:183-236 (line fit through the shared points + Pluecker projection).  The SVD is a
    stand-in on both sides (the same one: oracle/ref_shim/Eigen/SVD), so this pins the surrounding logic -- the
    point bookkeeping through std::map, the order of the proposals -- bit for bit."""
    sc = small_scene(seed=41, n_views=9, n_segs=50, n_neighbors=4)
    bpts, sfm = syn.make_bipartites(sc, seed=41)
    cfg = syn.default_triangulation_cfg(debug_mode=True, disable_one_point_triangulation=True)
    R = _pts_run(ref, sc, cfg, bpts, sfm if with_sfm else None)
    O = _pts_run(oracle, sc, cfg, bpts, sfm if with_sfm else None)
    st = _stage_by_stage(R, O)
    plain = run_oracle(oracle, sc, syn.default_triangulation_cfg(debug_mode=True))
    assert st["candidates"] > plain.stats()["candidates"]  # the branch produced candidates


def test_one_point_solver_against_the_reference_file(ref, oracle):
    """a18: limap::solvers::triangulation::triangulate_line_with_one_point (the generated quartic in the Lagrange
    multiplier, compiled unmodified; only PoseLib's root finder is a stand-in) against the oracle, which evaluates the
    same generated polynomials term by term (oracle/onepoint_terms.inc, written from that file by
    oracle/make_onepoint_terms.py) in the reference's order: BIT-IDENTICAL lines, sentinels on the same inputs."""
    assert oracle.get_one_point_solver()  # the default: the reference's solver, not the restated problem
    rng = np.random.default_rng(7)
    sc = small_scene(seed=43, n_views=8, n_segs=60, n_neighbors=4)
    n_ok = n_sentinel = 0
    for trial in range(1200):
        a, b = rng.choice(sc.n_images, 2, replace=False)
        cam1, cam2 = sc.cam11(int(a)), sc.cam11(int(b))
        s1 = sc.segs_of(int(a))[rng.integers(0, 40)]
        s2 = sc.segs_of(int(b))[rng.integers(0, 40)]
        # a 3D point that projects near l1 (what a shared SfM point looks like)
        t = rng.uniform(0.1, 0.9)
        px = (1 - t) * s1[0:2] + t * s1[2:4] + rng.normal(0, 0.5, 2)
        ray = oracle.cam_ray_direction(cam1, px)
        point = oracle.cam_center(cam1) + ray * rng.uniform(1.5, 6.0)
        lr = ref.triangulate_line_with_one_point(s1, cam1, s2, cam2, point)
        lo = oracle.triangulate_line_with_one_point(s1, cam1, s2, cam2, point)
        assert np.array_equal(lr, lo), (trial, lr, lo)  # including the failure sentinel Line3d((0,0,0),(1,1,1),-1)
        n_ok += int(lr[9] >= 0)
        n_sentinel += int(lr[9] < 0)
    assert n_ok > 400 and n_sentinel > 50, (n_ok, n_sentinel)


def test_one_point_proposal_in_the_pipeline(ref, oracle):
    """Everything on, matched mode: which candidates exist, their order, sources AND coordinates (bit for bit: the
    oracle runs the reference's generated one-point solver), arg-max, edges and track membership identical."""
    sc = small_scene(seed=45, n_views=8, n_segs=50, n_neighbors=4)
    bpts, sfm = syn.make_bipartites(sc, seed=45)
    vps = syn.make_vp_results(sc, seed=45)
    cfg = syn.default_triangulation_cfg(debug_mode=True, use_vp=True)

    def run(mod):
        T = mod.OracleTriangulator(cfg)
        T.SetRanges(sc.ranges)
        T.Init(sc.img_ids, sc.kvec, sc.qvec, sc.tvec, sc.seg_off, sc.segs)
        T.InitVPResults(vps); T.SetBipartites2d(bpts); T.SetSfMPoints(sfm)
        for i in sc.img_ids:
            T.TriangulateImage(int(i), sc.matches_of(int(i)))
        return T
    R, O = run(ref), run(oracle)
    ra, oa = R.get_all_tris(), O.get_all_tris()
    assert np.array_equal(ra["off"], oa["off"]) and np.array_equal(ra["src"], oa["src"])
    assert np.array_equal(ra["line"], oa["line"])  # every proposal kind, the one-point candidates included
    rb, ob = R.get_best(), O.get_best()
    assert np.array_equal(rb["has_best"], ob["has_best"]) and np.array_equal(rb["src"], ob["src"])
    rt, ot = R.ComputeLineTracks(), O.ComputeLineTracks()
    for key in ("off", "image_ids", "line_ids", "node_ids"):
        assert np.array_equal(rt[key], ot[key]), key


def test_post_triangulation_chain(ref, oracle):
    """merging_utils.cc:27-155 + RemergeLineTracks (merging.cc:513-644) in the order of line_triangulation.py:171-200."""
    sc = small_scene(seed=3, n_views=24, n_segs=160, n_neighbors=8)
    cfg = syn.default_triangulation_cfg()
    linker = dict(score_th=0.5, th_angle=5.0, th_overlap=0.001, th_smartoverlap=0.1, th_smartangle=1.0, th_perp=1.0,
                  th_innerseg=1.0)
    sets = []
    for mod in (ref, oracle):
        T = run_oracle(mod, sc, cfg)
        T.ComputeLineTracks()
        ts = mod.OracleTrackSet(T)
        stages = [ts.get()]
        ts.filter_by_reprojection(8.0, 5.0); stages.append(ts.get())
        ts.remerge(linker); stages.append(ts.get())
        ts.filter_by_reprojection(8.0, 5.0); stages.append(ts.get())
        ts.filter_by_sensitivity(75.0, 3); stages.append(ts.get())
        ts.filter_by_overlap(0.5, 3); stages.append(ts.get())
        sets.append(stages)
    assert len(sets[0][0]["off"]) - 1 > 100 and len(sets[0][-1]["off"]) < len(sets[0][0]["off"])
    for a, b in zip(*sets):
        for key in ("off", "image_ids", "line_ids", "node_ids", "active"):
            assert np.array_equal(a[key], b[key]), key
        np.testing.assert_allclose(a["scores"], b["scores"], rtol=1e-12)
        assert np.array_equal(a["line2d"], b["line2d"]) and np.array_equal(a["line3d"], b["line3d"])
        assert np.array_equal(a["line"], b["line"])


def test_free_functions_bit_exact(ref, oracle):
    rng = np.random.default_rng(11)
    sc = small_scene(seed=51, n_views=10, n_segs=60, n_neighbors=5)
    for trial in range(200):
        a, b = rng.choice(sc.n_images, 2, replace=False)
        cam1, cam2 = sc.cam11(int(a)), sc.cam11(int(b))
        if trial % 7 == 0:  # un-normalised quaternion: CameraPose normalises, R() normalises again
            cam1 = cam1.copy(); cam1[4:8] *= 1.7
        s1 = sc.segs_of(int(a))[rng.integers(0, 60)]
        s2 = sc.segs_of(int(b))[rng.integers(0, 60)]
        p3 = rng.uniform(-4, 4, 3)
        for name, args in (("get_normal_direction", (s1, cam1)), ("compute_essential_matrix", (cam1, cam2)),
                           ("compute_fundamental_matrix", (cam1, cam2)), ("compute_epipolar_IoU", (s1, cam1, s2, cam2)),
                           ("triangulate_line", (s1, cam1, s2, cam2)), ("triangulate_line_by_endpoints", (s1, cam1, s2, cam2)),
                           ("triangulate_point", (s1[:2], cam1, s2[:2], cam2)), ("cam_project", (cam1, p3)),
                           ("cam_ray_direction", (cam1, s1[:2])), ("cam_projdepth", (cam1, p3)), ("cam_R", (cam1,)),
                           ("cam_center", (cam1,)), ("get_direction_from_vp", (rng.normal(size=3), cam1)),
                           ("triangulate_line_with_direction", (s1, cam1, s2, cam2, rng.normal(size=3)))):
            r, o = getattr(ref, name)(*args), getattr(oracle, name)(*args)
            if isinstance(r, tuple):
                assert r[1] == o[1] and np.array_equal(np.asarray(r[0]), np.asarray(o[0])), name
            else:
                assert np.array_equal(np.asarray(r), np.asarray(o)), (name, r, o)
        l10 = oracle.triangulate_line(s1, cam1, s2, cam2)
        if l10[9] > 0:
            assert ref.line3d_sensitivity(l10, cam2) == oracle.line3d_sensitivity(l10, cam2)
            assert ref.line3d_uncertainty(l10, cam2, 2.0) == oracle.line3d_uncertainty(l10, cam2, 2.0)


def test_linkers_bit_exact(ref, oracle):
    rng = np.random.default_rng(13)
    cfgs = [syn.default_triangulation_cfg(),
            syn.default_triangulation_cfg(linker2d_config=dict(score_th=0.3, th_angle=8.0, th_perp=3.0, th_overlap=0.2,
                                                               use_innerseg=True, th_innerseg=2.0),
                                          linker3d_config=dict(score_th=0.4, th_angle=15.0, th_overlap=0.1, th_smartoverlap=0.3,
                                                               th_smartangle=3.0, th_perp=0.5, th_innerseg=0.5, th_scaleinv=0.05,
                                                               use_perp=True))]
    n_pos = 0
    for cfg in cfgs:
        for trial in range(400):
            base = rng.uniform(50, 700, 2)
            d = rng.normal(size=2); d /= np.linalg.norm(d)
            L = rng.uniform(20, 200)
            s1 = np.concatenate([base, base + L * d])
            ang = np.deg2rad(rng.normal(0, 4))
            d2 = np.array([np.cos(ang) * d[0] - np.sin(ang) * d[1], np.sin(ang) * d[0] + np.cos(ang) * d[1]])
            off = rng.normal(0, 1.5) * np.array([-d[1], d[0]]) + rng.uniform(-0.3, 0.3) * L * d
            s2 = np.concatenate([base + off, base + off + rng.uniform(0.5, 1.2) * L * d2])
            r, o = ref.linker2d_score(cfg, s1, s2), oracle.linker2d_score(cfg, s1, s2)
            assert r == o, (trial, r, o)
            n_pos += r > 0
            p = rng.uniform(-3, 3, 3)
            u = rng.normal(size=3); u /= np.linalg.norm(u)
            l1 = np.concatenate([p, p + u * rng.uniform(0.3, 2), rng.uniform(1, 6, 2), [rng.uniform(0.002, 0.02), 1.0]])
            u2 = u + rng.normal(0, 0.05, 3); u2 /= np.linalg.norm(u2)
            q = p + rng.normal(0, 0.01, 3) + u * rng.uniform(-0.3, 0.3)
            l2 = np.concatenate([q, q + u2 * rng.uniform(0.3, 2), rng.uniform(1, 6, 2), [rng.uniform(0.002, 0.02), 1.0]])
            for mode in (0, 1, 2, 3):
                r, o = ref.linker3d_score(cfg, mode, l1, l2), oracle.linker3d_score(cfg, mode, l1, l2)
                assert r == o, (trial, mode, r, o)
                n_pos += r > 0
    assert n_pos > 200  # the scores are not trivially zero


def test_track_labels_and_aggregator(ref, oracle):
    rng = np.random.default_rng(17)
    for trial in range(30):
        n = int(rng.integers(5, 60))
        node_img = rng.integers(0, 8, n).astype(np.int32)
        m = int(rng.integers(n, 4 * n))
        en = rng.integers(0, n, (m, 2)).astype(np.int32)
        en = en[en[:, 0] != en[:, 1]]
        sim = np.round(rng.uniform(0.5, 1.0, len(en)), 1)  # ties exercise the (sim, idx1, idx2) order
        assert np.array_equal(ref.track_labels_greedy(node_img, sim, en), oracle.track_labels_greedy(node_img, sim, en))
        k = int(rng.integers(1, 12))
        p = rng.uniform(-2, 2, 3); u = rng.normal(size=3); u /= np.linalg.norm(u)
        lines = np.zeros((k, 10))
        for i in range(k):
            a, b = np.sort(rng.uniform(-1, 1, 2))
            lines[i, 0:3] = p + a * u + rng.normal(0, 0.01, 3); lines[i, 3:6] = p + b * u + rng.normal(0, 0.01, 3)
            lines[i, 6:8] = rng.uniform(1, 5, 2); lines[i, 8] = rng.uniform(0.001, 0.01); lines[i, 9] = 1.0
        scores = rng.uniform(0, 5, k)
        for no in (0, 1, 2):
            if 2 * k - 1 - no < no:
                continue
            assert np.array_equal(ref.aggregate_line3d_list(lines, scores, no), oracle.aggregate_line3d_list(lines, scores, no))


def test_reference_error_conventions(ref):
    """base_line_triangulator.cc:87-94 (std::runtime_error for an out-of-index match), :79 (THROW_CHECK on the
    match shape is upstream of this ABI), map::at for unknown images."""
    sc = small_scene(seed=61, n_views=6, n_segs=30, n_neighbors=3)
    cfg = syn.default_triangulation_cfg()
    R = ref.OracleTriangulator(cfg)
    R.Init(sc.img_ids, sc.kvec, sc.qvec, sc.tvec, sc.seg_off, sc.segs)
    i0, i1 = int(sc.img_ids[0]), int(sc.img_ids[1])
    with pytest.raises(RuntimeError, match="IndexError"):
        R.TriangulateImage(i0, {i1: np.array([[1000, 0]], np.int32)})  # line_id of the image itself (:87)
    with pytest.raises(RuntimeError):
        R.TriangulateImage(12345, {i1: np.array([[0, 0]], np.int32)})
