"""Known-answer tests that pin the CPU oracle: the reference's own tests hold no
golden vectors for this path, so the restatement is checked against closed-form geometry computed
independently here with numpy.  Runs on CPU (-m "not gpu")."""
import numpy as np
import pytest

from limap_amd import synthetic as syn


def K_of(k4):
    return np.array([[k4[0], 0, k4[2]], [0, k4[1], k4[3]], [0, 0, 1.0]])


def project(cam, X):
    R = syn.quat_to_rot(cam[4:8])
    x = K_of(cam[:4]) @ (R @ X + cam[8:11])
    return x[:2] / x[2]


def depth(cam, X):
    return (syn.quat_to_rot(cam[4:8]) @ X + cam[8:11])[2]


def look_at_cam(C, target, f=700.0, roll=0.0):
    fwd = (target - C) / np.linalg.norm(target - C)
    up = np.array([0, 0, 1.0])
    right = np.cross(fwd, up); right /= np.linalg.norm(right)
    down = np.cross(fwd, right)
    r2 = np.cos(roll) * right + np.sin(roll) * down
    R = np.stack([r2, np.cross(fwd, r2), fwd], 0)
    q = syn._rot_to_quat(R)
    t = -syn.quat_to_rot(q) @ C
    return np.concatenate([[f, f, 400.0, 300.0], q, t])


@pytest.fixture(scope="module")
def two_views():
    rng = np.random.default_rng(5)
    P, Q = np.array([1.0, 4.0, 1.2]), np.array([2.5, 4.3, 1.9])
    c1 = look_at_cam(np.array([0.5, 0.0, 1.0]), 0.5 * (P + Q) + rng.normal(0, 0.2, 3), roll=0.05)
    c2 = look_at_cam(np.array([3.0, 0.5, 1.6]), 0.5 * (P + Q) + rng.normal(0, 0.2, 3), roll=-0.08)
    return P, Q, c1, c2


def test_camera_primitives(oracle, two_views):
    P, Q, c1, c2 = two_views
    np.testing.assert_allclose(oracle.cam_R(c1), syn.quat_to_rot(c1[4:8]), atol=1e-15)
    C = -syn.quat_to_rot(c1[4:8]).T @ c1[8:11]
    np.testing.assert_allclose(oracle.cam_center(c1), C, atol=1e-14)
    np.testing.assert_allclose(oracle.cam_project(c1, P), project(c1, P), rtol=1e-12)
    assert abs(oracle.cam_projdepth(c1, P) - depth(c1, P)) < 1e-13
    ray = oracle.cam_ray_direction(c1, project(c1, P))
    np.testing.assert_allclose(ray, (P - C) / np.linalg.norm(P - C), atol=1e-12)


def test_kat_i_exact_projection(oracle, two_views):
    """(i) a GT segment projected exactly into two views: IoU == 1, triangulate_line recovers the
    endpoints, depths equal projdepth."""
    P, Q, c1, c2 = two_views
    s1 = np.concatenate([project(c1, P), project(c1, Q)])
    s2 = np.concatenate([project(c2, P), project(c2, Q)])
    # dehomogeneous() divides by (z + 1e-12) (util/types.h:40-45) and z of the crossed, normalised
    # line coordinates is ~1e-6 here, so the reference's IoU is only exact to ~1e-6
    assert abs(oracle.compute_epipolar_IoU(s1, c1, s2, c2) - 1.0) < 1e-5
    l = oracle.triangulate_line(s1, c1, s2, c2)
    np.testing.assert_allclose(l[0:3], P, atol=1e-9)
    np.testing.assert_allclose(l[3:6], Q, atol=1e-9)
    assert abs(l[6] - depth(c1, P)) < 1e-9 and abs(l[7] - depth(c1, Q)) < 1e-9
    assert l[9] == 1.0
    le = oracle.triangulate_line_by_endpoints(s1, c1, s2, c2)
    np.testing.assert_allclose(le[0:6], np.concatenate([P, Q]), atol=1e-9)
    # F from the oracle satisfies the epipolar constraint x2^T F x1 = 0
    F = oracle.compute_fundamental_matrix(c1, c2)
    x1, x2 = np.append(project(c1, P), 1), np.append(project(c2, P), 1)
    assert abs(x2 @ F @ x1) / np.linalg.norm(F) < 1e-9
    # plane normal of the back-projected segment is orthogonal to both viewing rays
    n = oracle.get_normal_direction(s1, c1)
    C1 = oracle.cam_center(c1)
    assert abs(n @ (P - C1)) < 1e-9 and abs(n @ (Q - C1)) < 1e-9 and abs(np.linalg.norm(n) - 1) < 1e-12


def test_kat_ii_partial_overlap(oracle, two_views):
    """(ii) view 2 observes only a sub-interval: IoU equals the analytic interval ratio."""
    P, Q, c1, c2 = two_views
    X = lambda t: P + t * (Q - P)
    s1 = np.concatenate([project(c1, X(0.0)), project(c1, X(1.0))])
    a, b = 0.3, 1.4
    l2s, l2e = project(c2, X(a)), project(c2, X(b))
    s2 = np.concatenate([l2s, l2e])
    d = (l2e - l2s) / np.linalg.norm(l2e - l2s)
    L = np.linalg.norm(l2e - l2s)
    c_lo = (project(c2, X(0.0)) - l2s) @ d / L
    c_hi = (project(c2, X(1.0)) - l2s) @ d / L
    c_lo, c_hi = min(c_lo, c_hi), max(c_lo, c_hi)
    expect = (min(c_hi, 1) - max(c_lo, 0)) / (max(c_hi, 1) - min(c_lo, 0))
    assert abs(oracle.compute_epipolar_IoU(s1, c1, s2, c2) - expect) < 1e-5
    assert 0.2 < expect < 0.9


def _two_image_run(oracle, c1, c2, s1, s2, **cfg_over):
    cfg = syn.default_triangulation_cfg(debug_mode=True, **cfg_over)
    O = oracle.OracleTriangulator(cfg, faithful=True)
    O.Init([0, 1], np.stack([c1[:4], c2[:4]]), np.stack([c1[4:8], c2[4:8]]), np.stack([c1[8:], c2[8:]]), [0, 1, 2],
           np.stack([s1, s2]))
    O.TriangulateImage(0, {1: np.array([[0, 0]], np.int32)})
    return O


def test_kat_iii_degenerate_plane(oracle, two_views):
    """(iii) camera 1 inside the back-projected plane of l2: ray/plane angle 0 -> no proposal."""
    P, Q, c1, c2 = two_views
    C2 = oracle.cam_center(c2)
    C1_deg = C2 + 0.7 * (P - C2) * 0.2 + 0.5 * (Q - P)  # in the plane spanned by C2 and the line
    c1d = look_at_cam(C1_deg, 0.5 * (P + Q))
    s1 = np.concatenate([project(c1d, P), project(c1d, Q)])
    s2 = np.concatenate([project(c2, P), project(c2, Q)])
    assert _two_image_run(oracle, c1d, c2, s1, s2).get_num_tris()[0] == 0
    s1ok = np.concatenate([project(c1, P), project(c1, Q)])
    O = _two_image_run(oracle, c1, c2, s1ok, s2)
    assert O.get_num_tris()[0] == 1
    tri = O.get_all_tris()
    np.testing.assert_allclose(tri["line"][0, :6], np.concatenate([P, Q]), atol=1e-8)
    # uncertainty = min over both views of var2d * mean depth / f  (linebase.cc:109-116)
    u = min(2.0 * 0.5 * (depth(c, P) + depth(c, Q)) / 700.0 for c in (c1, c2))
    assert abs(tri["line"][0, 8] - u) < 1e-12


def test_kat_iv_behind_camera(oracle, two_views):
    """(iv) a segment behind both cameras triangulates to negative depths -> sentinel score -1."""
    P, Q, c1, c2 = two_views
    C1 = oracle.cam_center(c1)
    back = lambda X: C1 - (X - C1)  # mirrored through C1: projects to a valid pixel, depth < 0 in view 1
    s1 = np.concatenate([project(c1, back(P)), project(c1, back(Q))])
    s2 = np.concatenate([project(c2, back(P)), project(c2, back(Q))])
    l = oracle.triangulate_line(s1, c1, s2, c2)
    assert l[9] == -1.0 and np.allclose(l[:6], [0, 0, 0, 1, 1, 1])
    assert _two_image_run(oracle, c1, c2, s1, s2, line_tri_angle_threshold=0.0, IoU_threshold=-1e9).get_num_tris()[0] == 0


def test_kat_v_linker_scores(oracle):
    """(v) exp-scored gates: s = exp(-(x/(th m))^2/2), m = 1/sqrt(-2 ln score_th); exactly 0 above
    the threshold, == score_th at the threshold."""
    cfg = syn.default_triangulation_cfg()
    m = 1.0 / np.sqrt(-2.0 * np.log(0.5))
    base = np.array([100.0, 100.0, 300.0, 100.0])
    for theta in (0.5, 2.0, 4.9):
        rot = np.deg2rad(theta)
        seg = np.array([100.0, 100.0, 100 + 200 * np.cos(rot), 100 + 200 * np.sin(rot)])
        s = oracle.linker2d_score(cfg, seg, base)
        # angle, overlap (=1), smart-angle (no shrink: overlap 1), perpendicular (endpoint offset)
        perp = 200 * np.sin(rot)
        expect = min(np.exp(-(theta / (5.0 * m)) ** 2 / 2), np.exp(-(perp / (2.0 * m)) ** 2 / 2))
        expect = 0.0 if expect < 0.5 else expect
        assert abs(s - expect) < 1e-9, theta
    seg = np.array([100.0, 100.0, 100 + 200 * np.cos(np.deg2rad(5.2)), 100 + 200 * np.sin(np.deg2rad(5.2))])
    assert oracle.linker2d_score(cfg, seg, base) == 0.0
    # pure perpendicular offset d: parallel segments
    for d, want0 in ((0.5, False), (1.99, False), (2.01, True)):
        s = oracle.linker2d_score(cfg, base + np.array([0, d, 0, d]), base)
        e = np.exp(-(d / (2.0 * m)) ** 2 / 2)
        assert (s == 0.0) if want0 else abs(s - e) < 1e-12
    # score at the threshold equals score_th
    assert abs(np.exp(-(2.0 / (2.0 * m)) ** 2 / 2) - 0.5) < 1e-15
    # 3D shared-parent mode: angle + one-way scale-invariant endpoint distance using l1's depths
    l1 = np.array([0, 0, 5, 1, 0, 5, 5.0, 5.0, 0.01, 1.0])
    l2 = l1.copy(); l2[2] += 0.05; l2[5] += 0.05  # shift both endpoints by 0.05 at depth 5 -> d = 0.01
    s = oracle.linker3d_score(cfg, 1, l1, l2)
    assert abs(s - np.exp(-(0.01 / (0.015 * m)) ** 2 / 2)) < 1e-9
    l2[2] += 0.05; l2[5] += 0.05  # d = 0.02 > th_scaleinv 0.015
    assert oracle.linker3d_score(cfg, 1, l1, l2) == 0.0
    # spatial-merging mode needs overlap: disjoint collinear segments score 0
    l3 = l1.copy(); l3[0] += 5; l3[3] += 5
    assert oracle.linker3d_score(cfg, 2, l1, l3) == 0.0
    assert oracle.linker3d_score(cfg, 2, l1, l1) == 1.0


def test_kat_vi_multiview_support(oracle):
    """(vi) scoreOneNode: one GT line seen exactly in 5 views + 1 outlier match.  Every true
    candidate is supported once per OTHER neighbour image (score 1 each), the outlier gets 0;
    best = first strict maximum; valid edges = candidates with score >= fullscore_th."""
    P, Q = np.array([1.0, 4.0, 1.2]), np.array([2.2, 4.2, 1.7])
    centers = [np.array([0.0, 0, 0.3]), np.array([1.0, -0.5, 2.6]), np.array([2.0, 0.2, 0.1]),
               np.array([3.0, -0.3, 2.8]), np.array([3.8, 0.4, 0.2])]
    cams = [look_at_cam(c, 0.5 * (P + Q)) for c in centers]
    segs, off = [], [0]
    for n, c in enumerate(cams):
        segs.append(np.concatenate([project(c, P), project(c, Q)]))
        if n == 2:  # an unrelated segment in image 2, matched as an outlier
            segs.append(np.array([50.0, 60.0, 300.0, 400.0]))
        off.append(len(segs))
    cfg = syn.default_triangulation_cfg(debug_mode=True)
    O = oracle.OracleTriangulator(cfg)
    arr = np.stack(cams)
    O.Init(list(range(5)), arr[:, :4], arr[:, 4:8], arr[:, 8:], off, np.stack(segs))
    matches = {1: np.array([[0, 0]]), 2: np.array([[0, 0], [0, 1]]), 3: np.array([[0, 0]]), 4: np.array([[0, 0]])}
    O.TriangulateImage(0, matches)
    tri = O.get_all_tris()
    n0 = tri["off"][1]
    src = tri["src"][:n0]
    sc = tri["score"][:n0]
    true = ~((src[:, 0] == 2) & (src[:, 1] == 1))
    assert true.sum() == 4
    np.testing.assert_allclose(sc[true], 3.0, atol=1e-6)  # three other neighbour images each
    assert np.all(sc[~true] == 0.0)
    b = O.get_best()
    assert tuple(b["src"][0]) == tuple(src[int(np.argmax(sc))])  # first strict maximum in candidate order
    assert b["score"][0] == sc.max()
    eoff, edges = O.get_valid_edges()
    assert eoff[1] == 4 and set(map(tuple, edges[:4])) == {(0, 0), (1, 0), (2, 0), (3, 0)}


def test_kat_vii_union_find_labels(oracle):
    """(vii) ComputeLineTrackLabelsGreedy on a toy graph: edges sorted by (sim, n1, n2) descending,
    every edge unions, smaller image set hangs under the larger (ties: root2 under root1), labels in
    order of the first node whose direct parent is a final root."""
    node_img = [0, 1, 2, 0, 1, 5, 7]
    edges = [(0.9, 0, 1), (0.8, 1, 2), (0.95, 3, 4), (0.6, 4, 5)]
    lab = oracle.track_labels_greedy(node_img, [e[0] for e in edges], [[e[1], e[2]] for e in edges])
    # order: (3,4) -> parent[4]=3 ; (0,1) -> parent[1]=0 ; (1,2) -> root(1)=0 has 2 imgs > 1 -> parent[2]=0 ;
    # (4,5) -> parent[5]=3.  node 6 isolated.  first labelled root: node 1's parent 0 -> track 0 ; then 3 -> 1
    assert lab.tolist() == [0, 0, 0, 1, 1, 1, -1]
    # tie on sim falls back to the node indices (descending tuple order)
    lab = oracle.track_labels_greedy([0, 1, 0, 1], [0.7, 0.7], [[0, 1], [2, 3]])
    assert lab.tolist() == [0, 0, 1, 1]
    # smaller image set attaches under the larger one: {2 imgs} absorbs {1 img} regardless of edge direction
    lab = oracle.track_labels_greedy([0, 1, 2], [0.9, 0.5], [[1, 2], [0, 1]])
    assert lab.tolist() == [0, 0, 0]


def test_kat_viii_aggregator(oracle):
    """(viii) collinear supports: endpoints at sorted projections [k] and [2n-1-k]; < 4 supports
    returns the best-scored line with the minimum uncertainty."""
    d = np.array([1.0, 2.0, 2.0]) / 3.0
    o = np.array([0.5, -1.0, 2.0])
    spans = [(0.0, 1.0), (0.2, 1.5), (-0.3, 0.9), (0.1, 1.2), (0.4, 2.0)]
    lines = np.array([np.concatenate([o + a * d, o + b * d, [1, 1, 0.01 * (i + 1), 1]]) for i, (a, b) in enumerate(spans)])
    out = oracle.aggregate_line3d_list(lines, np.ones(5), 2)
    ts = sorted([v for ab in spans for v in ab])
    lo, hi = o + ts[2] * d, o + ts[-3] * d
    got = (out[:3], out[3:6])
    ok = (np.allclose(got[0], lo, atol=1e-10) and np.allclose(got[1], hi, atol=1e-10)) or \
         (np.allclose(got[0], hi, atol=1e-10) and np.allclose(got[1], lo, atol=1e-10))
    assert ok and abs(out[6] - 0.01) < 1e-15
    out3 = oracle.aggregate_line3d_list(lines[:3], np.array([0.5, 2.0, 1.0]), 2)
    np.testing.assert_array_equal(out3[:6], lines[1, :6])
    assert out3[6] == 0.01


def test_sensitivity_and_ranges(oracle, two_views):
    P, Q, c1, c2 = two_views
    C1 = oracle.cam_center(c1)
    l = np.concatenate([P, Q, [1, 1, -1, 1]])
    mid2d = 0.5 * (project(c1, P) + project(c1, Q))
    ray = oracle.cam_ray_direction(c1, mid2d)
    dirv = (Q - P) / np.linalg.norm(Q - P)
    expect = 90 - np.degrees(np.arccos(abs(dirv @ ray)))
    assert abs(oracle.line3d_sensitivity(l, c1) - expect) < 1e-9
    # ranges filter drops candidates outside the box (functions.cc:8-26)
    s1 = np.concatenate([project(c1, P), project(c1, Q)]); s2 = np.concatenate([project(c2, P), project(c2, Q)])
    cfg = syn.default_triangulation_cfg(debug_mode=True)
    for rng_, n in (((np.zeros(3), np.full(3, 10.0)), 1), ((np.zeros(3), np.array([10, 10, 1.5])), 0)):
        O = oracle.OracleTriangulator(cfg)
        O.SetRanges(rng_)
        O.Init([0, 1], np.stack([c1[:4], c2[:4]]), np.stack([c1[4:8], c2[4:8]]), np.stack([c1[8:], c2[8:]]), [0, 1, 2],
               np.stack([s1, s2]))
        O.TriangulateImage(0, {1: np.array([[0, 0]], np.int32)})
        assert O.get_num_tris()[0] == n


def test_out_of_index_match_raises(oracle, two_views):
    P, Q, c1, c2 = two_views
    s = np.array([10.0, 10.0, 200.0, 50.0])
    O = oracle.OracleTriangulator(syn.default_triangulation_cfg())
    O.Init([0, 1], np.stack([c1[:4], c2[:4]]), np.stack([c1[4:8], c2[4:8]]), np.stack([c1[8:], c2[8:]]), [0, 1, 2],
           np.stack([s, s]))
    with pytest.raises(RuntimeError, match="IndexError"):
        O.TriangulateImage(0, {1: np.array([[3, 0]], np.int32)})


def test_kat_vp_direction_triangulation(oracle):
    """KAT (xii): VP-guided proposal.  A noise-free scene: with the true 3D direction as VP of l1
    (vp = K R d), triangulate_line_with_direction reproduces the GT segment's endpoints (the
    rays of l1's endpoints meet the plane of l2 where the GT line is); a VP on neither line adds nothing."""
    import numpy as np
    from limap_amd import synthetic as syn
    K4 = np.array([500.0, 500.0, 320.0, 240.0])
    # two cameras looking down +z, second shifted in x and y
    q = np.array([[1.0, 0, 0, 0], [1.0, 0, 0, 0]])
    t = np.array([[0.0, 0, 0], [-0.6, -0.25, 0.0]])
    P, Q = np.array([-0.4, 0.1, 4.0]), np.array([0.5, 0.35, 5.0])

    def proj(X, tt):
        Xc = X + tt
        return np.array([K4[0] * Xc[0] / Xc[2] + K4[2], K4[1] * Xc[1] / Xc[2] + K4[3]])
    segs = np.array([[*proj(P, t[0]), *proj(Q, t[0])], [*proj(P, t[1]), *proj(Q, t[1])]])
    d = (Q - P) / np.linalg.norm(Q - P)
    Kmat = np.array([[K4[0], 0, K4[2]], [0, K4[1], K4[3]], [0, 0, 1.0]])
    vp = Kmat @ d
    cfg = syn.default_triangulation_cfg(debug_mode=True)
    cfg.update(use_vp=True, disable_algebraic_triangulation=True, min_num_outer_edges=0)
    for lab0, lab1, expect in ((0, -1, 1), (-1, 0, 1), (0, 0, 2), (-1, -1, 0)):
        O = oracle.OracleTriangulator(cfg, faithful=True)
        O.Init([7, 9], np.tile(K4, (2, 1)), q, t, [0, 1, 2], segs)
        O.InitVPResults({7: ([lab0], [vp]), 9: ([lab1], [vp])})
        O.TriangulateImage(7, {9: np.array([[0, 0]], np.int32)})
        a = O.get_all_tris()
        assert a["off"][1] - a["off"][0] == expect
        for k in range(expect):
            np.testing.assert_allclose(a["line"][k, :3], P, rtol=0, atol=1e-9)
            np.testing.assert_allclose(a["line"][k, 3:6], Q, rtol=0, atol=1e-9)


def test_kat_many_points_triangulation(oracle):
    """KAT (xiii): many-points proposal (fit + Pluecker projection).  Noise-free: three 3D points exactly on
    the GT segment, shared by l1 and l2 -> the fitted infinite line IS the GT line and the endpoints of the
    candidate are where l1's endpoint rays meet it, i.e. the GT endpoints; with SfM points given they are
    used as they are, without them the points are triangulated from the two views."""
    import numpy as np
    from limap_amd import synthetic as syn
    K4 = np.array([500.0, 500.0, 320.0, 240.0])
    q = np.array([[1.0, 0, 0, 0], [1.0, 0, 0, 0]])
    t = np.array([[0.0, 0, 0], [-0.6, -0.25, 0.0]])
    P, Q = np.array([-0.4, 0.1, 4.0]), np.array([0.5, 0.35, 5.0])

    def proj(X, tt):
        Xc = X + tt
        return np.array([K4[0] * Xc[0] / Xc[2] + K4[2], K4[1] * Xc[1] / Xc[2] + K4[3]])
    segs = np.array([[*proj(P, t[0]), *proj(Q, t[0])], [*proj(P, t[1]), *proj(Q, t[1])]])
    pts3 = {10 + k: P + s * (Q - P) for k, s in enumerate((0.2, 0.5, 0.9))}
    bp = {}
    for n, img in enumerate((7, 9)):
        bp[img] = dict(point_ids=np.arange(3), xy=np.array([proj(pts3[10 + k], t[n]) for k in range(3)]),
                       point3D_ids=np.array([10, 11, 12]), line_points=[[0, 1, 2]])
    cfg = syn.default_triangulation_cfg(debug_mode=True)
    cfg.update(disable_algebraic_triangulation=True, disable_one_point_triangulation=True, min_num_outer_edges=0)
    for with_sfm in (True, False):
        O = oracle.OracleTriangulator(cfg, faithful=True)
        O.Init([7, 9], np.tile(K4, (2, 1)), q, t, [0, 1, 2], segs)
        O.SetBipartites2d(bp)
        if with_sfm:
            O.SetSfMPoints(pts3)
        O.TriangulateImage(7, {9: np.array([[0, 0]], np.int32)})
        a = O.get_all_tris()
        assert a["off"][1] - a["off"][0] == 1
        np.testing.assert_allclose(a["line"][0, :3], P, rtol=0, atol=1e-8)
        np.testing.assert_allclose(a["line"][0, 3:6], Q, rtol=0, atol=1e-8)
    # a single shared point is not enough for the fit; no bipartites -> no proposal
    bp1 = {k: dict(v, line_points=[[0]]) for k, v in bp.items()}
    O = oracle.OracleTriangulator(cfg, faithful=True)
    O.Init([7, 9], np.tile(K4, (2, 1)), q, t, [0, 1, 2], segs)
    O.SetBipartites2d(bp1); O.SetSfMPoints(pts3)
    O.TriangulateImage(7, {9: np.array([[0, 0]], np.int32)})
    assert O.get_all_tris()["off"][-1] == 0


def test_kat_one_point_triangulation(oracle):
    """KAT (xiv): one-point proposal.  Noise-free: a 3D point exactly on the GT segment -- the line through
    its projection onto plane 1 that meets l1's endpoint rays where they cross plane 2 has zero error, so
    the minimiser is the GT segment itself.  Also checks the solver on a perturbed point against a brute-force
    scan of the same objective."""
    import numpy as np
    from limap_amd import synthetic as syn
    K4 = np.array([500.0, 500.0, 320.0, 240.0])
    q = np.array([[1.0, 0, 0, 0], [1.0, 0, 0, 0]])
    t = np.array([[0.0, 0, 0], [-0.6, -0.25, 0.0]])
    P, Q = np.array([-0.4, 0.1, 4.0]), np.array([0.5, 0.35, 5.0])

    def proj(X, tt):
        Xc = X + tt
        return np.array([K4[0] * Xc[0] / Xc[2] + K4[2], K4[1] * Xc[1] / Xc[2] + K4[3]])
    segs = np.array([[*proj(P, t[0]), *proj(Q, t[0])], [*proj(P, t[1]), *proj(Q, t[1])]])
    cfg = syn.default_triangulation_cfg(debug_mode=True)
    cfg.update(disable_algebraic_triangulation=True, disable_many_points_triangulation=True, min_num_outer_edges=0)
    for s_on_line, off in ((0.3, 0.0), (0.6, 0.0), (0.5, 0.02)):
        X = P + s_on_line * (Q - P) + off * np.array([0.3, -1.0, 0.2])
        bp = {}
        for n, img in enumerate((7, 9)):
            bp[img] = dict(point_ids=[0], xy=[proj(X, t[n])], point3D_ids=[42], line_points=[[0]])
        O = oracle.OracleTriangulator(cfg, faithful=True)
        O.Init([7, 9], np.tile(K4, (2, 1)), q, t, [0, 1, 2], segs)
        O.SetBipartites2d(bp); O.SetSfMPoints({42: X})
        O.TriangulateImage(7, {9: np.array([[0, 0]], np.int32)})
        a = O.get_all_tris()
        assert a["off"][1] - a["off"][0] == 1
        s3, e3 = a["line"][0, :3], a["line"][0, 3:6]
        if off == 0.0:
            np.testing.assert_allclose(s3, P, rtol=0, atol=1e-8)
            np.testing.assert_allclose(e3, Q, rtol=0, atol=1e-8)
        else:
            # brute force over lambda1: the endpoints lie on l1's rays (camera 1 at the origin, identity pose),
            # are collinear with the point's projection onto plane 1, and minimise the summed squared distance
            # to plane 2
            r1, r2 = P / np.linalg.norm(P), Q / np.linalg.norm(Q)
            n1 = np.cross(r1, r2); n1 /= np.linalg.norm(n1)
            Xp = X - n1 * (n1 @ X)
            C2 = -t[1]
            n2 = np.cross(P - C2, Q - C2); n2 /= np.linalg.norm(n2)

            def err_of(l1):
                A = l1 * r1
                # B on ray 2 collinear with A and Xp:  solve A + s (Xp - A) = l2 r2 in plane 1
                M = np.stack([Xp - A, -r2], 1)
                sol, *_ = np.linalg.lstsq(M, -A, rcond=None)
                l2 = sol[1]
                B = l2 * r2
                return (n2 @ (A - C2)) ** 2 + (n2 @ (B - C2)) ** 2, l2
            l1s = np.linspace(0.5 * np.linalg.norm(P), 1.5 * np.linalg.norm(P), 20001)
            errs = np.array([err_of(x)[0] if err_of(x)[1] > 0 else np.inf for x in l1s])
            best = l1s[int(np.argmin(errs))]
            assert abs(np.linalg.norm(s3) - best) < 2 * (l1s[1] - l1s[0])
            assert errs.min() >= err_of(np.linalg.norm(s3))[0] - 1e-12


def test_one_point_restated_problem_equals_generated_solver(oracle):
    """The middle link of the one-point chain (HIP == restated solver bit for bit in tests/test_gpu_points.py; generated
    solver == oracle/_ref bit for bit in tests/test_oracle_vs_ref.py): the restated optimisation problem against the
    reference's generated polynomials on the CPU.  Same stationary points, same selection rule => same line, to the
    conditioning of the GENERATED form: its expanded coefficients cancel over many digits, so where the two differ by
    more than rounding, the generated solver's own answer moves by a comparable amount when the known point is
    perturbed in its last bits -- the difference is the reference's noise, not a second solution."""
    import numpy as np
    from limap_amd import synthetic as syn
    sc = syn.make_scene(n_views=8, n_segs=60, n_neighbors=4, seed=43)
    rng = np.random.default_rng(11)
    rel, noisy, sentinel_mismatch = [], 0, 0
    try:
        for trial in range(1500):
            a, b = rng.choice(sc.n_images, 2, replace=False)
            cam1, cam2 = sc.cam11(int(a)), sc.cam11(int(b))
            s1 = sc.segs_of(int(a))[rng.integers(0, 40)]
            s2 = sc.segs_of(int(b))[rng.integers(0, 40)]
            t = rng.uniform(0.1, 0.9)
            px = (1 - t) * s1[0:2] + t * s1[2:4] + rng.normal(0, 0.5, 2)
            point = oracle.cam_center(cam1) + oracle.cam_ray_direction(cam1, px) * rng.uniform(1.5, 6.0)
            oracle.set_one_point_solver(True)
            lg = oracle.triangulate_line_with_one_point(s1, cam1, s2, cam2, point)
            oracle.set_one_point_solver(False)
            lr = oracle.triangulate_line_with_one_point(s1, cam1, s2, cam2, point)
            if (lg[9] < 0) != (lr[9] < 0):
                sentinel_mismatch += 1  # a root at the edge of the cheirality test
                continue
            if lg[9] < 0:
                continue
            scale = max(1.0, float(np.abs(lg[:6]).max()))
            d = float(np.abs(lg[:8] - lr[:8]).max()) / scale
            rel.append(d)
            if d > 1e-11:
                # the generated solver's own sensitivity: the point moved by a few ulps
                oracle.set_one_point_solver(True)
                moved = 0.0
                for k in range(4):
                    pp = point * (1.0 + 4e-16 * np.array([(-1) ** k, (-1) ** (k // 2), 1.0]))
                    lp = oracle.triangulate_line_with_one_point(s1, cam1, s2, cam2, pp)
                    if lp[9] >= 0:
                        moved = max(moved, float(np.abs(lp[:8] - lg[:8]).max()) / scale)
                noisy += 1
                assert moved > 1e-3 * d, (trial, d, moved)  # ill-conditioned there, by its own evidence
    finally:
        oracle.set_one_point_solver(True)
    rel = np.array(rel)
    assert len(rel) > 500 and sentinel_mismatch <= 3, (len(rel), sentinel_mismatch)
    assert np.median(rel) < 1e-14 and np.percentile(rel, 95) < 1e-11 and rel.max() < 1e-6, \
        (np.median(rel), np.percentile(rel, 95), rel.max(), noisy)
