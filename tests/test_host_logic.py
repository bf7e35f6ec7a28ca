"""Host-side logic that needs no GPU: the synthetic generator, the value types, the sharding
helpers, the config mirror."""
import numpy as np
import pytest

from limap_amd import base, dist as ltdist, synthetic as syn


def test_scene_is_deterministic_and_well_formed():
    a = syn.make_scene(n_views=12, n_segs=50, n_neighbors=5, seed=4)
    b = syn.make_scene(n_views=12, n_segs=50, n_neighbors=5, seed=4)
    assert np.array_equal(a.segs, b.segs) and np.array_equal(a.qvec, b.qvec)
    assert a.segs.shape == (12 * 50, 4) and a.seg_off[-1] == 600
    # endpoints slide +-15 % along the line after clipping, so they may poke slightly outside the image
    assert np.all(a.segs[:, [0, 2]] >= -0.3 * syn.W_IMG) and np.all(a.segs[:, [0, 2]] <= 1.3 * syn.W_IMG)
    lens = np.linalg.norm(a.segs[:, 2:] - a.segs[:, :2], axis=1).reshape(12, 50)
    assert np.all(np.diff(lens, axis=1) <= 1e-9), "segments are sorted by length (take_longest_k)"
    np.testing.assert_allclose(np.linalg.norm(a.qvec, axis=1), 1.0, atol=1e-12)
    for i, nbs in a.neighbors.items():
        assert i not in nbs and len(nbs) <= 5 and len(set(nbs)) == len(nbs)
    m = a.matches_of(3)
    m2 = a.matches_of(3)
    assert sorted(m.keys()) == sorted(a.neighbors[3])
    for k in m:
        assert m[k].dtype == np.int32 and m[k].shape[1] == 2 and np.array_equal(m[k], m2[k])
        assert m[k][:, 0].max() < 50 and m[k][:, 1].max() < 50
        assert np.all(np.diff(m[k][:, 0]) >= 0), "rows grouped by line id"
    c = syn.make_scene(n_views=12, n_segs=50, n_neighbors=5, seed=5)
    assert not np.array_equal(a.segs, c.segs)


def test_true_matches_are_geometrically_consistent():
    sc = syn.make_scene(n_views=10, n_segs=60, n_neighbors=4, seed=1)
    i, nb = 2, sc.neighbors[2][0]
    g1, g2 = sc.gt_ids[sc.seg_off[i]:sc.seg_off[i + 1]], sc.gt_ids[sc.seg_off[nb]:sc.seg_off[nb + 1]]
    m = sc.matches_of(i)[nb]
    same = (g1[m[:, 0]] >= 0) & (g1[m[:, 0]] == g2[m[:, 1]])
    vis_both = np.isin(g1[g1 >= 0], g2[g2 >= 0]).sum()
    # (a random distractor may coincide with the true match, hence count lines, not rows)
    assert len(np.unique(m[same, 0])) == vis_both > 0


def test_value_types():
    l = base.Line2d([0, 0], [3, 4])
    assert l.length() == 5 and np.allclose(l.direction(), [0.6, 0.8]) and l.as_array().shape == (2, 2)
    l3 = base.Line3d.from10(np.array([0, 0, 0, 0, 0, 2, 5, 6, 0.1, 1.0]))
    assert l3.length() == 2 and l3.uncertainty == 0.1 and np.allclose(l3.depths, [5, 6]) and l3.score == 1.0
    tr = base.LineTrack()
    tr.image_id_list, tr.line_id_list = [3, 1, 3], [0, 2, 5]
    tr.line2d_list = [l, l, l]
    assert tr.count_lines() == 3 and tr.count_images() == 2 and tr.GetSortedImageIds() == [1, 3]
    d = tr.as_dict()
    assert set(d) >= {"line", "image_id_list", "line_id_list", "node_id_list", "score_list", "line2d_list", "line3d_list", "active"}
    ic = base.ImageCollection.from_arrays([5, 2], np.tile([700.0, 700, 400, 300], (2, 1)),
                                          np.tile([1.0, 0, 0, 0], (2, 1)), np.zeros((2, 3)))
    assert ic.get_img_ids() == [2, 5] and ic.NumImages() == 2 and np.allclose(ic.camview(5).R(), np.eye(3))


@pytest.mark.parametrize("n,world", [(100, 1), (100, 8), (7, 3), (3, 8)])
def test_shard_bounds_cover_everything_once(n, world):
    b = ltdist.shard_bounds(n, world)
    assert b[0] == 0 and b[-1] == n and len(b) == world + 1 and all(x <= y for x, y in zip(b, b[1:]))
    ids = np.arange(n) * 3 + 1
    got = np.concatenate([ltdist.shard_images(ids, r, world) for r in range(world)])
    assert np.array_equal(got, ids)


def test_shard_bounds_balance_by_weight():
    w = np.array([1] * 10 + [10] * 10)
    b = ltdist.shard_bounds(20, 2, w)
    loads = [w[b[r]:b[r + 1]].sum() for r in range(2)]
    assert abs(loads[0] - loads[1]) <= 10


def test_config_mirror_class():
    pytest.importorskip("ctypes")
    from limap_amd import triangulation as tri
    c = tri.GlobalLineTriangulatorConfig(syn.default_triangulation_cfg())
    assert c.fullscore_th == 1.0 and c.merging_strategy == "greedy" and c.linker3d_config["th_scaleinv"] == 0.015
    c.var2d = 4.0
    c.linker2d_config = {"th_perp": 3.0}
    c.merging_strategy = "avg"
    assert c.var2d == 4.0 and c.linker2d_config["th_perp"] == 3.0 and c._s.merging_strategy == 2
    with pytest.raises(AttributeError):
        c.sensitivity_threshold_typo = 1  # like the pybind class: unknown attributes are rejected


def test_pymarshal_helper_matches_python_path():
    """limap_amd/_lt_pymarshal (CPython helper): hands the data pointers / row counts of C-contiguous int32
    (K,2) arrays to the C entry point; anything else returns None (the Python path converts by copy)."""
    import ctypes as C
    from limap_amd import triangulation as tri
    if tri._fast is None:
        pytest.skip("helper not built")
    seen = {}
    proto = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int32),
                        C.POINTER(C.POINTER(C.c_int32)), C.POINTER(C.c_int64))

    def fake(ctx, img_id, n_nb, nb, rows, cnt):
        seen["ctx"], seen["img"], seen["n"] = ctx, img_id, n_nb
        seen["nb"] = [nb[k] for k in range(n_nb)]
        seen["cnt"] = [cnt[k] for k in range(n_nb)]
        seen["first"] = [(rows[k][0], rows[k][1]) if cnt[k] else None for k in range(n_nb)]
        return 7
    cb = proto(fake)
    addr = C.cast(cb, C.c_void_p).value
    m = {5: np.array([[1, 2], [3, 4]], np.int32), 2: np.zeros((0, 2), np.int32), np.int64(9): np.array([[7, 8]], np.int32)}
    assert tri._fast.triangulate_image_rows(addr, 1234, 42, m) == 7
    assert seen == {"ctx": 1234, "img": 42, "n": 3, "nb": [5, 2, 9], "cnt": [2, 0, 1], "first": [(1, 2), None, (7, 8)]}
    for bad in (np.array([[1, 2]], np.int64), np.array([[1, 2, 3]], np.int32), np.zeros((4, 4), np.int32)[:, :2],
                np.zeros(0, np.int32), [[1, 2]]):
        assert tri._fast.triangulate_image_rows(addr, 1234, 42, {1: bad}) is None
    assert tri._fast.triangulate_image_rows(addr, 1234, 42, {"x": m[5]}) is None


def test_lazy_line_tracks_materialise_on_access():
    """The tracks ComputeLineTracks() returns build their member lists on first access (like the reference's
    pybind wrappers of C++ LineTracks) and then behave like plain LineTrack objects."""
    from limap_amd import triangulation as tri
    t = dict(off=np.array([0, 2, 5], np.int64),
             line=np.array([[0, 0, 0, 1, 1, 1, 0.5], [1, 2, 3, 4, 5, 6, 0.25]], float),
             image_ids=np.array([10, 11, 10, 12, 13], np.int32), line_ids=np.array([0, 1, 1, 0, 0], np.int32),
             node_ids=np.array([0, 1, 2, 3, 4], np.int32), scores=np.array([1.0, 2.0, 3.0, 4.0, 5.0]),
             line3d=np.arange(50, dtype=float).reshape(5, 10))
    segs = {i: np.arange(8, dtype=float).reshape(2, 4) + i for i in (10, 11, 12, 13)}
    tr = [tri._LazyLineTrack(t, n, segs) for n in range(2)]
    assert tr[0].count_lines() == 2 and tr[1].count_lines() == 3 and "image_id_list" not in tr[1].__dict__
    assert tr[1].image_id_list == [10, 12, 13] and tr[1].line_id_list == [1, 0, 0] and tr[1].node_id_list == [2, 3, 4]
    assert tr[1].score_list == [3.0, 4.0, 5.0] and tr[1].count_images() == 3 and tr[1].HasImage(12)
    assert np.array_equal(tr[1].line.start, [1, 2, 3]) and tr[1].line.uncertainty == 0.25
    assert np.array_equal(tr[1].line2d_list[0].start, segs[10][1, 0:2])
    assert np.array_equal(tr[0].line3d_list[1].end, t["line3d"][1, 3:6])
    assert tr[0].line3d_list[1].uncertainty == t["line3d"][1, 8] and tr[0].line3d_list[1].depths[1] == t["line3d"][1, 7]
    d = tr[1].as_dict()
    assert d["image_id_list"] == [10, 12, 13] and len(d["line2d_list"]) == 3 and d["active"] is True
    with pytest.raises(AttributeError):
        tr[0].no_such_field


def test_lazy_line_tracks_copy_and_pickle():
    """copy / deepcopy / pickle of the tracks ComputeLineTracks() returns (the reference's LineTrack
    is copyable and picklable) -- no recursion through __getattr__, fields preserved."""
    import copy
    import pickle
    from limap_amd import triangulation as tri
    from limap_amd.base import LineTrack
    t = dict(off=np.array([0, 2, 5], np.int64),
             line=np.array([[0, 0, 0, 1, 1, 1, 0.5], [1, 2, 3, 4, 5, 6, 0.25]], float),
             image_ids=np.array([10, 11, 10, 12, 13], np.int32), line_ids=np.array([0, 1, 1, 0, 0], np.int32),
             node_ids=np.array([0, 1, 2, 3, 4], np.int32), scores=np.array([1.0, 2.0, 3.0, 4.0, 5.0]),
             line3d=np.arange(50, dtype=float).reshape(5, 10))
    segs = {i: np.arange(8, dtype=float).reshape(2, 4) + i for i in (10, 11, 12, 13)}
    tr = tri._LazyLineTrack(t, 1, segs)
    for clone in (copy.copy(tr), copy.deepcopy(tr), pickle.loads(pickle.dumps(tr))):
        assert isinstance(clone, LineTrack)
        assert clone.image_id_list == [10, 12, 13] and clone.line_id_list == [1, 0, 0]
        assert clone.node_id_list == [2, 3, 4] and clone.score_list == [3.0, 4.0, 5.0] and clone.active is True
        assert np.array_equal(clone.line.end, [4, 5, 6]) and clone.count_lines() == 3
        assert np.array_equal(clone.line3d_list[2].start, t["line3d"][4, 0:3])
    arr = np.empty(1, object); arr[0] = tr
    assert pickle.loads(pickle.dumps(arr))[0].count_images() == 3   # what np.save(allow_pickle) does
    empty = tri._LazyLineTrack.__new__(tri._LazyLineTrack)            # an instance with an empty __dict__
    with pytest.raises(AttributeError):
        empty.line


def test_track_report_counts_images_not_lines():
    """limap's track report (visualize/trackvis/base.py:25-50): N_k counts tracks by supporting IMAGES."""
    from limap_amd.base import track_report
    rep = track_report([0, 2, 5, 5, 9], [1, 2, 3, 3, 4, 1, 2, 3, 4])
    assert rep["N2"] == 3 and rep["N4"] == 1 and rep["N6"] == 0
    assert rep["avg_supporting_images_ge3"] == 4.0 and rep["avg_supporting_lines_ge4"] == 4.0
    assert track_report([0], [])["N2"] == 0


def test_aggregator_matches_the_oracle_bit_for_bit_including_orientation():
    """Aggregator::aggregate_line3d_list (merging/aggregator.cc:53-101): the product's host tail against the oracle (which
    is pinned to oracle/_ref) on random bundles of near-parallel segments -- centre, principal axis, its SIGN (which end
    of the track line is `start`), the outlier-trimmed extent and the uncertainty, bit for bit.  No device involved."""
    import ctypes as C
    from limap_amd import _capi
    from oracle import oracle as ora
    L = _capi.load_library()
    rng = np.random.default_rng(5)
    dp = C.POINTER(C.c_double)
    for trial in range(300):
        n = int(rng.integers(1, 40))
        d = rng.normal(size=3); d /= np.linalg.norm(d)
        c0 = rng.normal(size=3) * 5
        lines = np.zeros((n, 10))
        for i in range(n):
            a, b = np.sort(rng.uniform(-3, 3, 2))
            flip = rng.random() < 0.3          # supporting lines come in either orientation
            s, e = c0 + a * d + rng.normal(size=3) * 0.02, c0 + b * d + rng.normal(size=3) * 0.02
            lines[i, :3], lines[i, 3:6] = (e, s) if flip else (s, e)
            lines[i, 6:8] = rng.uniform(1, 9, 2)
            lines[i, 8] = rng.uniform(0.01, 0.2)
            lines[i, 9] = 1.0
        scores = rng.uniform(0, 10, n)
        num_outliers = int(rng.integers(0, 3)) if n >= 4 else 2
        if n >= 4:
            num_outliers = min(num_outliers, n - 1)
        out = np.zeros(7)
        rc = L.lt_fn_aggregate_line3d_list(n, lines.ctypes.data_as(dp), scores.ctypes.data_as(dp), num_outliers, out.ctypes.data_as(dp))
        assert rc == 0
        want = ora.aggregate_line3d_list(lines, scores, num_outliers)
        assert np.array_equal(out, np.asarray(want).ravel()[:7]), (trial, n, out, want)


def test_principal_axis_against_an_independent_svd_and_how_stable_its_sign_is():
    """The one third-party assumption nothing in this container can compile away: Eigen's JacobiSVD decides which end of an
    aggregated track is `start` (merging/aggregator.cc:76-78: `matrixV().col(0)`), and lt_svd.h / oracle/eigen_svd_ref.h are
    restatements of Eigen 3.4's procedure, not Eigen.  Two properties that do not depend on either restatement:
      (a) the AXIS equals numpy's (LAPACK) first right singular vector of the centred endpoint matrix up to sign;
      (b) the SIGN is stable: perturbing the endpoints by 1e-9 relative -- far more than any reduction-order or FMA
          difference between a real Eigen build and the restatement could -- flips it for (almost) no track.  The measured
          rate is the size of what remains assumed ("Eigen behaviours assumed")."""
    import ctypes as C
    from limap_amd import _capi
    L = _capi.load_library()
    rng = np.random.default_rng(17)
    dp = C.POINTER(C.c_double)

    def agg(lines, scores):
        out = np.zeros(7)
        assert L.lt_fn_aggregate_line3d_list(len(lines), lines.ctypes.data_as(dp), scores.ctypes.data_as(dp), 0, out.ctypes.data_as(dp)) == 0
        return out

    n_tracks, n_flips, worst_axis = 1500, 0, 0.0
    for _ in range(n_tracks):
        n = int(rng.integers(4, 30))
        d = rng.normal(size=3); d /= np.linalg.norm(d)
        c0 = rng.normal(size=3) * 5
        noise = 10.0 ** rng.uniform(-4, -1)
        lines = np.zeros((n, 10))
        for i in range(n):
            a, b = np.sort(rng.uniform(-3, 3, 2))
            s, e = c0 + a * d + rng.normal(size=3) * noise, c0 + b * d + rng.normal(size=3) * noise
            lines[i, :3], lines[i, 3:6] = (e, s) if rng.random() < 0.5 else (s, e)
            lines[i, 6:8] = rng.uniform(1, 9, 2); lines[i, 8] = 0.05; lines[i, 9] = 1.0
        scores = rng.uniform(0, 10, n)
        out = agg(lines, scores)
        axis = out[3:6] - out[:3]
        axis /= np.linalg.norm(axis)
        pts = np.concatenate([lines[:, :3], lines[:, 3:6]], 0)
        v = np.linalg.svd(pts - pts.mean(0), full_matrices=False)[2][0]
        worst_axis = max(worst_axis, 1.0 - abs(float(axis @ v)))
        pert = lines.copy()
        pert[:, :6] *= 1.0 + 1e-9 * rng.uniform(-1, 1, (n, 6))
        out2 = agg(pert, scores)
        n_flips += int(float((out2[3:6] - out2[:3]) @ axis) < 0)
    assert worst_axis < 1e-9, worst_axis
    print(f"principal-axis sign flips under a 1e-9 perturbation: {n_flips} of {n_tracks}")
    assert n_flips <= n_tracks // 200, n_flips


def test_match_row_pass_vector_paths_equal_plain_numpy():
    """The host pass over a block of match rows (lt_rows.h: staged word line | neighbour line << 16, column maxima as
    unsigned, sortedness) -- scalar, AVX2 and AVX-512 forms against numpy, on ragged lengths around the vector widths,
    unaligned starts, negative and out-of-range ids and unsorted blocks (base_line_triangulator.cc:82-98)."""
    from limap_amd import _capi
    L = _capi.load_library()
    rng = np.random.default_rng(5)
    lengths = list(range(0, 70)) + [127, 128, 129, 1000, 5003]
    for n in lengths:
        for variant in range(4):
            rows = np.stack([np.sort(rng.integers(0, 500, n)), rng.integers(0, 65536, n)], 1).astype(np.int32)
            if variant == 1 and n > 2:   # one inversion somewhere (also at the vector seams)
                k = int(rng.integers(1, n))
                rows[k, 0] = rows[k - 1, 0] - 1
            if variant == 2 and n > 0:   # a negative / too large id in either column
                rows[int(rng.integers(0, n)), int(rng.integers(0, 2))] = int(rng.choice([-1, -70000, 65536, 2**31 - 1]))
            if variant == 3 and n > 1:   # inversion between the last two rows
                rows[n - 1, 0] = rows[n - 2, 0] - 3
            buf = np.zeros(2 * n + 3, np.int32)   # odd offset: the block does not start on a vector boundary
            src = buf[1:1 + 2 * n].reshape(n, 2)
            src[:] = rows
            u = rows.astype(np.int64) & 0xFFFFFFFF
            want = ((u[:, 0] & 0xFFFF) | ((u[:, 1] << 16) & 0xFFFFFFFF)).astype(np.uint32)
            want_stats = [int(u[:, 0].max()) if n else 0, int(u[:, 1].max()) if n else 0,
                          int(n > 1 and bool((rows[1:, 0] < rows[:-1, 0]).any()))]
            for level in (1, 2, 3, 0):
                out = np.full(n + 1, 0xABCDABCD, np.uint32)
                stats = np.zeros(3, np.uint32)
                rc = L.lt_fn_pack_match_rows(src.ctypes.data, n, out.ctypes.data, stats.ctypes.data, level)
                assert rc == 0
                assert np.array_equal(out[:n], want), (n, variant, level)
                assert out[n] == 0xABCDABCD, "wrote past the block"
                assert stats.tolist() == want_stats, (n, variant, level, stats.tolist(), want_stats)


def test_compressed_match_rows_decode_to_the_rows():
    """The compressed block form the match rows cross PCIe in (lt_rows.h, round 4): 16 bits of neighbour line per row + one
    "new line" bit, for blocks sorted by line with steps of 0 / +1.  Scalar and AVX-512 forms against numpy: the decoded
    rows equal the input, the column maxima / sortedness as in the plain pass, and every other shape (a gap in the line
    ids, an inversion) is reported irregular."""
    from limap_amd import _capi
    L = _capi.load_library()
    rng = np.random.default_rng(11)
    lengths = list(range(0, 70)) + [127, 128, 129, 130, 1000, 5003]
    for n in lengths:
        for variant in range(5):
            # regular: every line 0..M-1 present with 1..k rows, starting at an arbitrary first line
            reps = rng.integers(1, 12, max(n, 1))
            lines = np.repeat(np.arange(len(reps)), reps)[:n] + int(rng.integers(0, 40))
            rows = np.stack([lines, rng.integers(0, 65536, n)], 1).astype(np.int32)
            irregular = False
            if variant == 1 and n > 2:   # a line without rows: a step of +2 somewhere
                k = int(rng.integers(1, n))
                rows[k:, 0] += 1 if rows[k, 0] != rows[k - 1, 0] else 2
                irregular = True
            if variant == 2 and n > 2:   # an inversion
                k = int(rng.integers(1, n))
                rows[k, 0] = rows[k - 1, 0] - 1
                irregular = True
            if variant == 3 and n > 0:   # a too large neighbour id: reported through the maxima, like the plain pass
                rows[int(rng.integers(0, n)), 1] = 70000
            if variant == 4 and n > 1:   # all rows of one line
                rows[:, 0] = 7
            nw = int(L.lt_fn_compressed_block_words(n))
            assert nw % 2 == 0 and nw == ((((n + 1) // 2) + 1) & ~1) + 2 * ((n + 63) // 64)
            buf = np.zeros(2 * n + 3, np.int32)
            src = buf[1:1 + 2 * n].reshape(n, 2)
            src[:] = rows
            u = rows.astype(np.int64) & 0xFFFFFFFF
            for level in (1, 3, 0):
                out = np.full(nw + 2, 0xABCDABCD, np.uint32)
                assert out.ctypes.data % 8 == 0
                stats = np.zeros(4, np.uint32)
                assert L.lt_fn_pack_match_rows_compressed(src.ctypes.data, n, out.ctypes.data, stats.ctypes.data, level) == 0
                assert out[nw] == 0xABCDABCD and out[nw + 1] == 0xABCDABCD, "wrote past the block"
                assert stats[0] == (int(u[:, 0].max()) if n else 0) and stats[1] == (int(u[:, 1].max()) if n else 0)
                assert stats[2] == int(n > 1 and bool((rows[1:, 0] < rows[:-1, 0]).any()))
                d = np.diff(rows[:, 0].astype(np.int64)) if n > 1 else np.zeros(0, np.int64)
                assert bool(stats[3]) == bool(((d < 0) | (d > 1)).any()), (n, variant, level)
                assert bool(stats[3]) == irregular or variant in (3, 4)
                if stats[3] or variant == 3:
                    continue
                # decode like k_expand_rows: line = first line + number of "new line" bits up to the row
                nbw = (((n + 1) // 2) + 1) & ~1
                nb = out[:nbw].view(np.uint16)[:n]
                bits = out[nbw:nw].view(np.uint64)
                bit = np.array([(int(bits[r >> 6]) >> (r & 63)) & 1 for r in range(n)], np.int64)
                line = (rows[0, 0] if n else 0) + np.cumsum(bit)
                assert np.array_equal(line, rows[:, 0]) and np.array_equal(nb, rows[:, 1].astype(np.uint16)), (n, variant, level)
