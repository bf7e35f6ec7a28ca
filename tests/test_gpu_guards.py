"""-m gpu: the fast paths are conservative.  Stage A (`gate3`) only ever decides when the reference's
outcome is certain, and the scoring sweep only skips pairs that cannot score; switching either shortcut
off (everything then runs through the reference-exact expressions) must not change a single bit of the
results.  Also threshold / option variants of the gates against the oracle."""
import os

import numpy as np
import pytest

from limap_amd import synthetic as syn

from helpers import (compare_best, compare_candidates, compare_tracks, compare_valid_edges, run_oracle, small_scene,
                     run_product)

pytestmark = pytest.mark.gpu


def _results(T):
    ctx = T.context()
    allt = ctx.get_all_tris()
    best = ctx.get_best()
    edges = ctx.get_valid_edges()
    ctx.compute_tracks()
    return allt, best, edges, ctx.get_tracks(), ctx.timers(), ctx.stats()


def _same(a, b):
    (ta, ba, ea, ka, _, _), (tb, bb, eb, kb, _, _) = a, b
    for k in ("off", "src", "line", "score"):
        assert np.array_equal(ta[k], tb[k]), f"all_tris[{k}] changed"
    for k in ("has_best", "src", "line", "score"):
        assert np.array_equal(ba[k], bb[k]), f"best[{k}] changed"
    assert np.array_equal(ea[0], eb[0]) and np.array_equal(ea[1], eb[1]), "valid edges changed"
    for k in ("off", "image_ids", "line_ids", "node_ids", "scores", "line"):
        assert np.array_equal(ka[k], kb[k]), f"tracks[{k}] changed"


@pytest.fixture
def clean_env():
    keys = ("LT_TEST_NO_FAST_GATES", "LT_TEST_NO_SCORE_GUARDS", "LT_TEST_PLACE_COPY", "LT_TEST_NO_TILE_CLASSES",
            "LT_TEST_EX_TWO_PASS", "LT_TEST_EX_PASS1_BLOCK", "LT_TEST_EX_PASS2_BLOCK", "LT_TEST_EX_CAP_FRAC",
            "LT_TEST_SCORE_UNSORTED", "LT_FINE_TIMERS", "LT_TIMER_SAMPLE", "LT_TEST_SCORE_F64", "LT_SCORE_FUSED", "LT_SCORE_SPLIT", "LT_TEST_DENSE_TABLES", "LT_TEST_DENSE_FEW_ROWS",
            "LT_TEST_SPLIT_SLOT", "LT_TEST_SPLIT_CHUNKS", "LT_TEST_PAIR_SCORE_TERMS", "LT_TEST_NO_PAIR_CLASSES", "LT_TEST_GATES_IMAGE_MAJOR",
            "LT_SCORE_TWO_KERNELS", "LT_TEST_Q_LOSE_TILE", "LT_TEST_TRI_STATIC")
    saved = {k: os.environ.pop(k, None) for k in keys}
    yield
    for k in keys:
        os.environ.pop(k, None)
        if saved[k] is not None:
            os.environ[k] = saved[k]


def test_fast_paths_are_conservative(gpu_lib, clean_env):
    sc = syn.make_scene(n_views=24, n_segs=160, n_neighbors=8, seed=21)
    cfg = syn.default_triangulation_cfg(debug_mode=True)
    base = _results(run_product(sc, cfg))
    assert base[5]["candidates"] > 2000 and base[5]["valid_edges"] > 100

    os.environ["LT_TEST_NO_FAST_GATES"] = "1"
    exact_gates = _results(run_product(sc, cfg))
    del os.environ["LT_TEST_NO_FAST_GATES"]
    _same(base, exact_gates)
    # every row went to the exact gates, and far more rows than the cheap gates let through
    assert exact_gates[4]["survivors"] == exact_gates[5]["connections"]
    assert base[4]["survivors"] < 0.5 * exact_gates[4]["survivors"]

    os.environ["LT_TEST_NO_SCORE_GUARDS"] = "1"
    no_guards = _results(run_product(sc, cfg))
    del os.environ["LT_TEST_NO_SCORE_GUARDS"]
    _same(base, no_guards)
    assert no_guards[4]["pairs_eval"] > 2 * base[4]["pairs_eval"]


@pytest.mark.parametrize("over", [
    dict(min_length_2d=40.0),                       # the length gate bites (squared-length guard band)
    dict(line_tri_angle_threshold=5.0),             # wider degeneracy gate
    dict(IoU_threshold=0.35),                       # stricter weak-epipolar gate
    dict(sensitivity_threshold=40.0),               # the post-triangulation gate bites
    dict(min_length_2d=0.0, IoU_threshold=0.0),     # degenerate thresholds
])
def test_gate_variants_match_oracle(gpu_lib, oracle, clean_env, over):
    sc = syn.make_scene(n_views=12, n_segs=90, n_neighbors=5, seed=33)
    cfg = syn.default_triangulation_cfg(debug_mode=True)
    cfg.update(over)
    T = run_product(sc, cfg)
    O = run_oracle(oracle, sc, cfg)
    compare_candidates(T.context().get_all_tris(), O.get_all_tris())
    compare_best(T.context().get_best(), O.get_best())
    compare_valid_edges(T.context().get_valid_edges(), O.get_valid_edges())
    T.context().compute_tracks()
    compare_tracks(T.context().get_tracks(), O.ComputeLineTracks())
    # and the exact-gates run agrees with the fast one
    os.environ["LT_TEST_NO_FAST_GATES"] = "1"
    T2 = run_product(sc, cfg)
    a, b = T.context().get_all_tris(), T2.context().get_all_tris()
    del os.environ["LT_TEST_NO_FAST_GATES"]
    for k in ("off", "src", "line", "score"):
        assert np.array_equal(a[k], b[k])


def test_release_cached_memory(gpu_lib):
    sc = syn.make_scene(n_views=6, n_segs=40, n_neighbors=3, seed=5)
    cfg = syn.default_triangulation_cfg()
    T = run_product(sc, cfg)
    n1 = len(T.ComputeLineTracks())
    del T
    gpu_lib.lt_release_cached_memory()      # everything cached goes back to the driver ...
    T = run_product(sc, cfg)                # ... and a new context still works (fresh allocations)
    assert len(T.ComputeLineTracks()) == n1
    gpu_lib.lt_release_cached_memory()      # live context unaffected
    assert len(T.context().get_tracks()["off"]) == n1 + 1


def test_warmup_and_reserved_staging(gpu_lib):
    """limap_amd.warmup() (a synthetic scene of the expected shape through the whole call sequence in both modes + the
    post-triangulation chain) leaves results untouched, and lt_reserve_host puts distinct page-locked blocks into the
    host cache that the next context picks up."""
    import limap_amd
    sc = syn.make_scene(n_views=6, n_segs=40, n_neighbors=3, seed=5)
    cfg = syn.default_triangulation_cfg()
    T = run_product(sc, cfg)
    n1 = len(T.ComputeLineTracks())
    del T
    assert limap_amd.warmup(n_views=8, n_segs=50, n_neighbors=4) > 0.0
    assert gpu_lib.lt_reserve_host(1 << 20, 3) == 0 and gpu_lib.lt_reserve_host(0, 2) == 0
    T = run_product(sc, cfg)
    assert len(T.ComputeLineTracks()) == n1
    del T
    gpu_lib.lt_release_cached_memory()


@pytest.mark.parametrize("own_big,nb_big", [(True, True), (False, True), (True, False)])
def test_operand_table_variants(gpu_lib, oracle, clean_env, own_big, nb_big):
    """k_gates keeps the operand tables in LDS only for images with <= 1024 segments; the other three
    instantiations (own and/or neighbour operands gathered from HBM/L2) must give the same lists."""
    from limap_amd import triangulation as tri
    sc = syn.make_scene(n_views=6, n_segs=1100, n_neighbors=3, seed=51, topk=3)
    big = {0, 1, 2}                      # images that keep all 1100 segments; the others keep 200
    keep = [1100 if i in big else 200 for i in range(6)]
    segs = [sc.segs_of(i)[:keep[i]] for i in range(6)]
    off = np.zeros(7, np.int64); off[1:] = np.cumsum(keep)
    # jobs: images whose size class is `own_big`, matched against neighbours of class `nb_big`
    jobs = [i for i in range(6) if (i in big) == own_big]
    cfg = syn.default_triangulation_cfg(debug_mode=True)
    T = tri.GlobalLineTriangulator(cfg)
    O = oracle.OracleTriangulator(cfg, faithful=False)
    T.SetRanges(sc.ranges); O.SetRanges(sc.ranges)
    T.InitArrays(sc.img_ids, sc.kvec, sc.qvec, sc.tvec, segs)
    O.Init(sc.img_ids, sc.kvec, sc.qvec, sc.tvec, off, np.concatenate(segs, 0))
    rng = np.random.default_rng(7)
    n_rows = 0
    for i in jobs:
        m = {}
        for nb in range(6):
            if nb == i or (nb in big) != nb_big:
                continue
            k = 3 * keep[i]
            rows = np.stack([np.repeat(np.arange(keep[i]), 3), rng.integers(0, keep[nb], k)], 1).astype(np.int32)
            m[int(sc.img_ids[nb])] = rows
            n_rows += k
        T.TriangulateImage(int(sc.img_ids[i]), m)
        O.TriangulateImage(int(sc.img_ids[i]), m)
    assert n_rows > 0
    compare_candidates(T.context().get_all_tris(), O.get_all_tris())
    compare_best(T.context().get_best(), O.get_best())
    compare_valid_edges(T.context().get_valid_edges(), O.get_valid_edges())


def test_no_lds_tables_switch(gpu_lib, clean_env):
    """LT_GEN_NO_LDS_TABLE (developer switch): all operands gathered from HBM/L2, identical results."""
    sc = syn.make_scene(n_views=10, n_segs=120, n_neighbors=4, seed=52)
    cfg = syn.default_triangulation_cfg(debug_mode=True)
    base = _results(run_product(sc, cfg))
    os.environ["LT_GEN_NO_LDS_TABLE"] = "1"
    try:
        other = _results(run_product(sc, cfg))
    finally:
        del os.environ["LT_GEN_NO_LDS_TABLE"]
    _same(base, other)


@pytest.mark.parametrize("exhaustive", [False, True])
def test_single_precision_sweep_is_conservative(gpu_lib, clean_env, exhaustive):
    """k_score3's early exit runs in single precision with its rounding added to the guards; the
    double-precision sweep (LT_TEST_SCORE_F64) and the sweep that skips nothing (LT_TEST_NO_SCORE_GUARDS) must
    give the same bits -- also for a scene far from the origin (coordinates ~1e5: the float form works on
    window-local coordinates) -- and may only evaluate fewer or equally many pairs densely."""
    for shift in (0.0, 1.0e5):
        sc = syn.make_scene(n_views=10, n_segs=120, n_neighbors=4, seed=54)
        if shift:
            sc = syn.translate_scene(sc, np.array([shift, -0.5 * shift, 0.25 * shift]))
        cfg = syn.default_triangulation_cfg(debug_mode=True)
        base = _results(run_product(sc, cfg, exhaustive=exhaustive))
        outs = {}
        for var in ("LT_TEST_SCORE_F64", "LT_TEST_NO_SCORE_GUARDS"):
            os.environ[var] = "1"
            try:
                outs[var] = _results(run_product(sc, cfg, exhaustive=exhaustive))
            finally:
                del os.environ[var]
            _same(base, outs[var])
        assert base[5]["candidates"] > 0
        # the float guards pass a superset of what the double guards pass, both a subset of everything
        assert outs["LT_TEST_SCORE_F64"][4]["pairs_eval"] <= base[4]["pairs_eval"] <= outs["LT_TEST_NO_SCORE_GUARDS"][4]["pairs_eval"]
        assert base[4]["pairs_eval"] <= 1.01 * outs["LT_TEST_SCORE_F64"][4]["pairs_eval"] + 64


def test_candidate_count_stays_on_device(gpu_lib, clean_env):
    """Default (sorted rows, bounded size): the compact arrays are sized by the one-candidate-per-staging-slot
    bound, the kernels read the exact count on the device and the run has no host round trip.
    LT_TEST_SYNC_COUNT: exact count through a stream sync.  Identical results, with the algebraic proposal
    alone and with several candidates per row (VP proposals)."""
    sc = syn.make_scene(n_views=10, n_segs=120, n_neighbors=4, seed=53)
    for extra in (False, True):
        cfg = syn.default_triangulation_cfg(debug_mode=True)
        vps = None
        if extra:
            cfg.update(use_vp=True)
            vps = syn.make_vp_results(sc, seed=3)
        base = _results(run_product(sc, cfg, vps=vps))
        os.environ["LT_TEST_SYNC_COUNT"] = "1"
        try:
            other = _results(run_product(sc, cfg, vps=vps))
        finally:
            del os.environ["LT_TEST_SYNC_COUNT"]
        _same(base, other)
        assert base[5]["candidates"] == other[5]["candidates"] > 0


def test_async_runs_pipeline(gpu_lib, clean_env):
    """lt_run_device_async: runs enqueued back to back (the previous one is completed after the next one is
    enqueued, two alternating sets of events / result slots) give the results of a synchronous run; timer sums
    count every run; an error of a run in flight surfaces at the next call that completes it."""
    sc = syn.make_scene(n_views=10, n_segs=120, n_neighbors=4, seed=55)
    cfg = syn.default_triangulation_cfg(debug_mode=True)
    base = _results(run_product(sc, cfg))
    T = run_product(sc, cfg)
    ctx = T.context()
    ctx.upload()
    ctx.timer_sums(reset=True)
    for _ in range(5):
        ctx.run_device(wait=False)
    sums, n = ctx.timer_sums()      # completes the run in flight
    assert n == 5 and sums["run"] > 0
    _same(base, _results(T))
    ctx.run_device(wait=False)
    ctx.sync()
    assert ctx.timer_sums()[1] == 6
    # the stage events of pipelined runs are sampled (LT_TIMER_SAMPLE, include/limap_amd.h: lt_get_timers): the sums are
    # scaled to the number of runs, the whole-run timer is measured for every run, the results do not depend on it
    per_run = {}
    for sample in ("1", "4"):
        os.environ["LT_TIMER_SAMPLE"] = sample
        ctx.sync()
        ctx.timer_sums(reset=True)
        for _ in range(12):
            ctx.run_device(wait=False)
        sums, n = ctx.timer_sums(reset=True)
        assert n == 12 and sums["run"] > 0 and sums["score"] > 0 and sums["gen"] > 0
        per_run[sample] = {k: sums[k] / n for k in ("run", "score", "gen")}
        _same(base, _results(T))
    os.environ.pop("LT_TIMER_SAMPLE", None)
    for k in ("score", "gen"):
        assert 0.5 < per_run["4"][k] / per_run["1"][k] < 2.0, (k, per_run)
    # a failing run (a shared point3D id without an SfM point is detected on the device)
    from limap_amd import triangulation as tri
    bpts, sfm = syn.make_bipartites(sc, seed=3)
    cfg2 = dict(cfg, disable_one_point_triangulation=True)
    T2 = tri.GlobalLineTriangulator(cfg2)
    T2.InitArrays(sc.img_ids, sc.kvec, sc.qvec, sc.tvec, [sc.segs_of(i) for i in range(sc.n_images)])
    T2.SetBipartites2d(bpts)
    T2.SetSfMPoints({k: v for k, v in list(sfm.items())[:3]})
    for i in sc.img_ids:
        T2.TriangulateImage(int(i), sc.matches_of(int(i)))
    c2 = T2.context()
    c2.upload()
    c2.run_device(wait=False)       # enqueued; nothing to complete yet
    with pytest.raises(RuntimeError, match="point3D_id"):
        c2.sync()


def test_full_size_invariants(gpu_lib, clean_env):
    """BASELINE's full size (100 views x 500 segments, 10^7 connections): the oracle needs ~8 s per run
    here (bench.py times it and checks its counts), so this test uses size-independent properties --
    the exact-gates run, the no-guards run and a second default run must all reproduce the default run
    bit for bit -- plus the counts the oracle gives for this seed (bench.py `cpu_parity`)."""
    sc = syn.make_scene(n_views=100, n_segs=500, n_neighbors=20, seed=0)
    cfg = syn.default_triangulation_cfg(debug_mode=True)
    matches = {int(i): sc.matches_of(int(i)) for i in sc.img_ids}

    def run():
        from limap_amd import triangulation as tri
        T = tri.GlobalLineTriangulator(cfg)
        T.SetRanges(sc.ranges)
        T.InitArrays(sc.img_ids, sc.kvec, sc.qvec, sc.tvec, [sc.segs_of(i) for i in range(sc.n_images)])
        for i in sc.img_ids:
            T.TriangulateImage(int(i), matches[int(i)])
        return _results(T)

    base = run()
    assert base[5]["connections"] == 10_000_000
    assert base[5]["candidates"] == 579_235 and base[5]["tracks"] == 1_367 and base[5]["valid_edges"] == 96_488
    _same(base, run())                                   # idempotent
    os.environ["LT_TEST_NO_FAST_GATES"] = "1"
    exact = run()
    del os.environ["LT_TEST_NO_FAST_GATES"]
    _same(base, exact)
    assert exact[4]["survivors"] == 10_000_000
    os.environ["LT_TEST_NO_SCORE_GUARDS"] = "1"
    dense = run()
    del os.environ["LT_TEST_NO_SCORE_GUARDS"]
    _same(base, dense)
    assert dense[4]["pairs_eval"] > 10 * base[4]["pairs_eval"]


def test_place_by_permutation_equals_place_by_copy(gpu_lib, clean_env):
    """Matched fast path: k_place writes only perm[final position] = staging slot and every consumer (scoring,
    selection, edge lists, tail, debug read-outs) reads the records through it; LT_TEST_PLACE_COPY=1 moves the
    records into compact arrays instead.  Both forms give the same bits everywhere."""
    sc = syn.make_scene(n_views=24, n_segs=200, n_neighbors=8, seed=12)
    cfg = syn.default_triangulation_cfg(debug_mode=True)

    def run():
        from limap_amd import triangulation as tri
        T = tri.GlobalLineTriangulator(cfg)
        T.SetRanges(sc.ranges)
        T.InitArrays(sc.img_ids, sc.kvec, sc.qvec, sc.tvec, [sc.segs_of(i) for i in range(sc.n_images)])
        for i in sc.img_ids:
            T.TriangulateImage(int(i), sc.matches_of(int(i)))
        return _results(T)

    base = run()
    os.environ["LT_TEST_PLACE_COPY"] = "1"
    copy = run()
    del os.environ["LT_TEST_PLACE_COPY"]
    _same(base, copy)
    assert base[5]["candidates"] > 10_000 and base[5]["tracks"] > 50


def test_tile_cost_classes_do_not_change_results(gpu_lib, clean_env):
    """k_score3 draws its tiles by cost class (longest first); LT_TEST_NO_TILE_CLASSES draws them in natural order.
    Scheduling only: the same bits."""
    sc = syn.make_scene(n_views=24, n_segs=200, n_neighbors=8, seed=13)
    cfg = syn.default_triangulation_cfg(debug_mode=True)
    base = _results(run_product(sc, cfg))
    os.environ["LT_TEST_NO_TILE_CLASSES"] = "1"
    plain = _results(run_product(sc, cfg))
    del os.environ["LT_TEST_NO_TILE_CLASSES"]
    _same(base, plain)
    ex = _results(run_product(small_scene(seed=3, n_views=8, n_segs=60, n_neighbors=4), cfg, exhaustive=True))
    assert ex[5]["candidates"] > 0


def test_exhaustive_one_pass_equals_two_pass(gpu_lib, clean_env):
    """Plain exhaustive mode: the one-pass form (pass 1 writes the survivors to staging slots, a permutation orders
    them, the depth-sorted sweep runs over the staged records) against the two-pass forms (new kernels, and the
    wave-per-(node, neighbour) kernels the VP variant uses) -- identical bits everywhere.  A staging capacity that
    does not hold makes the run repeat itself in the two-pass form; the plain (unsorted) sweep over staged records
    is the matched path's permutation mode."""
    sc = small_scene(seed=21, n_views=10, n_segs=150, n_neighbors=5)
    cfg = syn.default_triangulation_cfg(debug_mode=True)
    base = _results(run_product(sc, cfg, exhaustive=True))
    assert base[0]["off"][-1] > 1000
    variants = ({"LT_TEST_EX_TWO_PASS": "1"},
                {"LT_TEST_EX_TWO_PASS": "1", "LT_TEST_EX_PASS2_BLOCK": "1"},
                {"LT_TEST_EX_PASS1_BLOCK": "1", "LT_TEST_EX_PASS2_BLOCK": "1"},
                {"LT_TEST_EX_CAP_FRAC": "0.0005"},
                {"LT_TEST_SCORE_UNSORTED": "1"})
    for env in variants:
        os.environ.update(env)
        try:
            _same(base, _results(run_product(sc, cfg, exhaustive=True)))
        finally:
            for k in env:
                os.environ.pop(k, None)
    # a context that runs the job repeatedly adapts the capacity to the measured yield
    T = run_product(sc, cfg, exhaustive=True)
    ctx = T.context()
    ctx.upload()
    for _ in range(3):
        ctx.run_device()
    _same(base, _results(T))


def test_exhaustive_many_chunks(gpu_lib, clean_env):
    """More than 64 chunks of 64 neighbour lines per image (4200 segments): the chunk loops of the one-pass and
    two-pass kernels against the wave-per-(node, neighbour) kernels (which the parity tests pin to the oracle at
    sizes the oracle can do)."""
    sc = small_scene(seed=32, n_views=6, n_segs=4200, n_neighbors=3)
    cfg = syn.default_triangulation_cfg()

    def res():
        T = run_product(sc, cfg, exhaustive=True)
        ctx = T.context()
        best, edges = ctx.get_best(), ctx.get_valid_edges()
        ctx.compute_tracks()
        return best, edges, ctx.get_tracks(), ctx.stats()

    base = res()
    assert base[3]["candidates"] > 10000
    for env in ({"LT_TEST_EX_TWO_PASS": "1"}, {"LT_TEST_EX_PASS1_BLOCK": "1", "LT_TEST_EX_PASS2_BLOCK": "1"}):
        os.environ.update(env)
        try:
            other = res()
        finally:
            for k in env:
                os.environ.pop(k, None)
        assert other[3]["candidates"] == base[3]["candidates"]
        for k in ("has_best", "src", "line", "score"):
            assert np.array_equal(base[0][k], other[0][k]), k
        assert np.array_equal(base[1][0], other[1][0]) and np.array_equal(base[1][1], other[1][1])
        for k in ("off", "image_ids", "line_ids", "node_ids", "scores", "line"):
            assert np.array_equal(base[2][k], other[2][k]), k


def test_per_kernel_event_levels(gpu_lib, clean_env):
    """LT_FINE_TIMERS is read per run: by default only k_score3 carries its own HIP events (what bench.py prices the
    dominant kernel with), 2 adds the generation kernels' (timers [13], [14]), 0 turns all of them off -- the stage
    timers [3]-[6] do not depend on it."""
    sc = small_scene(seed=41, n_views=12, n_segs=150, n_neighbors=6)
    T = run_product(sc, syn.default_triangulation_cfg())
    ctx = T.context()
    ctx.upload()
    seen = {}
    for level in (None, "2", "0"):
        if level is None:
            os.environ.pop("LT_FINE_TIMERS", None)
        else:
            os.environ["LT_FINE_TIMERS"] = level
        ctx.run_device()
        seen[level] = ctx.timers()
    os.environ.pop("LT_FINE_TIMERS", None)
    assert seen[None]["k_score3"] > 0 and seen[None]["k_gates"] == 0 and seen[None]["k_tri_rows"] == 0
    assert seen["2"]["k_score3"] > 0 and seen["2"]["k_gates"] > 0 and seen["2"]["k_tri_rows"] > 0
    assert seen["0"]["k_score3"] == 0 and seen["0"]["k_gates"] == 0
    for t in seen.values():
        assert t["gen"] > 0 and t["score"] > 0 and t["run"] >= t["gen"] + t["score"]


@pytest.mark.parametrize("topk,n_nb", [(10, 6), (60, 13)])
def test_scoring_sweep_forms_agree(gpu_lib, clean_env, topk, n_nb):
    """k_score3's single-precision conservative sweep against the double-precision sweep (LT_TEST_SCORE_F64), tiles in
    natural order instead of cost classes (window bounds from the lanes' records), and no guards at all (every pair of a
    node evaluated exactly): same bits everywhere; only WHICH pairs reach pair_score differs.  The second scene has nodes
    of up to 106 candidates: windows beyond one LDS chunk (79 of its 624 tiles have 129-197 window entries)."""
    sc = syn.make_scene(n_views=14, n_segs=140, n_neighbors=n_nb, seed=77, topk=topk)
    cfg = syn.default_triangulation_cfg(debug_mode=True)
    base = _results(run_product(sc, cfg, topk=topk))
    assert base[5]["candidates"] > 1000
    if topk > 10:
        assert np.diff(base[0]["off"]).max() > 100, "the scene is meant to have windows beyond one chunk"
    os.environ["LT_TEST_SCORE_F64"] = "1"
    f64 = _results(run_product(sc, cfg, topk=topk))
    _same(base, f64)
    del os.environ["LT_TEST_SCORE_F64"]
    os.environ["LT_TEST_NO_TILE_CLASSES"] = "1"
    nat = _results(run_product(sc, cfg, topk=topk))
    _same(base, nat)
    del os.environ["LT_TEST_NO_TILE_CLASSES"]
    os.environ["LT_TEST_NO_SCORE_GUARDS"] = "1"
    allp = _results(run_product(sc, cfg, topk=topk))
    _same(base, allp)
    assert allp[4]["pairs_eval"] > base[4]["pairs_eval"]


@pytest.mark.parametrize("topk,n_nb", [(10, 6), (60, 13)])
def test_split_and_fused_scoring_agree(gpu_lib, clean_env, topk, n_nb):
    """The scoring stage runs as ONE kernel by default (k_score_q, round 6: workgroups sweep their share of the tiles and
    publish each finished tile in their XCD's FIFO, then turn to evaluating units of finished tiles), keeps the two-kernel
    form of rounds 4-6 (sweep kernel -> k_dense8; LT_SCORE_TWO_KERNELS, and the fallback behind device flag 8) and the fused
    k_score3 as the last fallback.  Same bits from: the two-kernel form, also with overflow chains everywhere; a run in which
    a tile is never published (LT_TEST_Q_LOSE_TILE: a unit's wait runs into its bound, flag 8, the run is repeated in the
    two-kernel form and the context stays there); the fused kernel (LT_SCORE_FUSED); slots of four entries
    (every tile with more pairs continues in a chain of overflow chunks); an overflow store of one chunk (the store
    fills, device flag 7, the run is repeated fused and the context stays fused); the split form over the natural tile
    order; the split form of the exhaustive mode (not its default)."""
    sc = syn.make_scene(n_views=14, n_segs=140, n_neighbors=n_nb, seed=77, topk=topk)
    cfg = syn.default_triangulation_cfg(debug_mode=True)
    base = _results(run_product(sc, cfg, topk=topk))
    assert base[5]["candidates"] > 1000
    if topk > 10:
        assert np.diff(base[0]["off"]).max() > 100, "the scene is meant to have windows beyond one chunk"
    os.environ["LT_TEST_SCORE_F64"] = "1"
    f64 = _results(run_product(sc, cfg, topk=topk))
    _same(base, f64)
    del os.environ["LT_TEST_SCORE_F64"]
    os.environ["LT_TEST_NO_TILE_CLASSES"] = "1"
    nat = _results(run_product(sc, cfg, topk=topk))
    _same(base, nat)
    del os.environ["LT_TEST_NO_TILE_CLASSES"]
    os.environ["LT_TEST_NO_SCORE_GUARDS"] = "1"
    allp = _results(run_product(sc, cfg, topk=topk))
    _same(base, allp)
    assert allp[4]["pairs_eval"] > base[4]["pairs_eval"]


@pytest.mark.parametrize("topk,n_nb", [(10, 6), (60, 13)])
def test_split_and_fused_scoring_agree(gpu_lib, clean_env, topk, n_nb):
    """The scoring stage runs as two kernels by default (sweep -> pair slots per tile -> k_dense8 over units of tiles) and
    keeps the fused k_score3 as the fallback.  Same bits from: the ONE-kernel form of round 6 (k_score_q, LT_SCORE_ONE_KERNEL:
    workgroups alternate between sweeping tiles and evaluating units of finished tiles from their XCD's FIFO; measured, not
    the default), also with overflow chains everywhere; the fused kernel (LT_SCORE_FUSED); slots of four entries
    (every tile with more pairs continues in a chain of overflow chunks); an overflow store of one chunk (the store
    fills, device flag 7, the run is repeated fused and the context stays fused); the split form over the natural tile
    order; the split form of the exhaustive mode (not its default)."""
    sc = syn.make_scene(n_views=14, n_segs=140, n_neighbors=n_nb, seed=77, topk=topk)
    cfg = syn.default_triangulation_cfg(debug_mode=True)
    base = _results(run_product(sc, cfg, topk=topk))
    assert base[5]["candidates"] > 1000 and base[4]["pairs_eval"] > 1000 and base[4]["score_two_kernels"] == 0
    os.environ["LT_SCORE_TWO_KERNELS"] = "1"
    two = _results(run_product(sc, cfg, topk=topk))
    _same(base, two)
    assert two[4]["pairs_eval"] == base[4]["pairs_eval"]
    os.environ["LT_TEST_SPLIT_SLOT"] = "4"
    _same(base, _results(run_product(sc, cfg, topk=topk)))
    del os.environ["LT_SCORE_TWO_KERNELS"], os.environ["LT_TEST_SPLIT_SLOT"]
    os.environ["LT_TEST_Q_LOSE_TILE"] = "1"
    lost = _results(run_product(sc, cfg, topk=topk))
    _same(base, lost)
    assert lost[4]["score_two_kernels"] == 1
    del os.environ["LT_TEST_Q_LOSE_TILE"]
    os.environ["LT_SCORE_FUSED"] = "1"
    fused = _results(run_product(sc, cfg, topk=topk))
    _same(base, fused)
    assert fused[4]["pairs_eval"] == base[4]["pairs_eval"]
    del os.environ["LT_SCORE_FUSED"]
    os.environ["LT_TEST_SPLIT_SLOT"] = "4"
    chains = _results(run_product(sc, cfg, topk=topk))
    _same(base, chains)
    assert chains[4]["pairs_eval"] == base[4]["pairs_eval"]
    os.environ["LT_TEST_SPLIT_CHUNKS"] = "1"  # slots of four entries and one overflow chunk: the store fills
    full = _results(run_product(sc, cfg, topk=topk))
    _same(base, full)
    assert full[4]["score_fused"] == 1 and chains[4]["score_fused"] == 0 and base[4]["score_fused"] == 0
    del os.environ["LT_TEST_SPLIT_SLOT"], os.environ["LT_TEST_SPLIT_CHUNKS"]
    # k_dense8's units in the order of k_cand_meta's cost classes instead of the sweep's pair-count lists (round 6)
    os.environ["LT_TEST_NO_PAIR_CLASSES"] = "1"
    _same(base, _results(run_product(sc, cfg, topk=topk)))
    del os.environ["LT_TEST_NO_PAIR_CLASSES"]
    os.environ["LT_TEST_SPLIT_SLOT"] = "4"   # ... and the pair-count lists with overflow chains everywhere
    os.environ["LT_TEST_NO_PAIR_CLASSES"] = "1"
    _same(base, _results(run_product(sc, cfg, topk=topk)))
    del os.environ["LT_TEST_SPLIT_SLOT"], os.environ["LT_TEST_NO_PAIR_CLASSES"]
    os.environ["LT_TEST_NO_TILE_CLASSES"] = "1"
    os.environ["LT_SCORE_SPLIT"] = "1"
    nat = _results(run_product(sc, cfg, topk=topk))
    _same(base, nat)
    del os.environ["LT_TEST_NO_TILE_CLASSES"]
    del os.environ["LT_SCORE_SPLIT"]
    # exhaustive mode (depth-sorted tiles): split by default since round 6, fused when asked for
    ex_default = _results(run_product(sc, cfg, exhaustive=True))
    os.environ["LT_SCORE_FUSED"] = "1"
    ex_fused = _results(run_product(sc, cfg, exhaustive=True))
    _same(ex_default, ex_fused)
    assert ex_default[4]["score_fused"] == 0 and ex_fused[4]["pairs_eval"] == ex_default[4]["pairs_eval"]
    del os.environ["LT_SCORE_FUSED"]
    os.environ["LT_TEST_SPLIT_SLOT"] = "4"  # ... and its overflow chains
    _same(ex_default, _results(run_product(sc, cfg, exhaustive=True)))
    del os.environ["LT_TEST_SPLIT_SLOT"]
    # the dense kernel of the exhaustive mode: units of eight tiles over a row-compacted table (k_dense_rows) by default;
    # with a table of 64 rows (units worked off in groups of tiles), also over overflow chains; k_dense8's per-tile tables
    for env in ({"LT_TEST_DENSE_FEW_ROWS": "1"}, {"LT_TEST_DENSE_FEW_ROWS": "1", "LT_TEST_SPLIT_SLOT": "4"}, {"LT_TEST_DENSE_TABLES": "1"}):
        os.environ.update(env)
        r = _results(run_product(sc, cfg, exhaustive=True))
        _same(ex_default, r)
        assert r[4]["pairs_eval"] == ex_default[4]["pairs_eval"]
        for k in env:
            del os.environ[k]


@pytest.mark.parametrize("shape", [(24, 160, 8, 10), (40, 30, 12, 4), (9, 700, 5, 3)])
def test_stage_a_block_order_does_not_matter(gpu_lib, clean_env, shape):
    """k_gates_ln takes the blocks in (neighbour, image) order, one contiguous eighth of it per XCD, and keeps a neighbour's
    gate table in LDS across blocks that share it (round 6); LT_TEST_GATES_IMAGE_MAJOR=1 runs them image-major with a table
    load per block as in round 5.  Same bits: scenes with one block per workgroup, with many (more blocks than workgroup
    slots), and with tables of more than 512 segments (workgroups of eight waves)."""
    n_views, n_segs, nn, topk = shape
    sc = syn.make_scene(n_views=n_views, n_segs=n_segs, n_neighbors=nn, seed=91, topk=topk)
    cfg = syn.default_triangulation_cfg(debug_mode=True)
    new = _results(run_product(sc, cfg, topk=topk))
    assert new[5]["candidates"] > 300 and new[4]["line_slots"] == 1
    os.environ["LT_TEST_GATES_IMAGE_MAJOR"] = "1"
    old = _results(run_product(sc, cfg, topk=topk))
    _same(old, new)


def test_stage_b_unit_claims_equal_the_static_deal(gpu_lib, clean_env):
    """k_tri_rounds claims its units (the rounds x, x + 4, ... of a block) from per-XCD counters once a workgroup has four
    blocks or more (round 6: BASELINE config 3 0.855 -> 0.778 ms); LT_TEST_TRI_STATIC=1 deals the blocks g, g + G, ... as
    before.  Same bits on a scene with 4 400 blocks (220 views x 20 neighbours)."""
    sc = syn.make_scene(n_views=220, n_segs=40, n_neighbors=20, seed=31, topk=4)
    cfg = syn.default_triangulation_cfg(debug_mode=True)
    new = _results(run_product(sc, cfg, topk=4))
    assert new[5]["candidates"] > 2000 and new[4]["line_slots"] == 1
    os.environ["LT_TEST_TRI_STATIC"] = "1"
    old = _results(run_product(sc, cfg, topk=4))
    _same(old, new)


@pytest.mark.parametrize("variant", ["default", "no_smartangle", "no_overlap", "angle_only", "other_thresholds"])
def test_pair_score_fused_equals_term_by_term(gpu_lib, clean_env, variant):
    """pair_score evaluates the shared sub-expressions of the 2D linker once and ONE exponential (of the largest q) instead
    of the reference's five gated ones (lt_devfn.h); LT_TEST_PAIR_SCORE_TERMS=1 runs the reference's text term by term.
    Same bits -- every candidate's support score, hence every discrete result -- in both scoring forms and both modes."""
    sc = syn.make_scene(n_views=16, n_segs=150, n_neighbors=7, seed=123)
    cfg = syn.default_triangulation_cfg(debug_mode=True)
    l2 = dict(cfg["linker2d_config"]) if "linker2d_config" in cfg else None
    key2, key3 = ("linker2d_config", "linker3d_config")
    if variant != "default":
        assert key2 in cfg and key3 in cfg, sorted(cfg)
    if variant == "no_smartangle":
        cfg[key2] = dict(l2, use_smartangle=False)
    elif variant == "no_overlap":
        cfg[key2] = dict(l2, use_overlap=False)
    elif variant == "angle_only":
        cfg[key2] = dict(l2, use_overlap=False, use_smartangle=False, use_perp=False)
    elif variant == "other_thresholds":
        cfg[key2] = dict(l2, score_th=0.3, th_angle=9.0, th_perp=4.0, th_overlap=0.02, th_smartoverlap=0.2)
        cfg[key3] = dict(cfg[key3], score_th=0.7, th_angle=12.0, th_scaleinv=0.03)
    for exhaustive in (False, True):
        fused = _results(run_product(sc, cfg, exhaustive=exhaustive))
        assert fused[5]["candidates"] > 1000 and fused[4]["pairs_eval"] > 500
        os.environ["LT_TEST_PAIR_SCORE_TERMS"] = "1"
        terms = _results(run_product(sc, cfg, exhaustive=exhaustive))
        del os.environ["LT_TEST_PAIR_SCORE_TERMS"]
        _same(terms, fused)
    os.environ["LT_SCORE_FUSED"] = "1"
    one_kernel = _results(run_product(sc, cfg))
    os.environ["LT_TEST_PAIR_SCORE_TERMS"] = "1"
    _same(_results(run_product(sc, cfg)), one_kernel)


@pytest.mark.parametrize("n_nb", [100, 40])
def test_split_scoring_with_wide_neighbour_lists(gpu_lib, clean_env, n_nb):
    """k_dense8 holds one table of maxima per tile, 512 B per neighbour of the widest image: 100 neighbours leave room for
    ONE tile per unit (40: two).  Same bits as the fused kernel."""
    sc = syn.make_scene(n_views=110, n_segs=24, n_neighbors=n_nb, seed=5, topk=3)
    cfg = syn.default_triangulation_cfg(debug_mode=True)
    split = _results(run_product(sc, cfg, topk=3))
    assert split[5]["candidates"] > 500 and split[4]["score_fused"] == 0
    os.environ["LT_SCORE_FUSED"] = "1"
    fused = _results(run_product(sc, cfg, topk=3))
    _same(split, fused)
    assert split[4]["pairs_eval"] == fused[4]["pairs_eval"]


def test_second_batch_after_compute_tracks_device_and_host_tail(gpu_lib, oracle, clean_env):
    """ComputeLineTracks ends a batch.  A TriangulateImage call after it starts a NEW batch (the first one's results stay
    on the host) whether the tail ran on the device (default) or on the host (LT_TAIL_HOST); a repeated call for an image
    that is already triangulated is a no-op (already_scored_, global_line_triangulator.cc:73) and invalidates nothing.
    Tracks after the second ComputeLineTracks = the whole scene's, equal in both forms and equal to the oracle's."""
    from helpers import compare_tracks, run_oracle
    from limap_amd import triangulation as tri
    sc = syn.make_scene(n_views=12, n_segs=100, n_neighbors=5, seed=9)
    cfg = syn.default_triangulation_cfg()
    ids = [int(i) for i in sc.img_ids]

    def run():
        T = tri.GlobalLineTriangulator(cfg)
        T.SetRanges(sc.ranges)
        T.InitArrays(sc.img_ids, sc.kvec, sc.qvec, sc.tvec, [sc.segs_of(i) for i in range(sc.n_images)])
        for i in ids[:6]:
            T.TriangulateImage(i, sc.matches_of(i))
        first = T.ComputeLineTracks()
        st1 = T.stats()
        T.TriangulateImage(ids[0], sc.matches_of(ids[0]))      # already scored: nothing changes, nothing re-runs
        assert T.stats() == st1 and len(T.GetTracks()) == len(first)
        for i in ids[6:]:
            T.TriangulateImage(i, sc.matches_of(i))
        T.ComputeLineTracks()
        t = T.context().get_tracks()
        # the second run handled the second batch only
        assert T.stats()["connections"] < st1["connections"] * 3
        return len(first), t

    os.environ.pop("LT_TAIL_HOST", None)
    n1_dev, t_dev = run()
    os.environ["LT_TAIL_HOST"] = "1"
    try:
        n1_host, t_host = run()
    finally:
        del os.environ["LT_TAIL_HOST"]
    assert n1_dev == n1_host > 0
    for k in ("off", "image_ids", "line_ids", "node_ids", "scores", "line"):
        assert np.array_equal(t_dev[k], t_host[k]), k
    O = run_oracle(oracle, sc, cfg)
    compare_tracks(t_dev, O.ComputeLineTracks())


@pytest.mark.parametrize("between", ["tracks", "getter"])
def test_triangulate_all_second_batch_after_results_were_read(gpu_lib, oracle, clean_env, between):
    """TriangulateAll(first half) -> ComputeLineTracks or a getter -> TriangulateAll(second half).  Reading the results
    ends the first batch: the second call's begin_image() clears the staged rows, so its rows must be written at the
    start of the (emptied) staging block, not behind the first batch's old size (`base` was read before the
    loop that may clear the block, the device then received the FIRST batch's packed rows again).  Compared with the
    per-image loop in the same two batches and with the oracle on the whole scene."""
    from helpers import compare_best, compare_tracks, run_oracle
    from limap_amd import triangulation as tri
    sc = syn.make_scene(n_views=12, n_segs=100, n_neighbors=5, seed=21)
    cfg = syn.default_triangulation_cfg()
    ids = [int(i) for i in sc.img_ids]

    def run(batched):
        T = tri.GlobalLineTriangulator(cfg)
        T.SetRanges(sc.ranges)
        T.InitArrays(sc.img_ids, sc.kvec, sc.qvec, sc.tvec, [sc.segs_of(i) for i in range(sc.n_images)])
        for part in (ids[:6], ids[6:]):
            if batched:
                T.TriangulateAll({i: sc.matches_of(i) for i in part})
            else:
                for i in part:
                    T.TriangulateImage(i, sc.matches_of(i))
            if part is not ids[6:] and between == "tracks":
                T.ComputeLineTracks()
            elif between == "getter":
                T.context().get_best()
        T.ComputeLineTracks()
        return T.context().get_best(), T.context().get_tracks(), T.stats()

    b_all, t_all, st_all = run(True)
    b_one, t_one, st_one = run(False)
    assert st_all["candidates"] == st_one["candidates"] > 0
    for k in b_all:
        assert np.array_equal(b_all[k], b_one[k]), k
    for k in ("off", "image_ids", "line_ids", "node_ids", "scores", "line"):
        assert np.array_equal(t_all[k], t_one[k]), k
    O = run_oracle(oracle, sc, cfg)
    compare_best(b_all, O.get_best())
    compare_tracks(t_all, O.ComputeLineTracks())
