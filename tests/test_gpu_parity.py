"""-m gpu: HIP backend (through the C ABI) against the CPU oracle on identical seeded inputs.
Bars: indices bit-exact; coordinates that never pass through acos/exp bit-exact; scores 1e-12;
aggregated track endpoints 1e-5 relative (north_star)."""
import numpy as np
import pytest

from limap_amd import synthetic as syn

from helpers import (compare_best, compare_candidates, compare_tracks, compare_valid_edges, run_oracle,
                     run_product, small_scene)

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", [0, 1])
def test_matched_stage_by_stage(gpu_lib, oracle, seed):
    sc = small_scene(seed=seed)
    cfg = syn.default_triangulation_cfg(debug_mode=True)
    T = run_product(sc, cfg)
    O = run_oracle(oracle, sc, cfg)
    compare_candidates(T.context().get_all_tris(), O.get_all_tris())
    assert np.array_equal(T.context().get_num_tris(), O.get_num_tris())
    compare_best(T.context().get_best(), O.get_best())
    compare_valid_edges(T.context().get_valid_edges(), O.get_valid_edges())
    T.ComputeLineTracks()
    ot = O.ComputeLineTracks()
    compare_tracks(T.context().get_tracks(), ot)
    st, so = T.stats(), O.stats()
    for k in ("connections", "candidates", "pairs", "valid_edges", "graph_nodes", "graph_edges", "tracks"):
        assert st[k] == so[k], k


def test_exhaustive_stage_by_stage(gpu_lib, oracle):
    sc = small_scene(seed=2, n_views=10, n_segs=70, n_neighbors=5)
    cfg = syn.default_triangulation_cfg(debug_mode=True)
    T = run_product(sc, cfg, exhaustive=True)
    O = run_oracle(oracle, sc, cfg, exhaustive=True)
    compare_candidates(T.context().get_all_tris(), O.get_all_tris())
    compare_best(T.context().get_best(), O.get_best())
    compare_valid_edges(T.context().get_valid_edges(), O.get_valid_edges())
    T.ComputeLineTracks()
    compare_tracks(T.context().get_tracks(), O.ComputeLineTracks())


def test_smoke_entry(gpu_lib, oracle):
    import __graft_entry__ as g
    g.smoke()


def test_matched_unsorted_rows_generic_grouping(gpu_lib, oracle):
    """Rows of a block in arbitrary order (not grouped by line id): the backend falls back to the
    stable radix sort by node; the candidate order must still follow the match-row order."""
    sc = small_scene(seed=4, n_views=12, n_segs=90, n_neighbors=6)
    cfg = syn.default_triangulation_cfg(debug_mode=True)
    from limap_amd import triangulation as tri
    T = tri.GlobalLineTriangulator(cfg)
    O = oracle.OracleTriangulator(cfg, faithful=False)
    T.SetRanges(sc.ranges); O.SetRanges(sc.ranges)
    T.InitArrays(sc.img_ids, sc.kvec, sc.qvec, sc.tvec, [sc.segs_of(i) for i in range(sc.n_images)])
    O.Init(sc.img_ids, sc.kvec, sc.qvec, sc.tvec, sc.seg_off, sc.segs)
    rng = np.random.default_rng(0)
    for i in sc.img_ids:
        m = sc.matches_of(int(i))
        m = {k: v[rng.permutation(len(v))] for k, v in m.items()}
        # dict order scrambled too: the reference iterates a std::map (ascending neighbour id)
        keys = list(m.keys()); rng.shuffle(keys)
        m = {k: m[k] for k in keys}
        T.TriangulateImage(int(i), m)
        O.TriangulateImage(int(i), m)
    compare_candidates(T.context().get_all_tris(), O.get_all_tris())
    compare_best(T.context().get_best(), O.get_best())
    compare_valid_edges(T.context().get_valid_edges(), O.get_valid_edges())
    T.ComputeLineTracks()
    compare_tracks(T.context().get_tracks(), O.ComputeLineTracks())


def test_device_resident_scene_and_chunked_refresh(gpu_lib, oracle):
    """lt_init_device + lt_set_scene_chunks / lt_refresh_scene_chunks (the per-step path of the
    multi-GPU job, here with the scene cut into 3 artificial chunks) give the results of a plain
    host Init; kernels run on torch's current stream."""
    import torch
    from limap_amd import _capi
    sc = small_scene(seed=5, n_views=9, n_segs=70, n_neighbors=5)
    cfg = syn.default_triangulation_cfg()
    ref = run_product(sc, cfg).context().get_best()

    dev = torch.device("cuda", 0)
    ctx = _capi.Context(cfg_dict=cfg, device=0)
    ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    ctx.set_ranges(*sc.ranges)
    bounds = [0, 2, 6, 9]
    chunks = []
    for a, b in zip(bounds[:-1], bounds[1:]):
        s0, s1 = int(sc.seg_off[a]), int(sc.seg_off[b])
        buf = np.concatenate([sc.kvec[a:b].ravel(), sc.qvec[a:b].ravel(), sc.tvec[a:b].ravel(), sc.segs[s0:s1].ravel()])
        chunks.append((a, b, torch.from_numpy(buf).to(dev)))
    dk = torch.from_numpy(sc.kvec).to(dev); dq = torch.from_numpy(sc.qvec).to(dev)
    dt = torch.from_numpy(sc.tvec).to(dev); ds = torch.from_numpy(sc.segs).to(dev)
    torch.cuda.synchronize()
    ctx.init_device(sc.img_ids, dk.data_ptr(), dq.data_ptr(), dt.data_ptr(), sc.seg_off, ds.data_ptr())
    # now poison the invariants' sources and rebuild from the chunks
    dk.fill_(float("nan")); ds.fill_(float("nan"))
    ib = [a for a, _, _ in chunks]
    n = [b - a for a, b, _ in chunks]
    base = [t.data_ptr() for _, _, t in chunks]
    ctx.set_scene_chunks(ib, base, [p + 32 * k for p, k in zip(base, n)], [p + 64 * k for p, k in zip(base, n)],
                         [p + 88 * k for p, k in zip(base, n)])
    ctx.refresh_scene_chunks()
    for i in sc.img_ids:
        m = sc.matches_of(int(i))
        nb = list(m.keys())
        off = np.zeros(len(nb) + 1, np.int64); off[1:] = np.cumsum([len(m[k]) for k in nb])
        ctx.triangulate_image(int(i), nb, off, np.concatenate([m[k] for k in nb], 0))
    got = ctx.get_best()
    compare_best(got, ref)
    assert np.array_equal(got["line"], ref["line"])


def test_refresh_rebuilds_only_referenced_images(gpu_lib, oracle):
    """A rank of a multi-GPU job triangulates a shard: while its job is uploaded, lt_refresh_scene_chunks
    rebuilds the segment records of the images that job references only (here: scene data changes under a
    context that triangulates 4 of 12 images -- the per-step path of bench.py --gpus N)."""
    import torch
    from limap_amd import _capi
    from helpers import run_oracle
    sc = small_scene(seed=8, n_views=12, n_segs=60, n_neighbors=3)
    cfg = syn.default_triangulation_cfg(debug_mode=True)
    mine = [int(i) for i in sc.img_ids[4:8]]
    O = run_oracle(oracle, sc, cfg, images=mine)
    dev = torch.device("cuda", 0)
    ctx = _capi.Context(cfg_dict=cfg, device=0)
    ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    ctx.set_ranges(*sc.ranges)
    # Init from garbage-free but WRONG data (everything shifted), the true scene arrives through the chunks
    wrong = sc.segs + 3.0
    dk = torch.from_numpy(sc.kvec).to(dev); dq = torch.from_numpy(sc.qvec).to(dev)
    dt = torch.from_numpy(sc.tvec).to(dev); ds = torch.from_numpy(wrong).to(dev)
    ctx.init_device(sc.img_ids, dk.data_ptr(), dq.data_ptr(), dt.data_ptr(), sc.seg_off, ds.data_ptr())
    buf = torch.from_numpy(np.concatenate([sc.kvec.ravel(), sc.qvec.ravel(), sc.tvec.ravel(), sc.segs.ravel()])).to(dev)
    n = sc.n_images
    p = buf.data_ptr()
    ctx.set_scene_chunks([0], [p], [p + 32 * n], [p + 64 * n], [p + 88 * n])
    for i in mine:
        m = sc.matches_of(i)
        nb = list(m.keys())
        off = np.zeros(len(nb) + 1, np.int64); off[1:] = np.cumsum([len(m[k]) for k in nb])
        ctx.triangulate_image(i, nb, off, np.concatenate([m[k] for k in nb], 0))
    ctx.upload()
    ctx.refresh_scene_chunks()          # job uploaded: only its images and their neighbours are rebuilt
    ctx.run_device()
    ctx.download()
    compare_candidates(ctx.get_all_tris(), O.get_all_tris())
    compare_best(ctx.get_best(), O.get_best())


def test_incremental_batches(gpu_lib, oracle):
    """Images triangulated in several batches with result reads in between (streamed use, config 5):
    every flush merges into the persistent per-node results; the final tracks equal a one-batch run."""
    sc = small_scene(seed=6, n_views=14, n_segs=90, n_neighbors=7)
    cfg = syn.default_triangulation_cfg()
    from limap_amd import triangulation as tri
    T = tri.GlobalLineTriangulator(cfg)
    T.SetRanges(sc.ranges)
    T.InitArrays(sc.img_ids, sc.kvec, sc.qvec, sc.tvec, [sc.segs_of(i) for i in range(sc.n_images)])
    ids = [int(i) for i in sc.img_ids]
    for chunk in (ids[:5], ids[5:6], ids[6:]):
        for i in chunk:
            T.TriangulateImage(i, sc.matches_of(i))
        assert T.context().get_best()["has_best"].shape[0] == sc.seg_off[-1]  # forces a flush
    O = run_oracle(oracle, sc, cfg)
    compare_best(T.context().get_best(), O.get_best())
    compare_valid_edges(T.context().get_valid_edges(), O.get_valid_edges())
    T.ComputeLineTracks()
    compare_tracks(T.context().get_tracks(), O.ComputeLineTracks())


def test_two_rank_shards_merge_on_rank0(gpu_lib, oracle):
    """The multi-GPU flow on one GPU: two contexts ("ranks") share the scene, each triangulates its
    shard of the images, rank 1 exports its per-image results, rank 0 imports them and runs the tail.
    Tracks must equal the single-process run."""
    from limap_amd import _capi, dist as ltdist
    sc = small_scene(seed=7, n_views=12, n_segs=100, n_neighbors=6)
    cfg = syn.default_triangulation_cfg()
    ranks = []
    for r in range(2):
        ctx = _capi.Context(cfg_dict=cfg, device=0)
        ctx.set_ranges(*sc.ranges)
        ctx.init(sc.img_ids, sc.kvec, sc.qvec, sc.tvec, sc.seg_off, sc.segs)
        for i in ltdist.shard_images(sc.img_ids, r, 2):
            m = sc.matches_of(int(i))
            nb = list(m.keys())
            off = np.zeros(len(nb) + 1, np.int64); off[1:] = np.cumsum([len(m[k]) for k in nb])
            ctx.triangulate_image(int(i), nb, off, np.concatenate([m[k] for k in nb], 0))
        ranks.append(ctx)
    for i in ltdist.shard_images(sc.img_ids, 1, 2):
        ranks[0].import_image_results(ranks[1].export_image_results(int(i)))
    ranks[0].compute_tracks()
    O = run_oracle(oracle, sc, cfg)
    compare_best(ranks[0].get_best(), O.get_best())
    compare_valid_edges(ranks[0].get_valid_edges(), O.get_valid_edges())
    compare_tracks(ranks[0].get_tracks(), O.ComputeLineTracks())


_CFG_VARIANTS = [
    dict(linker3d_config=dict(score_th=0.5, th_angle=2.0, th_overlap=0.05, th_smartoverlap=0.1, th_smartangle=2.0,
                              th_perp=1.0, th_innerseg=1.0, th_scaleinv=0.002)),          # tight 3D gates
    dict(linker3d_config=dict(score_th=0.3, th_angle=45.0, th_overlap=0.05, th_smartoverlap=0.1, th_smartangle=2.0,
                              th_perp=1.0, th_innerseg=1.0, th_scaleinv=0.5)),            # loose: most pairs are evaluated
    dict(linker3d_config=dict(score_th=0.9, th_angle=89.9, th_overlap=0.05, th_smartoverlap=0.1, th_smartangle=2.0,
                              th_perp=1.0, th_innerseg=1.0, th_scaleinv=0.05)),           # angle guard near cos -> 0
    dict(linker2d_config=dict(score_th=0.8, th_angle=1.0, th_perp=0.5, th_overlap=0.3)),   # tight 2D gates
    dict(linker2d_config=dict(score_th=0.2, th_angle=30.0, th_perp=20.0, th_overlap=0.0)),
    dict(fullscore_th=3.0, max_valid_conns=3),                                              # ranked valid edges
    dict(fullscore_th=0.5, max_valid_conns=1, min_num_outer_edges=1),
    dict(use_endpoints_triangulation=True, add_halfpix=True, sensitivity_threshold=20.0),
    dict(num_outliers_aggregator=0, var2d=5.0, line_tri_angle_threshold=5.0, IoU_threshold=0.4),
]


@pytest.mark.parametrize("variant", range(len(_CFG_VARIANTS)))
@pytest.mark.parametrize("exhaustive", [False, True])
def test_config_variants_match_oracle(gpu_lib, oracle, variant, exhaustive):
    """Linker thresholds (they set the guards of the scoring sweep), selection knobs and generation switches
    away from cfgs/triangulation/default.yaml, matched and exhaustive, with and without ranges."""
    sc = small_scene(seed=20 + variant, n_views=9, n_segs=60, n_neighbors=4)
    if variant % 2:
        import dataclasses
        sc = dataclasses.replace(sc, ranges=None)
    cfg = syn.default_triangulation_cfg(debug_mode=True)
    cfg.update(_CFG_VARIANTS[variant])
    T = run_product(sc, cfg, exhaustive=exhaustive)
    O = run_oracle(oracle, sc, cfg, exhaustive=exhaustive)
    compare_candidates(T.context().get_all_tris(), O.get_all_tris())
    compare_best(T.context().get_best(), O.get_best())
    compare_valid_edges(T.context().get_valid_edges(), O.get_valid_edges())
    T.ComputeLineTracks()
    compare_tracks(T.context().get_tracks(), O.ComputeLineTracks())
    st, so = T.stats(), O.stats()
    for k in ("connections", "candidates", "valid_edges", "graph_nodes", "graph_edges", "tracks"):
        assert st[k] == so[k], k


@pytest.mark.parametrize("strategy", ["exhaustive", "avg"])
def test_merging_strategies_match_oracle(gpu_lib, oracle, strategy):
    """merging_strategy "exhaustive" / "avg" (global_line_triangulator.cc:306-316; merging/merging.cc:105-368):
    unions gated by LineLinker3d::check_connection in avgtest mode.  Same tracks as the oracle, and the gate
    actually bites on this scene (the tracks differ from the greedy strategy's)."""
    sc = small_scene(seed=5, n_views=20, n_segs=150, n_neighbors=8)
    cfg = syn.default_triangulation_cfg(merging_strategy=strategy)
    T = run_product(sc, cfg)
    O = run_oracle(oracle, sc, cfg)
    T.ComputeLineTracks()
    compare_tracks(T.context().get_tracks(), O.ComputeLineTracks())
    G = run_oracle(oracle, sc, syn.default_triangulation_cfg())
    G.ComputeLineTracks()
    assert not np.array_equal(O.get_tracks()["off"], G.get_tracks()["off"])


@pytest.mark.parametrize("exhaustive", [False, True])
def test_device_tail_equals_host_tail(gpu_lib, oracle, exhaustive):
    """ComputeLineTracks with the edge set + similarities built on the GPU and only the graph nodes downloaded
    (lt_kernels_tail.hip) gives the tracks of the host form (LT_TAIL_HOST=1: full download, std::set-order edges and
    score_3d on the host) bit for bit, and the oracle's; getters after the slim path still see every node."""
    import os
    sc = (small_scene(seed=9, n_views=9, n_segs=60, n_neighbors=4) if exhaustive
          else small_scene(seed=8, n_views=18, n_segs=130, n_neighbors=7))
    cfg = syn.default_triangulation_cfg()
    O = run_oracle(oracle, sc, cfg, exhaustive=exhaustive)
    ot = O.ComputeLineTracks()
    got = []
    for host in (False, True):
        if host:
            os.environ["LT_TAIL_HOST"] = "1"
        try:
            T = run_product(sc, cfg, exhaustive=exhaustive)
            T.ComputeLineTracks()
            got.append((T.context().get_tracks(), T))
        finally:
            os.environ.pop("LT_TAIL_HOST", None)
    a, b = got[0][0], got[1][0]
    for k in a:
        assert np.array_equal(a[k], b[k]), k
    compare_tracks(a, ot)
    T = got[0][1]  # the slim path ran; a getter now brings everything else down
    compare_best(T.context().get_best(), O.get_best())
    compare_valid_edges(T.context().get_valid_edges(), O.get_valid_edges())
    st, so = T.stats(), O.stats()
    for k in ("connections", "candidates", "pairs", "valid_edges", "graph_nodes", "graph_edges", "tracks"):
        assert st[k] == so[k], k
    compare_tracks(T.context().get_tracks(), ot)  # unchanged by the later download
