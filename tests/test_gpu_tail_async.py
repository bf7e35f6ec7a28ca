"""-m gpu: ComputeLineTracks in two halves (lt_compute_tracks_begin / _end): the device half of step k's tail is enqueued,
the next step's kernels go into the stream behind it, and the host half of step k runs while the device is busy --
same tracks as the one-call form, for every step of a streamed sequence."""
import numpy as np
import pytest

from limap_amd import synthetic as syn

pytestmark = pytest.mark.gpu


def _ctx(sc, cfg):
    from limap_amd import _capi
    ctx = _capi.Context(cfg_dict=cfg, device=0)
    ctx.set_ranges(*sc.ranges)
    ctx.init(sc.img_ids, sc.kvec, sc.qvec, sc.tvec, sc.seg_off, sc.segs)
    for i in sc.img_ids:
        m = sc.matches_of(int(i))
        nb = list(m.keys())
        off = np.zeros(len(nb) + 1, np.int64)
        off[1:] = np.cumsum([len(m[k]) for k in nb])
        ctx.triangulate_image(int(i), nb, off, np.concatenate([m[k] for k in nb], 0) if nb else np.zeros((0, 2), np.int32))
    ctx.upload()
    return ctx


def _same(a, b):
    for k in ("off", "image_ids", "line_ids", "node_ids", "scores", "line"):
        assert np.array_equal(a[k], b[k]), f"tracks[{k}] differ between the one-call and the two-half tail"


def test_tail_in_two_halves_with_the_next_step_in_between(gpu_lib):
    sc = syn.make_scene(n_views=24, n_segs=160, n_neighbors=8, seed=21)
    cfg = syn.default_triangulation_cfg()
    ref = _ctx(sc, cfg)
    ref.run_device()
    ref.compute_tracks()
    want = ref.get_tracks()
    assert len(want["off"]) > 20

    ctx = _ctx(sc, cfg)
    ctx.run_device(wait=False)
    ctx.compute_tracks_begin()
    for _ in range(3):  # steady state of a streaming caller: next step enqueued, previous tail collected
        ctx.run_device(wait=False)
        ctx.compute_tracks_end()
        _same(ctx.get_tracks(), want)
        ctx.compute_tracks_begin()
    ctx.compute_tracks_end()
    _same(ctx.get_tracks(), want)
    ctx.sync()
    # and the one-call form still works on the same context afterwards
    ctx.compute_tracks()
    _same(ctx.get_tracks(), want)


def test_two_half_tail_protocol_errors(gpu_lib):
    sc = syn.make_scene(n_views=10, n_segs=80, n_neighbors=5, seed=8)
    ctx = _ctx(sc, syn.default_triangulation_cfg())
    with pytest.raises(RuntimeError, match="without lt_compute_tracks_begin"):
        ctx.compute_tracks_end()
    ctx.compute_tracks_begin()
    with pytest.raises(RuntimeError, match="already in flight"):
        ctx.compute_tracks_begin()
    ctx.compute_tracks_end()
    # the host form of the tail (forced by the developer switch) has no two-half form
    import os
    os.environ["LT_TAIL_HOST"] = "1"
    try:
        c2 = _ctx(sc, syn.default_triangulation_cfg())
        with pytest.raises(RuntimeError, match="device form of the tail"):
            c2.compute_tracks_begin()
        c2.compute_tracks()
    finally:
        del os.environ["LT_TAIL_HOST"]


@pytest.mark.parametrize("min_outer", [1, 2, 3])
def test_node_filter_on_the_device(gpu_lib, oracle, min_outer):
    """filterNodeByNumOuterEdges (global_line_triangulator.cc:168-232) in the device form of the tail (k_outer_filter,
    round 5): valid flags, tracks and their lines equal the oracle's and the host form's -- one-call and two-half tail."""
    import os
    from helpers import compare_tracks, run_oracle
    sc = syn.make_scene(n_views=16, n_segs=120, n_neighbors=7, seed=33)
    cfg = syn.default_triangulation_cfg()
    cfg["min_num_outer_edges"] = min_outer
    cfg["max_valid_conns"] = 4
    O = run_oracle(oracle, sc, cfg)
    ot = O.ComputeLineTracks()
    dev = _ctx(sc, cfg)
    dev.run_device(wait=False)
    dev.compute_tracks_begin()
    dev.run_device(wait=False)
    dev.compute_tracks_end()
    got = dev.get_tracks()
    compare_tracks(got, ot)
    flags_dev = np.asarray(dev.get_valid_flags())
    assert 0 < int(flags_dev.sum()) < len(flags_dev), "the filter should remove some nodes and keep some"
    os.environ["LT_TAIL_HOST"] = "1"
    try:
        host = _ctx(sc, cfg)
        host.compute_tracks()
        _same(host.get_tracks(), got)
        assert np.array_equal(np.asarray(host.get_valid_flags()), flags_dev)
    finally:
        del os.environ["LT_TAIL_HOST"]
