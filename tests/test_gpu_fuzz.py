"""-m gpu: randomised differential runs against the CPU oracle (what tools/fuzz_parity.py runs by the hundred, here a fixed
set of seeds so that the driver sees them) and the batching invariance of a streamed scene (tools/stream_scene.py --check)."""
import numpy as np
import pytest

from limap_amd import synthetic as syn

from helpers import compare_best, compare_candidates, compare_tracks, compare_valid_edges, run_oracle, run_product

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("k", range(20))
def test_random_scene_against_the_oracle(gpu_lib, oracle, k):
    """Random scene size, neighbour count, mode (every third: exhaustive) and gate / linker thresholds; candidates,
    arg-max, valid edges and tracks as in tests/test_gpu_parity.py.  Seeds 4000...: not the ones the tool has been run on."""
    seed = 4000 + k
    rng = np.random.default_rng(seed)
    if k % 4 == 3:  # larger scenes: tracks exist, the tail is exercised
        nv, ns, nn = int(rng.integers(16, 31)), int(rng.integers(100, 260)), int(rng.integers(6, 11))
    else:
        nv, ns, nn = int(rng.integers(5, 15)), int(rng.integers(20, 160)), int(rng.integers(2, 8))
    sc = syn.make_scene(n_views=nv, n_segs=ns, n_neighbors=min(nn, nv - 1), seed=seed)
    cfg = syn.default_triangulation_cfg(debug_mode=True)
    cfg["linker3d_config"]["th_angle"] = float(rng.choice([5.0, 10.0, 20.0]))
    cfg["linker3d_config"]["th_scaleinv"] = float(rng.choice([0.005, 0.015, 0.05]))
    cfg["IoU_threshold"] = float(rng.choice([0.05, 0.1, 0.3]))
    cfg["fullscore_th"] = float(rng.choice([1.0, 2.0]))
    cfg["max_valid_conns"] = int(rng.choice([1000, 4]))
    cfg["add_halfpix"] = bool(rng.integers(0, 2))
    ex = k % 3 == 2
    T = run_product(sc, cfg, exhaustive=ex)
    O = run_oracle(oracle, sc, cfg, exhaustive=ex)
    compare_candidates(T.context().get_all_tris(), O.get_all_tris())
    compare_best(T.context().get_best(), O.get_best())
    compare_valid_edges(T.context().get_valid_edges(), O.get_valid_edges())
    T.ComputeLineTracks()
    compare_tracks(T.context().get_tracks(), O.ComputeLineTracks())


def _stream(scene, batch):
    """One context, the images in batches: TriangulateImage x batch -> upload -> run -> download, tracks at the end."""
    from limap_amd import triangulation as tri
    T = tri.GlobalLineTriangulator(syn.default_triangulation_cfg())
    T.SetRanges(scene.ranges)
    T.InitArrays(scene.img_ids, scene.kvec, scene.qvec, scene.tvec, [scene.segs_of(j) for j in range(scene.n_images)])
    ctx = T.context()
    ids = [int(i) for i in scene.img_ids]
    for b0 in range(0, len(ids), batch):
        for i in ids[b0:b0 + batch]:
            T.TriangulateImage(i, scene.matches_of(i))
        ctx.upload(); ctx.run_device(); ctx.download()
    T.ComputeLineTracks()
    return ctx.get_best(), ctx.get_tracks(), T.stats()


def test_streamed_scene_does_not_depend_on_the_batching(gpu_lib):
    """BASELINE configs[4]'s shape (a large model streamed through one context) at 400 views x 300 segments: one batch,
    three batches and batches of 37 images give the same best candidates and the same tracks, bit for bit."""
    sc = syn.make_scene(n_views=400, n_segs=300, n_neighbors=12, seed=5)
    ref = _stream(sc, 400)
    assert ref[2]["tracks"] > 500
    for batch in (134, 37):
        got = _stream(sc, batch)
        for k in ("has_best", "src", "line", "score"):
            assert np.array_equal(ref[0][k], got[0][k]), (batch, k)
        for k in ("off", "image_ids", "line_ids", "node_ids", "scores", "line"):
            assert np.array_equal(ref[1][k], got[1][k]), (batch, k)
