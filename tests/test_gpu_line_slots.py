"""-m gpu: the line-slot form of stage A (k_gates_ln, one lane per line of the image; round 5) against the row-slot form
(k_gates) and against the oracle.  The form is chosen per upload -- every (image, neighbour) block sorted with contiguous
lines (the compressed row format), no run of equal line ids longer than 32 rows, neighbour tables within the LDS -- so the
cases below shape the match rows to sit on either side of every condition: ragged runs, blocks that start and end in the
middle of the image, images of 64 k +- 1 lines, more than 512 lines (several items per block, single-buffered tables),
a run of 33 rows (row-slot fallback decided on the device), an irregular block (plain form)."""
import os

import numpy as np
import pytest

from limap_amd import synthetic as syn
from limap_amd import triangulation as tri

from helpers import compare_best, compare_candidates, compare_valid_edges

pytestmark = pytest.mark.gpu


def _run(sc, cfg, matches, oracle=None):
    T = tri.GlobalLineTriangulator(cfg) if oracle is None else oracle.OracleTriangulator(cfg, faithful=False)
    if sc.ranges is not None:
        T.SetRanges(sc.ranges)
    if oracle is None:
        T.InitArrays(sc.img_ids, sc.kvec, sc.qvec, sc.tvec, [sc.segs_of(i) for i in range(sc.n_images)])
    else:
        T.Init(sc.img_ids, sc.kvec, sc.qvec, sc.tvec, sc.seg_off, sc.segs)
    for i in sc.img_ids:
        T.TriangulateImage(int(i), matches[int(i)])
    return T


def _results(T):
    ctx = T.context()
    allt, best, edges = ctx.get_all_tris(), ctx.get_best(), ctx.get_valid_edges()
    ctx.compute_tracks()
    return allt, best, edges, ctx.get_tracks(), ctx.timers()


def _same(a, b):
    for k in ("off", "src", "line", "score"):
        assert np.array_equal(a[0][k], b[0][k]), f"all_tris[{k}] differs between the slot forms"
    for k in ("has_best", "src", "line", "score"):
        assert np.array_equal(a[1][k], b[1][k]), f"best[{k}] differs between the slot forms"
    assert np.array_equal(a[2][0], b[2][0]) and np.array_equal(a[2][1], b[2][1])
    for k in ("off", "image_ids", "line_ids", "node_ids", "scores", "line"):
        assert np.array_equal(a[3][k], b[3][k]), f"tracks[{k}] differs between the slot forms"


def _ragged(sc, seed, keep=0.6, trim=True):
    """Every block keeps a random subset of its rows -- at least one per line, so the lines stay contiguous -- and, with
    `trim`, only the lines of a random range [lo, hi] of the image."""
    rng = np.random.default_rng(seed)
    out = {}
    for i in sc.img_ids:
        m = sc.matches_of(int(i))
        d = {}
        for nb, rows in m.items():
            rows = np.asarray(rows)
            if rows.shape[0] == 0:
                d[nb] = rows
                continue
            lines = rows[:, 0]
            first = np.r_[True, lines[1:] != lines[:-1]]
            sel = first | (rng.random(rows.shape[0]) < keep)
            if trim:
                lo, hi = sorted(rng.integers(int(lines.min()), int(lines.max()) + 1, size=2))
                sel &= (lines >= lo) & (lines <= hi)
            d[nb] = np.ascontiguousarray(rows[sel])
        out[int(i)] = d
    return out


@pytest.fixture
def env_clean():
    saved = os.environ.pop("LT_GEN_ROW_SLOTS", None)
    yield
    os.environ.pop("LT_GEN_ROW_SLOTS", None)
    if saved is not None:
        os.environ["LT_GEN_ROW_SLOTS"] = saved


def _both_forms(sc, cfg, matches, expect_ln=True):
    ln = _results(_run(sc, cfg, matches))
    os.environ["LT_GEN_ROW_SLOTS"] = "1"
    try:
        rows = _results(_run(sc, cfg, matches))
    finally:
        del os.environ["LT_GEN_ROW_SLOTS"]
    # timers[20]: 1 when the run used the line-slot form
    assert rows[4]["line_slots"] == 0.0
    assert ln[4]["line_slots"] == (1.0 if expect_ln else 0.0), "unexpected slot form"
    _same(ln, rows)
    return ln


@pytest.mark.parametrize("n_segs", [63, 64, 65, 130, 500])
def test_line_slots_equal_row_slots_full_topk(gpu_lib, env_clean, n_segs):
    sc = syn.make_scene(n_views=8, n_segs=n_segs, n_neighbors=4, seed=100 + n_segs)
    cfg = syn.default_triangulation_cfg(debug_mode=True)
    matches = {int(i): sc.matches_of(int(i)) for i in sc.img_ids}
    ln = _both_forms(sc, cfg, matches)
    assert ln[0]["off"][-1] > 0


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_line_slots_ragged_runs_vs_oracle(gpu_lib, env_clean, seed):
    from oracle import oracle as ora
    sc = syn.make_scene(n_views=10, n_segs=150 + 37 * seed, n_neighbors=5, seed=40 + seed)
    cfg = syn.default_triangulation_cfg(debug_mode=True)
    matches = _ragged(sc, seed)
    _both_forms(sc, cfg, matches)
    T = _run(sc, cfg, matches)
    O = _run(sc, cfg, matches, oracle=ora)
    compare_candidates(T.context().get_all_tris(), O.get_all_tris())
    compare_best(T.context().get_best(), O.get_best())
    compare_valid_edges(T.context().get_valid_edges(), O.get_valid_edges())


def test_line_slots_many_lines_single_buffer(gpu_lib, env_clean):
    """More than 512 segments per image: several items per block, neighbour tables beyond 40 KB (one LDS buffer)."""
    sc = syn.make_scene(n_views=6, n_segs=700, n_neighbors=3, seed=77)
    cfg = syn.default_triangulation_cfg(debug_mode=True)
    matches = _ragged(sc, 5, keep=0.8, trim=False)
    _both_forms(sc, cfg, matches)


def test_long_run_falls_back_to_row_slots(gpu_lib, env_clean):
    """One line with 33 rows in one block: the device flags the run, the upload takes the row-slot form."""
    sc = syn.make_scene(n_views=6, n_segs=90, n_neighbors=3, seed=9)
    cfg = syn.default_triangulation_cfg(debug_mode=True)
    matches = {int(i): dict(sc.matches_of(int(i))) for i in sc.img_ids}
    i0 = int(sc.img_ids[2])
    nb0 = sorted(matches[i0])[1]
    rows = np.asarray(matches[i0][nb0])
    line = int(rows[len(rows) // 2, 0])
    k = int(np.searchsorted(rows[:, 0], line, side="right"))
    n_nb = sc.segs_of(list(sc.img_ids).index(nb0)).shape[0]
    have = int((rows[:, 0] == line).sum())
    extra = np.stack([np.full(33 - have, line), np.arange(33 - have) % n_nb], axis=1).astype(rows.dtype)
    matches[i0][nb0] = np.ascontiguousarray(np.concatenate([rows[:k], extra, rows[k:]]))
    assert int((matches[i0][nb0][:, 0] == line).sum()) == 33
    _both_forms(sc, cfg, matches, expect_ln=False)
    # 32 rows still fit
    matches[i0][nb0] = np.ascontiguousarray(np.concatenate([rows[:k], extra[:-1], rows[k:]]))
    _both_forms(sc, cfg, matches, expect_ln=True)


def test_irregular_block_takes_row_slots(gpu_lib, env_clean):
    """A block that skips a line (plain row form): not a line-slot job."""
    sc = syn.make_scene(n_views=6, n_segs=90, n_neighbors=3, seed=10)
    cfg = syn.default_triangulation_cfg(debug_mode=True)
    matches = {int(i): dict(sc.matches_of(int(i))) for i in sc.img_ids}
    i0 = int(sc.img_ids[1])
    nb0 = sorted(matches[i0])[0]
    rows = np.asarray(matches[i0][nb0])
    matches[i0][nb0] = np.ascontiguousarray(rows[rows[:, 0] != 17])
    _both_forms(sc, cfg, matches, expect_ln=False)
