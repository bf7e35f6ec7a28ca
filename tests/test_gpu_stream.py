"""-m gpu: the STREAMED path (limap_amd/stream.py; BASELINE.json configs[4], runners/rome16k/triangulation.py:15-45) --
a model cut into chunks of images, every chunk run on a worker context that holds only the chunk's neighbour closure,
the per-image results imported into rank 0's accumulator, ONE ComputeLineTracks at the end -- against the oracle running
the reference's plain call sequence on the whole model."""
import json
import os

import numpy as np
import pytest

from limap_amd import stream as ltstream
from limap_amd import synthetic as syn

from digests import CASES, result_digests
from helpers import compare_best, compare_tracks, run_oracle

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "digests.json")


def _stream(sc, cfg, chunk_images, world=1, fine=False):
    """the whole job on one GPU: the ranks of a `world`-rank job run one after the other, the other ranks' results
    travel to rank 0 through the same pack / unpack as the collective's"""
    from limap_amd import dist as ltdist
    jobs = [ltstream.StreamedTriangulation(cfg, sc.img_ids, sc.kvec, sc.qvec, sc.tvec, sc.seg_off, sc.segs, sc.neighbors,
                                           sc.ranges, chunk_images=chunk_images, rank=r, world=world, device=0)
            for r in range(world)]
    for st in jobs:
        for ch in st.my_chunks():
            st.run_chunk(ch, sc.matches_of, fine_timers=fine)
    root = jobs[0]
    root.all_chunks = [rec for st in jobs for rec in st.per_chunk]
    for st in jobs[1:]:
        ints, dbls = ltstream.merge_blobs(st.results)
        # (through the Python packer as well: the two layouts are one)
        back = ltdist.pack_image_results(ltdist.unpack_image_results(ints, dbls))
        assert np.array_equal(back[0], ints) and np.array_equal(back[1], dbls)
        root._accumulator().import_images_packed(ints, dbls)
        root.n_imported += int(ints[0])
    root.world = 1  # (the exchange was done by hand)
    return root, root.finish()


@pytest.mark.parametrize("world,chunk", [(1, 7), (2, 5), (3, 16)])
def test_streamed_chunks_equal_the_oracles_whole_scene(gpu_lib, world, chunk):
    """closures that wrap around the trajectory, a last chunk that is shorter, more ranks than some rounds have chunks"""
    from oracle import oracle as ora
    sc = syn.make_scene(n_views=33, n_segs=60, n_neighbors=6, seed=11)
    cfg = syn.default_triangulation_cfg()
    cfg["add_halfpix"] = True  # cfgs/triangulation/rome16k.yaml
    root, A = _stream(sc, cfg, chunk, world)
    plan = root.chunks
    assert sorted(int(i) for c in plan for i in c.images) == sc.img_ids.tolist()
    for c in plan:
        need = set(int(i) for i in c.images) | {int(n) for i in c.images for n in sc.neighbors[int(i)]}
        assert set(c.closure.tolist()) == need and c.rank == c.index % world
        assert len(c.closure) < sc.n_images or chunk >= 16  # a chunk really holds less than the model
    O = run_oracle(ora, sc, cfg)
    ot = O.ComputeLineTracks()
    compare_best(A.get_best(), O.get_best())
    assert np.array_equal(A.get_num_tris(), O.get_num_tris())
    (aoff, ae), (boff, be) = A.get_valid_edges(), O.get_valid_edges()
    assert np.array_equal(aoff, boff)
    for g in range(len(aoff) - 1):  # the order inside a node feeds a std::set
        assert sorted(map(tuple, ae[aoff[g]:aoff[g + 1]])) == sorted(map(tuple, be[boff[g]:boff[g + 1]]))
    compare_tracks(A.get_tracks(), ot)


def test_streamed_1000x600_in_chunks_matches_the_oracles_digests(gpu_lib):
    """A fifth of configs[4]'s stand-in, five chunks of 200 images on two (emulated) ranks, against the digests the oracle
    wrote for the WHOLE scene in one piece (tests/golden/make_digests.py: best candidate of every node -- which and its
    geometry bit for bit --, valid-edge sets, track members, track lines)."""
    gold = json.load(open(GOLD))
    name = "stream_1000x600"
    assert name in gold, "tests/golden/digests.json lacks this case: run tests/golden/make_digests.py"
    case = CASES[name]
    sc = syn.make_scene(**case["scene"])
    cfg = syn.default_triangulation_cfg()
    cfg.update(case["cfg"])
    root, A = _stream(sc, cfg, 200, world=2, fine=True)
    assert len(root.chunks) == 5 and max(len(c.closure) for c in root.chunks) < 400
    st = A.stats()
    # connections / candidates are counted per run: summed over the chunks of both ranks they are the whole scene's
    got = result_digests(A.get_best(), A.get_valid_edges(), A.get_tracks(),
                         dict(st, connections=sum(r["connections"] for r in root.all_chunks),
                              candidates=sum(r["candidates"] for r in root.all_chunks),
                              valid_edges=len(A.get_valid_edges()[1])))
    assert int(A.get_num_tris().sum()) == got["counts"]["candidates"]
    assert got["counts"] == gold[name]["counts"]
    for k in ("best_src", "best_line", "valid_edges", "track_members", "track_lines"):
        assert got[k] == gold[name][k], k
    rec = root.per_chunk[0]
    assert rec["k_gates"] > 0 and rec["k_tri_rows"] > 0 and rec["k_score3"] > 0  # the per-kernel events of the bench leg
