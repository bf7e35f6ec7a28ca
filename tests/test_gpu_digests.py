"""-m gpu: whole-scene runs at BASELINE.json's sizes against the ORACLE'S committed digests (tests/golden/digests.json,
written by tests/golden/make_digests.py in the build container -- the oracle needs minutes for these, the GPU 0.2 s / 4 ms):
best candidate of every node (which one and its geometry, bit-exact), valid-edge sets, track members, track lines."""
import json
import os

import pytest

from limap_amd import synthetic as syn

from digests import CASES, result_digests
from helpers import run_product

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "digests.json")


@pytest.mark.parametrize("name", sorted(CASES))
def test_whole_scene_digests_equal_the_oracles(gpu_lib, name):
    gold = json.load(open(GOLD))
    assert name in gold, "tests/golden/digests.json lacks this case: run tests/golden/make_digests.py"
    case = CASES[name]
    sc = syn.make_scene(**case["scene"])
    cfg = syn.default_triangulation_cfg()
    cfg.update(case.get("cfg", {}))
    T = run_product(sc, cfg, exhaustive=case["exhaustive"])
    T.ComputeLineTracks()
    ctx = T.context()
    got = result_digests(ctx.get_best(), ctx.get_valid_edges(), ctx.get_tracks(), T.stats())
    assert got["counts"] == gold[name]["counts"]
    for k in ("best_src", "best_line", "valid_edges", "track_members", "track_lines"):
        assert got[k] == gold[name][k], k
