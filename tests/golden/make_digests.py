"""Writes tests/golden/digests.json: the ORACLE's result digests (tests/digests.py) of the whole-scene runs that are
too long for the GPU box's test suite.  Run in the build container (minutes on 8 cores):
    python tests/golden/make_digests.py [case ...]"""
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from digests import CASES, result_digests  # noqa: E402
from limap_amd import synthetic as syn  # noqa: E402
from oracle import oracle as ora  # noqa: E402

out_path = os.path.join(HERE, "digests.json")
done = json.load(open(out_path)) if os.path.exists(out_path) else {}
ora.build()
ora.set_num_threads(os.cpu_count() or 1)
for name in (sys.argv[1:] or list(CASES)):
    case = CASES[name]
    t0 = time.time()
    sc = syn.make_scene(**case["scene"])
    cfg = syn.default_triangulation_cfg()
    cfg.update(case.get("cfg", {}))
    O = ora.OracleTriangulator(cfg, faithful=False)
    O.SetRanges(sc.ranges)
    O.Init(sc.img_ids, sc.kvec, sc.qvec, sc.tvec, sc.seg_off, sc.segs)
    for i in sc.img_ids:
        if case["exhaustive"]:
            O.TriangulateImageExhaustiveMatch(int(i), sc.neighbors[int(i)])
        else:
            O.TriangulateImage(int(i), sc.matches_of(int(i)))
    tracks = O.ComputeLineTracks()
    done[name] = result_digests(O.get_best(), O.get_valid_edges(), tracks, O.stats())
    done[name]["oracle_wall_s"] = round(time.time() - t0, 1)
    print(name, done[name], flush=True)
    with open(out_path, "w") as f:
        json.dump(done, f, indent=1, sort_keys=True)
