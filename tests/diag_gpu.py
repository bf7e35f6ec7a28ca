# quick on-GPU diagnosis script: prints mismatch summaries instead of asserting
import sys, time, functools
print = functools.partial(print, flush=True)
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from limap_amd import synthetic as syn
from oracle import oracle as ora
from helpers import run_product, run_oracle, ulp_diff, edge_sets
ora.build()
for mode in ("matched", "exhaustive"):
    ex = mode == "exhaustive"
    sc = syn.make_scene(n_views=16 if not ex else 10, n_segs=120 if not ex else 70, n_neighbors=8 if not ex else 5, seed=0)
    cfg = syn.default_triangulation_cfg(debug_mode=True)
    print(mode, "scene ready")
    t0 = time.time(); T = run_product(sc, cfg, exhaustive=ex); print("  buffered", time.time() - t0)
    T.context().upload(); print("  uploaded", time.time() - t0)
    T.context().run_device(); print("  ran", time.time() - t0, T.timers())
    T.context().download(); print("  downloaded", time.time() - t0)
    ga = T.context().get_all_tris(); t1 = time.time()
    O = run_oracle(ora, sc, cfg, exhaustive=ex); oa = O.get_all_tris(); print('  oracle done', time.time() - t1)
    print(mode, "product %.3fs" % (t1 - t0), "stats", T.stats(), "timers", {k: round(v, 3) for k, v in T.timers().items()})
    print("  oracle stats", O.stats())
    same_off = np.array_equal(ga["off"], oa["off"])
    print("  cand off equal:", same_off, "n", len(ga["score"]), len(oa["score"]))
    if same_off:
        print("  src equal:", np.array_equal(ga["src"], oa["src"]), " line bit-exact:", np.array_equal(ga["line"], oa["line"]),
              " max ulp line:", ulp_diff(ga["line"], oa["line"]))
        nz = (ga["score"] == 0) != (oa["score"] == 0)
        print("  score zero-set mismatches:", int(nz.sum()), " max rel score err:",
              float(np.max(np.abs(ga["score"] - oa["score"]) / np.maximum(oa["score"], 1e-300))) if len(oa["score"]) else 0)
        bad = np.nonzero(np.abs(ga["score"] - oa["score"]) > 1e-9)[0]
        print("  n score mismatches > 1e-9:", len(bad), bad[:10], ga["score"][bad[:5]], oa["score"][bad[:5]])
    else:
        d = np.nonzero(np.diff(ga["off"]) != np.diff(oa["off"]))[0]
        print("  nodes with differing counts:", len(d), d[:10], np.diff(ga["off"])[d[:10]], np.diff(oa["off"])[d[:10]])
    gb, ob = T.context().get_best(), O.get_best()
    print("  best: has", np.array_equal(gb["has_best"], ob["has_best"]), "src", np.array_equal(gb["src"], ob["src"]),
          "line", np.array_equal(gb["line"], ob["line"]), "n src mismatch", int((gb["src"] != ob["src"]).any(1).sum()))
    (go, ge), (oo, oe) = T.context().get_valid_edges(), O.get_valid_edges()
    print("  valid edges: off equal", np.array_equal(go, oo), "E", len(ge), len(oe),
          "sets equal", edge_sets(go, ge) == edge_sets(oo, oe) if np.array_equal(go, oo) else None)
    T.ComputeLineTracks(); ot = O.ComputeLineTracks(); gt = T.context().get_tracks()
    print("  tracks:", len(gt["off"]) - 1, len(ot["off"]) - 1, "members equal",
          np.array_equal(gt["off"], ot["off"]) and np.array_equal(gt["image_ids"], ot["image_ids"]) and np.array_equal(gt["line_ids"], ot["line_ids"]) and np.array_equal(gt["node_ids"], ot["node_ids"]))
    if np.array_equal(gt["off"], ot["off"]) and len(gt["line"]):
        gl, ol = gt["line"], ot["line"]
        sw = np.concatenate([ol[:, 3:6], ol[:, :3]], 1)
        e = np.minimum(np.abs(gl[:, :6] - ol[:, :6]).max(1), np.abs(gl[:, :6] - sw).max(1))
        print("  track line max abs err (mod swap):", e.max(), " n swapped:", int((np.abs(gl[:, :6] - ol[:, :6]).max(1) > 1e-6).sum()))
    print("  stats after tracks", T.stats(), O.stats())
