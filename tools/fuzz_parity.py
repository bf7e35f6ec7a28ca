"""Developer tool: randomised differential run, product (HIP) against the CPU oracle (test infrastructure) on
random small scenes / modes / a few configuration knobs.  usage (GPU box): python tools/fuzz_parity.py [n] [seed0] [big|wide]
("wide": 205-260 views with 20 neighbours each -- more than 4 096 (image, neighbour) blocks, the regime in which stage B claims its units)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from limap_amd import synthetic as syn
from oracle import oracle as ora
from helpers import compare_best, compare_candidates, compare_tracks, compare_valid_edges, run_oracle, run_product

n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
big = len(sys.argv) > 3 and sys.argv[3] == "big"  # larger scenes: tracks exist, the tail is exercised
wide = len(sys.argv) > 3 and sys.argv[3] == "wide"
ora.build()
bad = 0
for k in range(n):
    rng = np.random.default_rng(seed0 + k)
    if wide:
        nv, ns, nn = int(rng.integers(205, 261)), int(rng.integers(20, 61)), 20
    elif big:
        nv, ns, nn = int(rng.integers(16, 31)), int(rng.integers(100, 260)), int(rng.integers(6, 11))
    else:
        nv, ns, nn = int(rng.integers(5, 15)), int(rng.integers(20, 160)), int(rng.integers(2, 8))
    sc = syn.make_scene(n_views=nv, n_segs=ns, n_neighbors=min(nn, nv - 1), seed=seed0 + k)
    cfg = syn.default_triangulation_cfg(debug_mode=True)
    cfg["linker3d_config"]["th_angle"] = float(rng.choice([5.0, 10.0, 20.0]))
    cfg["linker3d_config"]["th_scaleinv"] = float(rng.choice([0.005, 0.015, 0.05]))
    cfg["IoU_threshold"] = float(rng.choice([0.05, 0.1, 0.3]))
    cfg["fullscore_th"] = float(rng.choice([1.0, 2.0]))
    cfg["max_valid_conns"] = int(rng.choice([1000, 4]))
    cfg["add_halfpix"] = bool(rng.integers(0, 2))
    # round 6: the 2D linker's terms and thresholds (pair_score_fused: which q binds, the gate bands), the node filter
    rng2 = np.random.default_rng([seed0 + k, 6])
    l2 = cfg["linker2d_config"]
    l2["th_angle"] = float(rng2.choice([3.0, 5.0, 8.0]))
    l2["th_perp"] = float(rng2.choice([1.0, 2.0, 4.0]))
    l2["th_overlap"] = float(rng2.choice([0.02, 0.05, 0.2]))
    l2["score_th"] = float(rng2.choice([0.3, 0.5, 0.7]))
    if rng2.integers(0, 4) == 0:
        l2["use_smartangle"] = False
    if rng2.integers(0, 6) == 0:
        l2["use_perp"] = False
    cfg["linker3d_config"]["score_th"] = float(rng2.choice([0.4, 0.5, 0.6]))
    if rng2.integers(0, 5) == 0:
        cfg["min_num_outer_edges"] = int(rng2.integers(1, 3))
    ex = bool(k % 3 == 2) and not wide  # (wide: the matched mode is what the block count matters for)
    try:
        T = run_product(sc, cfg, exhaustive=ex)
        O = run_oracle(ora, sc, cfg, exhaustive=ex)
        compare_candidates(T.context().get_all_tris(), O.get_all_tris())
        compare_best(T.context().get_best(), O.get_best())
        compare_valid_edges(T.context().get_valid_edges(), O.get_valid_edges())
        T.ComputeLineTracks()
        compare_tracks(T.context().get_tracks(), O.ComputeLineTracks())
        st = T.stats()
        print(f"ok   seed {seed0 + k}: {nv} views x {ns} segs nn {nn} {'exhaustive' if ex else 'matched'}: "
              f"{st['candidates']} candidates, {st['tracks']} tracks")
    except AssertionError as e:
        bad += 1
        print(f"FAIL seed {seed0 + k}: {nv} views x {ns} segs nn {nn} ex={ex}: {str(e)[:200]}")
print("failures:", bad)
sys.exit(1 if bad else 0)
