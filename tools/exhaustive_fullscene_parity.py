"""Whole-scene parity run of the EXHAUSTIVE mode at the benchmark size (100 views x 500 segments, 5e8 connections,
2.2e7 candidates): every image through TriangulateImageExhaustiveMatch on the GPU backend and on the CPU oracle, then
best candidate per node, valid-edge sets, track members and track lines compared as bench.py's cpu_parity does.
The oracle needs minutes for this (hoisted-invariant mode, OpenMP), so it is not part of the test suite; the result
is written as JSON:   python tools/exhaustive_fullscene_parity.py [out.json] [views segs neighbors]   (on the GPU box)"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

import bench
from limap_amd import synthetic as syn, triangulation as tri
from oracle import oracle as ora


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "exhaustive_fullscene_parity.json")
    views, segs, nn = (int(x) for x in sys.argv[2:5]) if len(sys.argv) > 4 else (100, 500, 20)
    sc = syn.make_scene(n_views=views, n_segs=segs, n_neighbors=nn, seed=0)
    cfg = syn.default_triangulation_cfg()
    t0 = time.perf_counter()
    T = tri.GlobalLineTriangulator(cfg)
    T.SetRanges(sc.ranges)
    T.InitArrays(sc.img_ids, sc.kvec, sc.qvec, sc.tvec, [sc.segs_of(i) for i in range(sc.n_images)])
    for i in sc.img_ids:
        T.TriangulateImageExhaustiveMatch(int(i), sc.neighbors[int(i)])
    T.ComputeLineTracks()
    gpu_s = time.perf_counter() - t0
    ora.build()
    threads = min(os.cpu_count() or 1, 32)
    ora.set_num_threads(threads)
    O = ora.OracleTriangulator(cfg, faithful=False)
    t0 = time.perf_counter()
    O.SetRanges(sc.ranges)
    O.Init(sc.img_ids, sc.kvec, sc.qvec, sc.tvec, sc.seg_off, sc.segs)
    for i in sc.img_ids:
        O.TriangulateImageExhaustiveMatch(int(i), sc.neighbors[int(i)])
    O.ComputeLineTracks()
    cpu_s = time.perf_counter() - t0
    ok, rep = bench.cpu_parity(T, O)
    rep.update(images=int(sc.n_images), views=views, segs_per_view=segs, n_neighbors=nn, mode="exhaustive",
               gpu_wall_s=gpu_s, cpu_wall_s=cpu_s, cpu_threads=threads, cpu_kind="oracle, per-camera invariants hoisted",
               stats_gpu=T.stats(), device_source_hash=bench.device_source_hash())
    os.makedirs(os.path.dirname(out_path), exist_ok=True)
    with open(out_path, "w") as f:
        json.dump(rep, f, indent=1)
    print(json.dumps({k: rep[k] for k in ("ok", "images", "candidates_gpu", "candidates_cpu", "tracks_gpu", "tracks_cpu",
                                          "n_swapped", "max_endpoint_rel_err", "gpu_wall_s", "cpu_wall_s")}))
    return 0 if ok else 3


if __name__ == "__main__":
    sys.exit(main())
