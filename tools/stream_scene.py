"""Developer tool: a large scene streamed through ONE context in batches of images (single-GPU stand-in for
BASELINE.json configs[4], "large model (>= 5k images) streamed triangulation").

    python tools/stream_scene.py [--views 5000] [--segs 500] [--batch 250] [--check]

Init holds the whole scene (cameras + segments: ~210 B per segment in HBM); the match rows -- the bulk of the input,
80 MB per 100 images at top-10 -- exist only one batch at a time: TriangulateImage x batch -> upload -> run ->
download (per-node results of the batch's images to the host), and ComputeLineTracks once at the end over all
nodes.  Prints one JSON line: totals, per-stage host/device times, images/s.  --check: the same 300-view scene in
one batch and in three -- identical tracks (results must not depend on the batching).
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from limap_amd import synthetic as syn, triangulation as tri  # noqa: E402


def run(scene, batch):
    T = tri.GlobalLineTriangulator(syn.default_triangulation_cfg())
    T.SetRanges(scene.ranges)
    t0 = time.perf_counter()
    T.InitArrays(scene.img_ids, scene.kvec, scene.qvec, scene.tvec, [scene.segs_of(j) for j in range(scene.n_images)])
    ctx = T.context()
    acc = {"init": time.perf_counter() - t0, "matches": 0.0, "buffer": 0.0, "upload": 0.0, "run": 0.0, "download": 0.0}
    dev_ms = 0.0
    conns = cands = 0
    ids = [int(i) for i in scene.img_ids]
    for b0 in range(0, len(ids), batch):
        part = ids[b0:b0 + batch]
        t = time.perf_counter()
        ms = [scene.matches_of(i) for i in part]  # stands in for reading matches_{id}.npy
        acc["matches"] += time.perf_counter() - t
        t = time.perf_counter()
        for i, m in zip(part, ms):
            T.TriangulateImage(i, m)
        acc["buffer"] += time.perf_counter() - t
        t = time.perf_counter()
        ctx.upload()
        acc["upload"] += time.perf_counter() - t
        t = time.perf_counter()
        ctx.run_device()
        acc["run"] += time.perf_counter() - t
        dev_ms += ctx.timers()["run"]
        t = time.perf_counter()
        ctx.download()
        acc["download"] += time.perf_counter() - t
        st = ctx.stats()
        conns += st["connections"]
        cands += st["candidates"]
    t = time.perf_counter()
    tracks = T.ComputeLineTracks()
    acc["compute_tracks"] = time.perf_counter() - t
    return T, tracks, acc, dev_ms, conns, cands


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--views", type=int, default=5000)
    ap.add_argument("--segs", type=int, default=500)
    ap.add_argument("--neighbors", type=int, default=20)
    ap.add_argument("--batch", type=int, default=250)
    ap.add_argument("--check", action="store_true")
    args = ap.parse_args()
    if args.check:
        sc = syn.make_scene(n_views=300, n_segs=200, n_neighbors=10, n_rooms=3, seed=5)
        T1, tr1, *_ = run(sc, 300)
        T3, tr3, *_ = run(sc, 100)
        a, b = T1.context().get_tracks(), T3.context().get_tracks()
        same = all(np.array_equal(a[k], b[k]) for k in ("off", "image_ids", "line_ids", "node_ids", "scores", "line"))
        print(json.dumps({"check": "one batch vs three", "tracks": len(tr1), "identical": bool(same)}))
        sys.exit(0 if same else 3)
    t0 = time.perf_counter()
    sc = syn.make_scene(n_views=args.views, n_segs=args.segs, n_neighbors=args.neighbors,
                        n_rooms=max(1, args.views // 100), seed=2)
    t_scene = time.perf_counter() - t0
    t0 = time.perf_counter()
    T, tracks, acc, dev_ms, conns, cands = run(sc, args.batch)
    wall = time.perf_counter() - t0
    stream = wall - acc["matches"]  # the generator of the synthetic matches is not part of the pipeline
    print(json.dumps({
        "workload": f"synthetic {args.views} views x {args.segs} segs, {args.neighbors} neighbours, matched top-10, "
                    f"{(args.views + args.batch - 1) // args.batch} batches of {args.batch} images through one context",
        "connections": int(conns), "candidates": int(cands), "tracks": len(tracks),
        "device_ms_total": round(dev_ms, 2), "candidates_per_s_device": round(cands / (dev_ms * 1e-3), 1),
        "host_s": {k: round(v, 3) for k, v in acc.items()}, "scene_generation_s": round(t_scene, 1),
        "wall_s_without_match_generation": round(stream, 3), "images_per_s": round(args.views / stream, 1)}))


if __name__ == "__main__":
    main()
