"""Developer tool (GPU box): what does the FIRST call sequence of a process pay, and what does a tiny warm-up scene take
off it?   python tools/cold_probe.py [warm]"""
import os, sys, time
t_imp = time.perf_counter()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from limap_amd import synthetic as syn, triangulation as tri
t_imp = time.perf_counter() - t_imp


def run(sc, cfg, matches, segs_list, label):
    t0 = time.perf_counter()
    T = tri.GlobalLineTriangulator(cfg)
    T.SetRanges(sc.ranges)
    t1 = time.perf_counter()
    T.InitArrays(sc.img_ids, sc.kvec, sc.qvec, sc.tvec, segs_list)
    t2 = time.perf_counter()
    for i in sc.img_ids:
        T.TriangulateImage(int(i), matches[int(i)])
    t3 = time.perf_counter()
    tr = T.ComputeLineTracks()
    t4 = time.perf_counter()
    tm = T.timers()
    print(f"{label}: ctor {1e3*(t1-t0):.2f} init {1e3*(t2-t1):.2f} buffer {1e3*(t3-t2):.2f} compute_tracks {1e3*(t4-t3):.2f} "
          f"[upload {tm['upload']:.2f} run {tm['run']:.2f} download {tm['download']:.2f} tail {tm['tail']:.2f}] total {1e3*(t4-t0):.2f} ms, {len(tr)} tracks", flush=True)
    del T


cfg = syn.default_triangulation_cfg()
big = syn.make_scene(n_views=100, n_segs=500, n_neighbors=20, seed=0)
bm = {int(i): big.matches_of(int(i)) for i in big.img_ids}
bs = [big.segs_of(j) for j in range(big.n_images)]
print(f"import {1e3*t_imp:.1f} ms")
if len(sys.argv) > 1 and sys.argv[1] == "warm":
    import limap_amd
    t0 = time.perf_counter()
    limap_amd.warmup()   # a synthetic scene of the default shape (100 x 500, another seed) + a toy exhaustive scene
    print(f"warmup() {1e3*(time.perf_counter()-t0):.1f} ms", flush=True)
import gc
for rep in range(5):
    run(big, cfg, bm, bs, f"rep{rep}")
    if "gc" in sys.argv:
        t0 = time.perf_counter(); n = gc.collect(); print(f"   gc.collect() freed {n} objects in {1e3*(time.perf_counter()-t0):.2f} ms")
