"""Host-side laps of the batched call sequence (ctor, SetRanges, InitArrays, TriangulateAll, ComputeLineTracks).
LT_TAIL_TRACE=1 adds the native laps of lt_init and of the tail on stderr."""
import os, sys, time, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from limap_amd import synthetic as syn, triangulation as tri

sc = syn.make_scene(n_views=100, n_segs=500, n_neighbors=20, seed=0)
cfg = syn.default_triangulation_cfg()
matches = {int(i): sc.matches_of(int(i)) for i in sc.img_ids}
segs_list = [sc.segs_of(j) for j in range(sc.n_images)]
for rep in range(4):
    gc.collect(); gc.disable()
    print(f"--- rep {rep}", file=sys.stderr, flush=True)
    t0 = time.perf_counter()
    T = tri.GlobalLineTriangulator(cfg)
    t1 = time.perf_counter()
    T.SetRanges(sc.ranges)
    t2 = time.perf_counter()
    T.InitArrays(sc.img_ids, sc.kvec, sc.qvec, sc.tvec, segs_list)
    t3 = time.perf_counter()
    T.TriangulateAll(matches)
    t4 = time.perf_counter()
    tr = T.ComputeLineTracks()
    t5 = time.perf_counter()
    gc.enable()
    tm = T.timers()
    print(f"rep{rep}: ctor {1e3*(t1-t0):.3f} ranges {1e3*(t2-t1):.3f} init {1e3*(t3-t2):.3f} all {1e3*(t4-t3):.3f} "
          f"tracks {1e3*(t5-t4):.3f} [buffer_c {tm['buffer']:.3f} upload {tm['upload']:.3f} run {tm['run']:.3f} "
          f"download {tm['download']:.3f} tail {tm['tail']:.3f}] total {1e3*(t5-t0):.3f}", flush=True)
    del T
import cProfile, pstats
gc.collect()
pr = cProfile.Profile()
pr.enable()
T = tri.GlobalLineTriangulator(cfg); T.SetRanges(sc.ranges)
T.InitArrays(sc.img_ids, sc.kvec, sc.qvec, sc.tvec, segs_list)
T.TriangulateAll(matches)
tr = T.ComputeLineTracks()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
