"""Where does the end-to-end wall-clock of the reference's API sequence go? (host side)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from limap_amd import synthetic as syn, triangulation as tri

sc = syn.make_scene(n_views=100, n_segs=500, n_neighbors=20, seed=0)
cfg = syn.default_triangulation_cfg()
matches = {int(i): sc.matches_of(int(i)) for i in sc.img_ids}
segs_list = [sc.segs_of(j) for j in range(sc.n_images)]
for rep in range(6):
    t0 = time.perf_counter()
    T = tri.GlobalLineTriangulator(cfg)
    T.SetRanges(sc.ranges)
    t1 = time.perf_counter()
    T.InitArrays(sc.img_ids, sc.kvec, sc.qvec, sc.tvec, segs_list)
    t2 = time.perf_counter()
    for i in sc.img_ids:
        T.TriangulateImage(int(i), matches[int(i)])
    t3 = time.perf_counter()
    T.context().compute_tracks()
    t4 = time.perf_counter()
    tr = T.context().get_tracks()
    t5 = time.perf_counter()
    tm = T.timers()
    print(f"rep{rep}: ctor {1e3*(t1-t0):.2f}  init {1e3*(t2-t1):.2f}  buffer(100x TriangulateImage) {1e3*(t3-t2):.2f}  "
          f"compute_tracks {1e3*(t4-t3):.2f} [buffer_c {tm['buffer']:.2f} upload {tm['upload']:.2f} run {tm['run']:.2f} download {tm['download']:.2f} tail {tm['tail']:.2f}]  "
          f"get_tracks {1e3*(t5-t4):.2f}  total {1e3*(t4-t0):.2f}")
    del T
