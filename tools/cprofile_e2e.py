"""Developer tool (GPU box): cProfile of the reference's call sequence on the bench scene (host side of the end-to-end figure)."""
import cProfile, os, pstats, sys, time, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401  (as in bench.py)
from limap_amd import synthetic as syn, triangulation as tri

sc = syn.make_scene(n_views=100, n_segs=500, n_neighbors=20, seed=0)
cfg = syn.default_triangulation_cfg()
matches = {int(i): sc.matches_of(int(i)) for i in sc.img_ids}
segs_list = [sc.segs_of(j) for j in range(sc.n_images)]


def once():
    T = tri.GlobalLineTriangulator(cfg)
    T.SetRanges(sc.ranges)
    T.InitArrays(sc.img_ids, sc.kvec, sc.qvec, sc.tvec, segs_list)
    for i in sc.img_ids:
        T.TriangulateImage(int(i), matches[int(i)])
    return T.ComputeLineTracks()


for _ in range(3):
    once()
gc.collect(); gc.disable()
ts = []
for _ in range(5):
    t0 = time.perf_counter(); once(); ts.append(1e3 * (time.perf_counter() - t0))
print("e2e ms", [round(t, 2) for t in ts])
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    once()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(22)
