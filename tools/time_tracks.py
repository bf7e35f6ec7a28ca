import time, numpy as np, sys
sys.path.insert(0,'/root/repo')
from limap_amd import synthetic as syn, triangulation as tri
sc = syn.make_scene(n_views=100, n_segs=500, n_neighbors=20, seed=0)
cfg = syn.default_triangulation_cfg()
matches = {int(i): sc.matches_of(int(i)) for i in sc.img_ids}
segs_list = [sc.segs_of(j) for j in range(sc.n_images)]
for rep in range(3):
    T = tri.GlobalLineTriangulator(cfg); T.SetRanges(sc.ranges)
    T.InitArrays(sc.img_ids, sc.kvec, sc.qvec, sc.tvec, segs_list)
    for i in sc.img_ids: T.TriangulateImage(int(i), matches[int(i)])
    t0=time.perf_counter(); tr = T.ComputeLineTracks(); t1=time.perf_counter()
    n = sum(t.count_lines() for t in tr); t2=time.perf_counter()
    m = sum(len(t.line2d_list) for t in tr); t3=time.perf_counter()
    print(f"ComputeLineTracks {1e3*(t1-t0):.2f} ms, count_lines {1e3*(t2-t1):.2f} ms, materialise line2d {1e3*(t3-t2):.2f} ms", len(tr), n, m)
    del T
