import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from limap_amd import synthetic as syn, triangulation as tri
cfg = syn.default_triangulation_cfg()
scs = [syn.make_scene(n_views=10 + 3 * k, n_segs=60 + 20 * k, n_neighbors=4 + k, seed=k) for k in range(4)]
free0 = torch.cuda.mem_get_info()[0]
n_tr = []
t0 = time.time()
for it in range(60):
    sc = scs[it % 4]
    T = tri.GlobalLineTriangulator(cfg)
    T.SetRanges(sc.ranges)
    T.InitArrays(sc.img_ids, sc.kvec, sc.qvec, sc.tvec, [sc.segs_of(i) for i in range(sc.n_images)])
    for i in sc.img_ids:
        if it % 3 == 2:
            T.TriangulateImageExhaustiveMatch(int(i), sc.neighbors[int(i)])
        else:
            T.TriangulateImage(int(i), sc.matches_of(int(i)))
    n_tr.append(len(T.ComputeLineTracks()))
    del T
free1 = torch.cuda.mem_get_info()[0]
print("60 contexts ok in %.1fs, tracks %s, device memory held by the caches: %.1f MB" % (time.time() - t0, n_tr[:8], (free0 - free1) / 1e6))
assert n_tr[0:4] == n_tr[12:16]
