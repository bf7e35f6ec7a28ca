"""Developer tool (GPU box): stage times of ComputeLineTracks on resident results (LT_TAIL_TRACE), config 2 or 3.
   python tools/time_tail.py [--config3]"""
import os, sys, time
os.environ["LT_TAIL_TRACE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from limap_amd import synthetic as syn, triangulation as tri
if "--config3" in sys.argv:
    sc = syn.make_scene(n_views=1000, n_segs=1000, n_neighbors=20, n_rooms=4, n_gt=3000, seed=1, topk=10)
else:
    sc = syn.make_scene(n_views=100, n_segs=500, n_neighbors=20, seed=0)
T = tri.GlobalLineTriangulator(syn.default_triangulation_cfg())
T.SetRanges(sc.ranges)
T.InitArrays(sc.img_ids, sc.kvec, sc.qvec, sc.tvec, [sc.segs_of(j) for j in range(sc.n_images)])
T.TriangulateAll({int(i): sc.matches_of(int(i)) for i in sc.img_ids})
ctx = T.context()
ctx.upload()
for rep in range(4):
    ctx.run_device()
    ctx.sync()
    sys.stderr.write(f"--- rep {rep}\n")
    t0 = time.perf_counter()
    ctx.compute_tracks()
    sys.stderr.write(f"compute_tracks {1e3 * (time.perf_counter() - t0):.3f} ms, {ctx.stats()}\n")
