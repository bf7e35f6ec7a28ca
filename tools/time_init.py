"""Developer tool (GPU box): where the constructor + Init of a scene go (LT_TAIL_TRACE laps of lt_init on stderr + the Python side).
   python tools/time_init.py [--config3] [--torch]     (--torch: `import torch` first, the condition of the bench process)"""
import os, sys, time
os.environ["LT_TAIL_TRACE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if "--torch" in sys.argv:
    import torch
import numpy as np
from limap_amd import synthetic as syn, triangulation as tri
if "--config3" in sys.argv:
    sc = syn.make_scene(n_views=1000, n_segs=1000, n_neighbors=20, n_rooms=4, n_gt=3000, seed=1, topk=10)
else:
    sc = syn.make_scene(n_views=100, n_segs=500, n_neighbors=20, seed=0)
segs_list = [sc.segs_of(j) for j in range(sc.n_images)]
cfg = syn.default_triangulation_cfg()
for rep in range(6):
    sys.stderr.write(f"--- rep {rep}\n")
    t0 = time.perf_counter()
    T = tri.GlobalLineTriangulator(cfg)
    T.SetRanges(sc.ranges)
    t1 = time.perf_counter()
    T.InitArrays(sc.img_ids, sc.kvec, sc.qvec, sc.tvec, segs_list)
    t2 = time.perf_counter()
    sys.stderr.write(f"ctor {1e3 * (t1 - t0):.3f} ms, InitArrays {1e3 * (t2 - t1):.3f} ms\n")
    del T
