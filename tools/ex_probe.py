"""Developer tool: five runs of the full-size exhaustive job on one context -- per-run stage times and the
staging need / capacity of the one-pass form (lt_get_timers [17], [18])."""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from limap_amd import synthetic as syn, triangulation as tri
sc = syn.make_scene(n_views=100, n_segs=500, n_neighbors=20, seed=0)
T = tri.GlobalLineTriangulator(syn.default_triangulation_cfg())
T.SetRanges(sc.ranges)
T.InitArrays(sc.img_ids, sc.kvec, sc.qvec, sc.tvec, [sc.segs_of(j) for j in range(sc.n_images)])
for i in sc.img_ids:
    T.TriangulateImageExhaustiveMatch(int(i), sc.neighbors[int(i)])
ctx = T.context()
ctx.upload()
for k in range(5):
    ctx.run_device()
    t = ctx.timers()
    print(k, {a: round(t[a], 3) for a in ("run", "gen", "compact", "score", "ex_slots", "ex_cap")}, ctx.stats()["candidates"])
