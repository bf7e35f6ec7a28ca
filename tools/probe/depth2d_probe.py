"""Probe (GPU box): how many sweep tests would a second depth key save in the exhaustive mode?
For every node of a small exhaustive scene: ordered pairs inside the start-depth window (what k_score3<sorted> sweeps now)
against pairs inside the start-depth window AND the end-depth window (a 2-D pruning by (z_start, z_end))."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from limap_amd import synthetic as syn, triangulation as tri

views, segs = int(os.environ.get("V", 40)), int(os.environ.get("S", 300))
sc = syn.make_scene(n_views=views, n_segs=segs, n_neighbors=min(20, views - 1), seed=0)
T = tri.GlobalLineTriangulator(syn.default_triangulation_cfg(debug_mode=True))
T.SetRanges(sc.ranges)
T.InitArrays(sc.img_ids, sc.kvec, sc.qvec, sc.tvec, [sc.segs_of(j) for j in range(sc.n_images)])
for i in sc.img_ids:
    T.TriangulateImageExhaustiveMatch(int(i), sc.neighbors[int(i)])
ctx = T.context()
ctx.upload(); ctx.run_device(); ctx.download()
allt = ctx.get_all_tris()
off, line = allt["off"], allt["line"]
print("nodes", len(off) - 1, "candidates", off[-1], "pairs_eval", int(ctx.timers()["pairs_eval"]))
# depths in the node's own view
R = np.stack([syn.quat_to_rot(q) for q in sc.qvec]); t = sc.tvec
n_per = np.diff(off)
node_img = np.repeat(np.arange(sc.n_images), [len(sc.segs_of(j)) for j in range(sc.n_images)])
g = 0.015 * (1 + 1e-6)
t1 = t2 = tall = 0
rng = np.random.default_rng(0)
nodes = rng.choice(len(n_per), size=min(4000, len(n_per)), replace=False)
for nd in nodes:
    a, b = off[nd], off[nd + 1]
    if b - a < 2:
        continue
    img = node_img[nd]
    L = line[a:b]
    zs = L[:, 0:3] @ R[img][2] + t[img][2]
    ze = L[:, 3:6] @ R[img][2] + t[img][2]
    ds = np.abs(zs[:, None] - zs[None, :]) <= g * (zs[:, None] + 1e-10)
    de = np.abs(ze[:, None] - ze[None, :]) <= g * (ze[:, None] + 1e-10)
    n = b - a
    tall += n * (n - 1)
    t1 += int(ds.sum()) - n
    t2 += int((ds & de).sum()) - n
print(f"ordered pairs {tall}, start-depth window {t1} ({t1 / tall:.3%}), start AND end window {t2} ({t2 / tall:.3%}), ratio {t1 / max(t2, 1):.1f}x")
