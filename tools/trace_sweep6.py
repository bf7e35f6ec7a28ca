"""Developer tool: per-tile phase breakdown of k_sweep6 from a -DLT_TRACE build (limap_amd/variants/libT.so, made by
   bash tools/build_variant.sh T -DLT_TRACE):  python tools/trace_score4.py   (on the GPU box)"""
import ctypes as C, os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
os.environ.setdefault("LIMAP_AMD_LIB", os.path.join(root, "limap_amd/variants/libT.so"))
import numpy as np
from limap_amd import synthetic as syn, triangulation as tri, _capi

sc = syn.make_scene(n_views=100, n_segs=500, n_neighbors=20, seed=0)
T = tri.GlobalLineTriangulator(syn.default_triangulation_cfg())
T.SetRanges(sc.ranges)
T.InitArrays(sc.img_ids, sc.kvec, sc.qvec, sc.tvec, [sc.segs_of(j) for j in range(sc.n_images)])
for i in sc.img_ids:
    T.TriangulateImage(int(i), sc.matches_of(int(i)))
ctx = T.context()
ctx.upload()
for _ in range(3):
    ctx.run_device()
L = _capi.load_library()
n = 4 * 4 * 65536
buf = np.zeros(n, dtype=np.uint64)
assert L.lt_debug_read_trace(buf.ctypes.data_as(C.c_void_p), C.c_size_t(n)) == 0
all_ = buf.reshape(4, 65536, 4).astype(np.int64)
t, x = all_[2], all_[3]
act = x[:, 3] > 0
t, x = t[act], x[act]
t0 = t[:, 0].min()
us = (t - t0) / 100.0
end = (x[:, 3] - t0) / 100.0
dense = x[:, 0] / 100.0
rounds = x[:, 1]
pairs = x[:, 2]
print("tiles", act.sum(), "kernel span us", end.max().round(1), "rounds", rounds.sum(), "pairs", pairs.sum(),
      "lanes per round", (pairs.sum() / max(rounds.sum(), 1)).round(1))
stage = us[:, 1] - us[:, 0]
body = us[:, 2] - us[:, 1]
sweep = body - dense
epi = end - us[:, 2]
tot = end - us[:, 0]
def line(name, d):
    print(f"{name:28s} sum ms {d.sum() / 1e3:7.2f}  share {d.sum() / tot.sum():5.1%}  per tile us pct 10/50/90/100:",
          np.percentile(d, [10, 50, 90, 100]).round(2))
line("prologue + first window", stage)
line("sweep (+ later windows)", sweep)
line("flush (atomic + pair writes)", dense)
line("ordered sums / store", epi)
line("tile total", tot)
print("per dense round us:", (dense.sum() / max(rounds.sum(), 1)).round(2))
print("tile time ms:", (tot.sum() / 1e3).round(1), " = resident waves x span if nothing idles: 2048 x span =",
      (2048 * end.max() / 1e3).round(1))
ev = np.concatenate([np.stack([us[:, 0], np.ones(len(us))], 1), np.stack([end, -np.ones(len(us))], 1)])
ev = ev[np.argsort(ev[:, 0])]
res = np.cumsum(ev[:, 1])
for q in (5, 20, 40, 60, 80, 100, 120, 140):
    idx = np.searchsorted(ev[:, 0], q)
    if idx < len(res):
        print(f"t={q}us tiles in flight {int(res[idx])}")
