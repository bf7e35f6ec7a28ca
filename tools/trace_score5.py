"""Developer tool: per-wave counters of k_score5 from a -DLT_TRACE build (limap_amd/variants/libT.so)."""
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
act = t[:, 3] > 0
t, x = t[act], x[act]
t0 = t[:, 0].min()
start = (t[:, 0] - t0) / 100.0
fin0 = (t[:, 1] - t0) / 100.0
end = (t[:, 3] - t0) / 100.0
dense = x[:, 0] / 100.0
rounds, pairs = x[:, 1], x[:, 2]
sleeps, tiles = x[:, 3] & 0xFFFFFFFF, x[:, 3] >> 32
print("waves", act.sum(), "span us", end.max().round(1), "tiles", tiles.sum(), "rounds", rounds.sum(), "pairs", pairs.sum(),
      "lanes/round", (pairs.sum() / max(rounds.sum(), 1)).round(1), "sleeps", sleeps.sum())
print("per wave: life us pct 0/50/100", np.percentile(end - start, [0, 50, 100]).round(1),
      " dense us pct", np.percentile(dense, [0, 50, 100]).round(1), " dense share of life", (dense.sum() / (end - start).sum()).round(3))
print("per round us", (dense.sum() / rounds.sum()).round(2), " sleeps per wave pct", np.percentile(sleeps, [0, 50, 90, 100]))
print("entered final state at us pct 0/10/50/90/100:", np.percentile(fin0, [0, 10, 50, 90, 100]).round(1),
      " time in final state sum ms", ((end - fin0).sum() / 1e3).round(2), "of", ((end - start).sum() / 1e3).round(2))
print("rounds per wave pct", np.percentile(rounds, [0, 10, 50, 90, 100]), " tiles per wave pct", np.percentile(tiles, [0, 10, 50, 90, 100]))
acc = np.concatenate([all_[0][act], all_[1][act][:, :3]], 1) / 100.0
names = ["prologue", "stage", "sweep", "flush", "wait", "dense(+claim)", "sums"]
life = (end - start).sum()
for k, nm in enumerate(names):
    print(f"{nm:14s} sum ms {acc[:, k].sum() / 1e3:8.2f}  share {acc[:, k].sum() / life:6.1%}  per tile us {acc[:, k].sum() / tiles.sum():6.2f}")
