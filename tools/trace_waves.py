"""Developer tool: per-wave timeline of k_gates from a -DLT_TRACE build (variants/libT.so)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("LIMAP_AMD_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "limap_amd/variants/libT.so"))
import numpy as np
from limap_amd import synthetic as syn, triangulation as tri, _capi

sc = syn.make_scene(n_views=100, n_segs=500, n_neighbors=20, seed=0)
cfg = syn.default_triangulation_cfg()
T = tri.GlobalLineTriangulator(cfg)
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
rc = L.lt_debug_read_trace(buf.ctypes.data_as(C.c_void_p), C.c_size_t(n))
assert rc == 0, rc
tt = buf.reshape(4, 65536, 4)
t = tt[int(os.environ.get("TRACE_KERNEL", "0"))]
act = t[:, 2] > 0
t0 = t[act, 0].min()
st = (t[act, 0] - t0) / 100.0   # us
tb = (t[act, 1].astype(np.int64) - t[act, 0].astype(np.int64)) / 100.0
en = (t[act, 2] - t0) / 100.0
dur = en - st
print("waves traced", act.sum(), "kernel span us", en.max())
print("start time pct (us):", np.percentile(st, [0, 10, 25, 50, 75, 90, 100]).round(1))
print("table phase us pct:", np.percentile(tb, [0, 10, 50, 90, 100]).round(2))
print("wave duration us pct:", np.percentile(dur, [0, 10, 50, 90, 100]).round(2))
hw = t[act, 3]
xcc = (hw >> np.uint64(32)).astype(np.int64)
cu = ((hw >> np.uint64(8)) & np.uint64(0xF)).astype(np.int64)
se = ((hw >> np.uint64(13)) & np.uint64(0x7)).astype(np.int64)
print("waves per xcc:", np.bincount(xcc))
# resident waves over time
ev = np.concatenate([np.stack([st, np.ones_like(st)], 1), np.stack([en, -np.ones_like(en)], 1)])
ev = ev[np.argsort(ev[:, 0])]
res = np.cumsum(ev[:, 1])
for q in (5, 20, 40, 60, 80, 100, 120):
    idx = np.searchsorted(ev[:, 0], q)
    if idx < len(res):
        print(f"t={q}us resident waves {int(res[idx])}")
