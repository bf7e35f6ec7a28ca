"""Developer tool: per-wave phases of k_tri_rows from a -DLT_TRACE build (variants/libT.so).
Marks: 0 wave start, 1 survivor counts read, 3 first round done, 2 end."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("LIMAP_AMD_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "limap_amd/variants/libT.so"))
import numpy as np
from limap_amd import synthetic as syn, triangulation as tri, _capi

tri._pb = None
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
t = buf.reshape(4, 65536, 4)[1].astype(np.int64)
act = (t[:, 2] > 0) & (t[:, 1] > 0) & (t[:, 3] > 0)
t0 = t[t[:, 0] > 0, 0].min()
t = (t[act] - t0) / 100.0
pc = [0, 10, 50, 90, 100]
print("rounds traced", act.sum(), "span us", t[:, 2].max().round(1))
print("start us              :", np.percentile(t[:, 0], pc).round(1))
print("row entry load us     :", np.percentile(t[:, 1] - t[:, 0], pc).round(2))
print("segments + compute us  :", np.percentile(t[:, 3] - t[:, 1], pc).round(2))
print("compaction + stores us :", np.percentile(t[:, 2] - t[:, 3], pc).round(2))
print("round total us        :", np.percentile(t[:, 2] - t[:, 0], pc).round(2))
for lo in range(0, 60, 10):
    m = (t[:, 0] >= lo) & (t[:, 0] < lo + 10)
    if m.sum():
        print(f"rounds starting in [{lo},{lo+10}) us: n={m.sum()} total {np.median(t[m,2]-t[m,0]):.2f} entry {np.median(t[m,1]-t[m,0]):.2f} compute {np.median(t[m,3]-t[m,1]):.2f} store {np.median(t[m,2]-t[m,3]):.2f}")
ev = np.concatenate([np.stack([t[:, 0], np.ones(len(t))], 1), np.stack([t[:, 2], -np.ones(len(t))], 1)])
ev = ev[np.argsort(ev[:, 0])]
res = np.cumsum(ev[:, 1])
for q in (2, 5, 10, 20, 30, 40, 50, 60):
    idx = np.searchsorted(ev[:, 0], q)
    if idx < len(res):
        print(f"t={q}us resident waves {int(res[idx])}")
