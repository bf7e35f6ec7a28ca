"""Developer tool: per-(wave, item) phases of k_gates_ln from a -DLT_TRACE build (variants/libT.so).
Marks: 0 item start, 1 end of the gate loop, 2 survivors written, 3 behind the barrier."""
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
t = buf.reshape(4, 65536, 4)[0].astype(np.int64)
act = t[:, 3] > 0
t0 = t[act, 0].min()
t = (t[act] - t0) / 100.0
print("items x waves traced", act.sum(), "kernel span us", t[:, 3].max().round(1))
pc = [0, 10, 50, 90, 100]
print("start of item us      :", np.percentile(t[:, 0], pc).round(1))
print("gate loop us          :", np.percentile(t[:, 1] - t[:, 0], pc).round(2))
print("requests + survivors  :", np.percentile(t[:, 2] - t[:, 1], pc).round(2))
print("wait + barrier us     :", np.percentile(t[:, 3] - t[:, 2], pc).round(2))
print("item total us         :", np.percentile(t[:, 3] - t[:, 0], pc).round(2))
idx = np.nonzero(act)[0]
item = idx // 4
order = item % 4
for o in range(4):
    m = order == o
    print(f"order {o}: start {np.percentile(t[m, 0], [10, 50, 90]).round(1)} gate {np.percentile(t[m, 1] - t[m, 0], [10, 50, 90]).round(2)} "
          f"surv {np.percentile(t[m, 2] - t[m, 1], [10, 50, 90]).round(2)} barrier {np.percentile(t[m, 3] - t[m, 2], [10, 50, 90]).round(2)}")
wave = idx % 4
for w in range(8):
    m = wave == w
    print(f"wave {w}: gate {np.percentile(t[m, 1] - t[m, 0], [10, 50, 90]).round(2)} barrier {np.percentile(t[m, 3] - t[m, 2], [10, 50, 90]).round(2)}")
wg = item // 4
g = t[:, 1] - t[:, 0]
m0 = (order == 0) & (wave == 4)
print("order 0, wave 4: gate loop by workgroup index (mean over 25 consecutive workgroups)")
wgs = wg[m0]; gs = g[m0]; st0 = t[m0, 0]
o = np.argsort(wgs)
print(np.array([gs[o][k:k + 25].mean() for k in range(0, len(o), 25)]).round(1))
print("start:", np.array([st0[o][k:k + 25].mean() for k in range(0, len(o), 25)]).round(1))
slow = gs > 12
print("slow workgroups (order 0):", np.sort(wgs[slow])[:80])
