"""Developer tool (GPU box): dump the per-tile timestamps of the scoring kernel from a -DLT_TRACE build to
gpurun_out/trace_<label>.npy (analysed offline):  LIMAP_AMD_LIB=... python tools/trace_dump.py label"""
import ctypes as C, os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import numpy as np
from limap_amd import synthetic as syn, triangulation as tri, _capi
tri._pb = None
sc = syn.make_scene(n_views=100, n_segs=500, n_neighbors=20, seed=0)
T = tri.GlobalLineTriangulator(syn.default_triangulation_cfg(debug_mode=True))
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
os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
np.save(os.path.join(root, "gpurun_out", "trace_%s.npy" % sys.argv[1]), buf.reshape(4, 65536, 4)[2:4, :16384].copy())
print(sys.argv[1], ctx.timers()["k_score3"])
