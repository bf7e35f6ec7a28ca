"""Developer tool (GPU box): per-call times of the post-triangulation chain (runners/line_triangulation.py:171-200) on the
bench scene, several repetitions: TrackSet.from_triangulator, filter_by_reprojection, remerge (per pass), the second
reprojection filter, sensitivity, overlap."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from limap_amd import merging, synthetic as syn, triangulation as tri  # noqa: E402

REMERGE_LINKER = dict(score_th=0.5, th_angle=5.0, th_overlap=0.001, th_smartoverlap=0.1, th_smartangle=1.0,
                      th_perp=1.0, th_innerseg=1.0, th_scaleinv=0.015)
sc = syn.make_scene(n_views=100, n_segs=500, n_neighbors=20, seed=0)
cfg = syn.default_triangulation_cfg()
matches = {int(i): sc.matches_of(int(i)) for i in sc.img_ids}
segs_list = [sc.segs_of(j) for j in range(sc.n_images)]
for rep in range(4):
    T = tri.GlobalLineTriangulator(cfg)
    T.SetRanges(sc.ranges)
    T.InitArrays(sc.img_ids, sc.kvec, sc.qvec, sc.tvec, segs_list)
    T.TriangulateAll(matches)
    T.ComputeLineTracks()
    lap = []
    t0 = time.perf_counter()

    def mark(name):
        global t0
        t1 = time.perf_counter()
        lap.append((name, 1e3 * (t1 - t0)))
        t0 = t1
    ts = merging.TrackSet.from_triangulator(T); mark("from_triangulator")
    ts.filter_by_reprojection(8.0, 5.0); mark("reproj1 (%d)" % len(ts))
    lcfg = merging._linker_cfg(REMERGE_LINKER)
    n = len(ts)
    while True:
        ts.ctx.chk(ts.L.lt_ts_remerge_once(ts.ctx.h, ts.h, C.byref(lcfg), 2)); mark("remerge pass (%d)" % len(ts))
        if len(ts) == n:
            break
        n = len(ts)
    ts.filter_by_reprojection(8.0, 5.0); mark("reproj2 (%d)" % len(ts))
    ts.filter_by_sensitivity(75.0, 3); mark("sensitivity (%d)" % len(ts))
    ts.filter_by_overlap(0.5, 3); mark("overlap (%d)" % len(ts))
    print("rep", rep, "total %.2f ms |" % sum(v for _, v in lap), " | ".join("%s %.2f" % kv for kv in lap))
    del ts, T
