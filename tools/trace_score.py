"""Developer tool: per-wave phase timeline of the FUSED k_score3 from a -DLT_TRACE build (variants/libT.so):
   make -C limap_amd/csrc BUILD=build_T OUT=../variants/libT.so EXTRA=-DLT_TRACE"""
import ctypes as C, os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
os.environ.setdefault("LIMAP_AMD_LIB", os.path.join(root, "limap_amd/variants/libT.so"))
os.environ["LT_ENABLE_TEST_SWITCHES"] = "1"
os.environ["LT_SCORE_FUSED"] = "1"  # this tool reads the FUSED kernel's marks (the two-kernel form: tools/trace_split.py)
import numpy as np
from limap_amd import synthetic as syn, triangulation as tri, _capi

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
t = buf.reshape(4, 65536, 4)[2].astype(np.int64)
act = t[:, 3] > 0
t = t[act]
t0 = t[:, 0].min()
us = (t - t0) / 100.0
print("waves", act.sum(), "kernel span us", us[:, 3].max().round(1))
names = ["prologue+window", "sweep (+mid drains)", "final drain"]
for k in range(3):
    d = us[:, k + 1] - us[:, k]
    print(f"{names[k]:22s} us pct 0/10/50/90/100:", np.percentile(d, [0, 10, 50, 90, 100]).round(2), "sum ms", (d.sum() / 1e3).round(2))
dur = us[:, 3] - us[:, 0]
print("wave duration pct:", np.percentile(dur, [0, 10, 50, 90, 100]).round(2), "sum ms", (dur.sum() / 1e3).round(2))
ev = np.concatenate([np.stack([us[:, 0], np.ones(len(us))], 1), np.stack([us[:, 3], -np.ones(len(us))], 1)])
ev = ev[np.argsort(ev[:, 0])]
res = np.cumsum(ev[:, 1])
for q in (5, 20, 40, 60, 80, 100, 120, 140):
    idx = np.searchsorted(ev[:, 0], q)
    if idx < len(res):
        print(f"t={q}us resident waves {int(res[idx])}")
print("start time pct:", np.percentile(us[:, 0], [0, 10, 25, 50, 75, 90, 100]).round(1))
# duration by start-time bin, and by the tile's largest node
order = np.argsort(us[:, 0])
for lo_, hi_ in ((0, 5), (5, 30), (30, 60), (60, 90), (90, 110), (110, 200)):
    m = (us[:, 0] >= lo_) & (us[:, 0] < hi_)
    if m.any():
        print(f"start in [{lo_},{hi_}) us: {m.sum()} tiles, duration pct 10/50/90/100:", np.percentile(dur[m], [10, 50, 90, 100]).round(1))
allt = ctx.get_all_tris()
off = allt["off"]
n_per_node = np.diff(off)
cand_n = np.repeat(n_per_node, n_per_node)
tiles = np.nonzero(act)[0]
nmax = np.array([cand_n[64 * t_: 64 * t_ + 64].max() for t_ in tiles])
for lo_, hi_ in ((0, 16), (16, 32), (32, 64), (64, 128), (128, 400)):
    m = (nmax >= lo_) & (nmax < hi_)
    if m.any():
        print(f"nmax in [{lo_},{hi_}): {m.sum()} tiles, start pct 0/50/100:", np.percentile(us[m, 0], [0, 50, 100]).round(1),
              "duration pct 10/50/90/100:", np.percentile(dur[m], [10, 50, 90, 100]).round(1))
nsum = np.array([cand_n[64 * t_: 64 * t_ + 64].sum() for t_ in tiles])
print("corr(duration, sum n) =", np.corrcoef(dur, nsum)[0, 1].round(3), " corr(duration, nmax) =", np.corrcoef(dur, nmax)[0, 1].round(3))
qs = np.percentile(nsum, [0, 25, 50, 75, 90, 97, 100])
for lo_, hi_ in zip(qs[:-1], qs[1:]):
    m = (nsum >= lo_) & (nsum <= hi_)
    print(f"sum n in [{int(lo_)},{int(hi_)}]: {m.sum()} tiles, duration pct 10/50/90/100:", np.percentile(dur[m], [10, 50, 90, 100]).round(1))

# ---- what would another tile ORDER buy?  list scheduling of the measured tile durations on the resident waves ----
import heapq
def simulate(order, n_workers=2048):
    h = [0.0] * n_workers
    heapq.heapify(h)
    end = 0.0
    for t_ in order:
        s = heapq.heappop(h)
        e = s + dur[t_]
        end = max(end, e)
        heapq.heappush(h, e)
    return end
nt = len(dur)
nat = np.arange(nt)
print("simulated kernel span (us), 2048 waves: natural order", round(simulate(nat), 1),
      "| by sum n descending", round(simulate(np.argsort(-nsum, kind="stable")), 1),
      "| by TRUE duration descending (bound)", round(simulate(np.argsort(-dur, kind="stable")), 1),
      "| ideal", round(dur.sum() / 2048, 1))
for frac_last in (0.2, 0.3, 0.5):
    thr = np.quantile(nsum, frac_last)
    cheap = nsum <= thr
    for frac_first in (0.0, 0.1, 0.2):
        thr2 = np.quantile(nsum, 1 - frac_first) if frac_first > 0 else np.inf
        heavy = nsum >= thr2
        order = np.concatenate([nat[heavy], nat[~heavy & ~cheap], nat[cheap]])
        print(f"  heaviest {frac_first:.0%} first, cheapest {frac_last:.0%} last:", round(simulate(order), 1))
# per-lane predictor: sum over lanes of n (= sum n) vs sum over the tile's nodes of n^2
# static split of the predicted-heavy tiles into R virtual tiles (each repeats the prologue, shares sweep + drain)
pro = us[:, 1] - us[:, 0]
for frac in (0.1, 0.2, 0.3, 0.5, 1.0):
    for R in (2, 4):
        thr = np.quantile(nsum, 1 - frac) if frac < 1.0 else -1
        d2 = []
        for t_ in range(nt):
            if nsum[t_] >= thr:
                d2 += [pro[t_] + (dur[t_] - pro[t_]) / R] * R
            else:
                d2.append(dur[t_])
        d2 = np.array(d2)
        h = [0.0] * 2048
        heapq.heapify(h)
        end = 0.0
        for x in d2:
            s = heapq.heappop(h); e = s + x; end = max(end, e); heapq.heappush(h, e)
        print(f"split top {frac:.0%} by sum n into {R}: span {end:.1f} us, wave time {d2.sum()/1e3:.1f} ms (ideal {d2.sum()/2048:.1f})")
