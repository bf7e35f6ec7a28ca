"""Developer tool: per-tile / per-unit phase timeline of the split scoring form (k_score3<.., kSplit> + k_dense8) from a
-DLT_TRACE build:  bash tools/build_variant.sh T -DLT_TRACE ; python tools/trace_split.py"""
import ctypes as C, os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
os.environ.setdefault("LIMAP_AMD_LIB", os.path.join(root, "limap_amd/variants/libT.so"))
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


def report(t, names, label, slots):
    act = t[:, 3] > 0
    t = t[act]
    us = (t - t[:, 0].min()) / 100.0
    print(f"{label}: items {act.sum()}  span us {us[:, 3].max().round(1)}")
    for k in range(3):
        d = us[:, k + 1] - us[:, k]
        print(f"  {names[k]:28s} us pct 0/10/50/90/100:", np.percentile(d, [0, 10, 50, 90, 100]).round(2), "sum ms", (d.sum() / 1e3).round(2))
    dur = us[:, 3] - us[:, 0]
    print("  item duration pct:", np.percentile(dur, [0, 10, 50, 90, 100]).round(2), "sum ms", (dur.sum() / 1e3).round(2),
          f"-> {dur.sum() / slots:.1f} us on {slots} slots")
    ev = np.concatenate([np.stack([us[:, 0], np.ones(len(us))], 1), np.stack([us[:, 3], -np.ones(len(us))], 1)])
    ev = ev[np.argsort(ev[:, 0])]
    res = np.cumsum(ev[:, 1])
    print("  items in flight at t:", {q: int(res[min(np.searchsorted(ev[:, 0], q), len(res) - 1)]) for q in (2, 5, 10, 20, 30, 40, 60, 80, 100)})
    return us


t2 = buf.reshape(4, 65536, 4)[2].astype(np.int64)
t3 = buf.reshape(4, 65536, 4)[3].astype(np.int64)
allt = ctx.get_all_tris()
n_per_node = np.diff(allt["off"])
cand_n = np.repeat(n_per_node, n_per_node)
act2 = t2[:, 3] > 0
tiles = np.nonzero(act2)[0]
us2 = (t2[act2] - t2[act2][:, 0].min()) / 100.0
nmax = np.array([cand_n[64 * t_: 64 * t_ + 64].max() for t_ in tiles])
nsum = np.array([cand_n[64 * t_: 64 * t_ + 64].sum() for t_ in tiles])
pro, swp = us2[:, 1] - us2[:, 0], us2[:, 2] - us2[:, 1]
for lo_, hi_ in ((0, 2), (2, 10), (10, 20), (20, 30), (30, 45), (45, 200)):
    m = (us2[:, 0] >= lo_) & (us2[:, 0] < hi_)
    if m.any():
        print(f"start in [{lo_},{hi_}) us: {m.sum()} tiles, prologue pct 10/50/90/100:", np.percentile(pro[m], [10, 50, 90, 100]).round(1),
              "sweep:", np.percentile(swp[m], [10, 50, 90, 100]).round(1), "nmax med", np.median(nmax[m]), "nsum med", np.median(nsum[m]))
for lo_, hi_ in ((0, 16), (16, 24), (24, 32), (32, 48), (48, 64), (64, 400)):
    m = (nmax >= lo_) & (nmax < hi_)
    if m.any():
        print(f"nmax in [{lo_},{hi_}): {m.sum()} tiles, sweep pct 10/50/90/100:", np.percentile(swp[m], [10, 50, 90, 100]).round(1))
report(t2, ["prologue+window", "sweep (+mid stores)", "final stores"], "sweep kernel (tiles)", int(os.environ.get("SLOTS1", 3072)))
if (t3[:, 3] > 0).any():
    report(t3, ["clear+loads", "rounds", "sums"], "k_dense8 (units)", int(os.environ.get("SLOTS2", 768)))
    a2, a3 = t2[t2[:, 3] > 0], t3[t3[:, 3] > 0]
    o = a2[:, 0].min()
    print("common clock (us from the first tile's start): tiles end", (a2[:, 3].max() - o) / 100.0, "| first unit starts", (a3[:, 0].min() - o) / 100.0,
          "| unit starts pct 10/50/90", np.percentile((a3[:, 0] - o) / 100.0, [10, 50, 90]).round(1), "| last unit ends", (a3[:, 3].max() - o) / 100.0)
    # busy workgroup-time of the dense role against the slots' time from the first unit's start to the last unit's end
    st_, en_ = (a3[:, 0] - o) / 100.0, (a3[:, 3] - o) / 100.0
    for lo_, hi_ in ((0, 20), (20, 30), (30, 40), (40, 50), (50, 60), (60, 70), (70, 80), (80, 90), (90, 120)):
        busy = np.clip(np.minimum(en_, hi_) - np.maximum(st_, lo_), 0, None).sum() / (hi_ - lo_)
        til = np.clip(np.minimum((a2[:, 3] - o) / 100.0, hi_) - np.maximum((a2[:, 0] - o) / 100.0, lo_), 0, None).sum() / (hi_ - lo_)
        print(f"  [{lo_},{hi_}) us: units busy on average {busy:.0f} workgroups, tiles busy {til:.0f} waves")
print("timers", {k: round(v, 4) for k, v in ctx.timers().items() if k in ("k_score3", "run")})
