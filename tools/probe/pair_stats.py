"""Developer probe (CPU only): pair statistics of scoreOneNode on a synthetic scene -- node-size histogram, share of the
ordered pairs that pass the conservative sweep guards, and the gate of pair_score at which the dense pairs end.
    python tools/probe/pair_stats.py [views segs neighbors topk]
"""
import ctypes as C
import math
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from limap_amd import synthetic as syn  # noqa: E402
from oracle import oracle as ora  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(HERE, "pair_stats.so")
src = os.path.join(HERE, "pair_stats.cpp")
if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-D__HIP_PLATFORM_AMD__",
                           "-I/opt/rocm/include", src, "-o", so])
L = C.CDLL(so)

views, segs, nn, topk = (list(map(int, sys.argv[1:5])) + [100, 500, 20, 10][len(sys.argv) - 1:])[:4]
sc = syn.make_scene(n_views=views, n_segs=segs, n_neighbors=nn, seed=0, topk=topk)
cfg = syn.default_triangulation_cfg(debug_mode=True)
O = ora.OracleTriangulator(cfg, faithful=False)
O.SetRanges(sc.ranges)
O.Init(sc.img_ids, sc.kvec, sc.qvec, sc.tvec, sc.seg_off, sc.segs)
for i in sc.img_ids:
    O.TriangulateImage(int(i), sc.matches_of(int(i), topk))
a = O.get_all_tris()
off, line, src_ = a["off"], np.ascontiguousarray(a["line"]), a["src"]
Ccount = int(off[-1])
print("nodes", len(off) - 1, "candidates", Ccount)
id2idx = {int(v): k for k, v in enumerate(sc.img_ids)}
img_idx = np.array([id2idx[int(v)] for v in src_[:, 0]], np.int32)
seg4 = np.ascontiguousarray(np.stack([sc.segs_of(int(ii))[int(l)] for ii, l in zip(img_idx, src_[:, 1])]).reshape(-1, 4))

cams = np.zeros(L.cam_bytes() * sc.n_images, np.uint8)
L.build_cams(C.c_int(sc.n_images), np.ascontiguousarray(sc.kvec, float).ctypes, np.ascontiguousarray(sc.qvec, float).ctypes,
             np.ascontiguousarray(sc.tvec, float).ctypes, cams.ctypes)


class L2(C.Structure):
    _fields_ = [(n, C.c_double) for n in "score_th th_angle th_overlap th_smartoverlap th_smartangle th_perp th_innerseg mult".split()] + \
               [(n, C.c_int) for n in "use_angle use_overlap use_smartangle use_perp use_innerseg pad_".split()]


class L3(C.Structure):
    _fields_ = [(n, C.c_double) for n in "score_th th_angle th_overlap th_smartoverlap th_smartangle th_perp th_innerseg th_scaleinv mult".split()] + \
               [(n, C.c_int) for n in "use_angle use_overlap use_smartangle use_perp use_innerseg use_scaleinv".split()]


class SC(C.Structure):
    _fields_ = [("l2", L2), ("l3", L3), ("cos_guard", C.c_double), ("fullscore_th", C.c_double), ("max_valid_conns", C.c_int), ("pad_", C.c_int)]


assert C.sizeof(SC) == L.score_cfg_bytes()
mult = lambda th: 1.0 / math.sqrt(-math.log(th) * 2.0)
l2c, l3c = cfg["linker2d_config"], cfg["linker3d_config"]
s = SC()
s.l2 = L2(l2c.get("score_th", 0.5), l2c.get("th_angle", 8.0), l2c.get("th_overlap", 0.1), l2c.get("th_smartoverlap", 0.2),
          l2c.get("th_smartangle", 1.0), l2c.get("th_perp", 5.0), l2c.get("th_innerseg", 5.0), mult(l2c.get("score_th", 0.5)),
          1, 1, 1, 1, 0, 0)
s.l3 = L3(l3c["score_th"], l3c["th_angle"], l3c["th_overlap"], l3c["th_smartoverlap"], l3c["th_smartangle"], l3c["th_perp"],
          l3c["th_innerseg"], l3c["th_scaleinv"], mult(l3c["score_th"]), 1, 0, 0, 0, 0, 1)
th = s.l3.th_angle * (1 + 1e-6) + 1e-6
s.cos_guard = math.cos(th * math.pi / 180)
g = s.l3.th_scaleinv * (1 + 1e-6)
out = np.zeros(32, np.int64)
dense_of = np.zeros(len(line), np.int32)
L.pair_stats2(C.c_longlong(len(off) - 1), off.ctypes, line.ctypes, seg4.ctypes, img_idx.ctypes, img_idx.ctypes, cams.ctypes,
              C.byref(s), C.c_double(g * g), out.ctypes, dense_of.ctypes)
names = ["dense (pass the sweep guards)", "die: 3D angle", "die: scale-inv distance", "die: 2D angle", "die: 2D overlap",
         "die: smart angle", "die: perpendicular distance", "score > 0"]
n = np.diff(off)
print("node sizes: mean %.1f  max %d  >64: %d nodes (%d candidates)  sum n^2 %.3g" %
      (n.mean(), n.max(), (n > 64).sum(), n[n > 64].sum(), float((n.astype(float) ** 2).sum())))
print("size histogram (<=8, <=16, <=32, <=64, <=128, >128):", [(int(((n > a) & (n <= b)).sum())) for a, b in
      [(-1, 8), (8, 16), (16, 32), (32, 64), (64, 128), (128, 10 ** 9)]])
for k, nm in enumerate(names):
    print("%-34s %10d  %5.1f %%" % (nm, out[k], 100.0 * out[k] / max(out[0], 1)))
print("score2d cross-check mismatches:", out[31])
# does the oracle's score agree?  (sum over images of the maxima is not reproduced here)

# ---- tile packing: current (64 consecutive candidates, window = all nodes touched) vs node-aligned classes ----
C_ = int(off[-1]); nt = (C_ + 63) // 64
node_of = np.repeat(np.arange(len(n)), n)
cm = np.zeros(nt, np.int64); wsz = np.zeros(nt, np.int64)
for t in range(nt):
    a_, b_ = node_of[t * 64], node_of[min(t * 64 + 63, C_ - 1)]
    cm[t] = n[a_:b_ + 1].max(); wsz[t] = off[b_ + 1] - off[a_]
print("current : tiles %d  sum cmax %d  mean window %.1f  (ideal sum n^2/64 = %d)" % (nt, cm.sum(), wsz.mean(), (n.astype(float) ** 2).sum() / 64))
kk = np.where(n > 0, np.minimum(16, 64 // np.maximum(n, 1)), 0)
tiles = 0; scm = 0
for k in range(1, 17):
    c = int((kk == k).sum())
    if c == 0:
        continue
    t_k = -(-c // k)
    tiles += t_k; scm += t_k * int(n[kk == k].max())
    print("  class k=%2d cap=%2d nodes=%6d tiles=%5d  lanes used %.0f %%" % (k, 64 // k, c, t_k, 100.0 * n[kk == k].sum() / (t_k * 64)))
print("classes : tiles %d  sum cmax %d" % (tiles, scm))

# ---- dense rounds per tile (64 consecutive candidates): full vs partial ----
pt = np.add.reduceat(dense_of[:C_], np.arange(0, C_, 64))
rounds = -(-pt // 64)
print("dense pairs per tile: mean %.1f median %d p90 %d max %d; rounds sum %d (ideal %d, %.0f %% lane use); tiles with 0 pairs %d" %
      (pt.mean(), np.median(pt), np.percentile(pt, 90), pt.max(), rounds.sum(), -(-pt.sum() // 64), 100.0 * pt.sum() / (64.0 * rounds.sum()), (pt == 0).sum()))
nsum = np.add.reduceat(np.repeat(n, n)[:C_], np.arange(0, C_, 64))
print("corr(pairs per tile, sum n) = %.3f" % np.corrcoef(pt, nsum)[0, 1])
np.save("/tmp/pairs_per_tile.npy", pt)
