"""Developer probe (CPU, numpy): how many match rows would a single-precision conservative pre-gate in NORMALISED image
coordinates reject, against the reference's own decision (oracle free functions would be slow: the double-precision
algebra of gen_gates is restated here in numpy).  Decides whether k_gates' pre-gate is worth building."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from limap_amd import synthetic as syn  # noqa: E402

sc = syn.make_scene(n_views=100, n_segs=500, n_neighbors=20, seed=0)
cfg = syn.default_triangulation_cfg()
TH_IOU, TH_ANG = cfg["IoU_threshold"], cfg["line_tri_angle_threshold"]
SIN_LO = np.sin(np.deg2rad(TH_ANG))


def cam(n):
    q = sc.qvec[n] / np.linalg.norm(sc.qvec[n])
    w, x, y, z = q
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                  [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                  [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    return sc.kvec[n], R, sc.tvec[n]


def seg_norm(n):
    k, R, t = cam(n)
    s = sc.segs_of(n)
    sn = np.stack([(s[:, 0] - k[2]) / k[0], (s[:, 1] - k[3]) / k[1], (s[:, 2] - k[2]) / k[0], (s[:, 3] - k[3]) / k[1]], 1)
    return sn, R, t


tot = dict(rows=0, ref_rej=0, pre_rej=0, pre_rej_wrong=0, not_well=0, ang_rej=0)
for n in range(0, 100, 10):
    img = int(sc.img_ids[n])
    s1, R1, t1 = seg_norm(n)
    m = sc.matches_of(img)
    for nb, rows in m.items():
        j = int(np.searchsorted(sc.img_ids, nb))
        s2, R2, t2 = seg_norm(j)
        Rr = R2 @ R1.T
        tr = t2 - Rr @ t1
        sk = np.array([[0, -tr[2], tr[1]], [tr[2], 0, -tr[0]], [-tr[1], tr[0], 0]])
        E = sk @ Rr
        a1, b2 = s1[rows[:, 0]], s2[rows[:, 1]]

        def gate(dt, eps_scale):
            E_ = E.astype(dt); A = a1.astype(dt); B = b2.astype(dt)
            one = np.ones(len(A), dt)
            ps = np.stack([A[:, 0], A[:, 1], one], 1); pe = np.stack([A[:, 2], A[:, 3], one], 1)
            s2h = np.stack([B[:, 0], B[:, 1], one], 1); e2h = np.stack([B[:, 2], B[:, 3], one], 1)
            lc = np.cross(s2h, e2h)
            v = B[:, 2:4] - B[:, 0:2]
            w1 = lc[:, 1] * v[:, 0] - lc[:, 0] * v[:, 1]; P = lc[:, 2] * v[:, 1]; Q = -lc[:, 2] * v[:, 0]
            sv = B[:, 0] * v[:, 0] + B[:, 1] * v[:, 1]; q2 = v[:, 0] ** 2 + v[:, 1] ** 2
            cs, errs, well = [], [], np.ones(len(A), bool)
            for p in (ps, pe):
                a = p @ E_.T
                Sa = np.abs(p) @ np.abs(E_).T
                t1_, t2_ = lc[:, 0] * a[:, 1], lc[:, 1] * a[:, 0]
                D = t1_ - t2_
                cerrD = np.abs(t1_) + np.abs(t2_) + np.abs(lc[:, 0]) * Sa[:, 1] + np.abs(lc[:, 1]) * Sa[:, 0]
                m0, m1, m2, m3 = a[:, 2] * w1, a[:, 0] * P, a[:, 1] * Q, D * sv
                numer = (m0 + m1 + m2) - m3
                cerrN = (np.abs(m0) + np.abs(m1) + np.abs(m2) + np.abs(m3) + np.abs(w1) * Sa[:, 2] + np.abs(P) * Sa[:, 0]
                         + np.abs(Q) * Sa[:, 1] + np.abs(sv) * cerrD)
                with np.errstate(all="ignore"):
                    c = numer / (D * q2)
                    err = eps_scale * (cerrN / np.abs(D * q2) + np.abs(c) * cerrD / np.abs(D) + np.abs(c))
                well &= np.abs(D) > 1e-2 * cerrD
                cs.append(c); errs.append(err)
            c1, c2 = np.minimum(cs[0], cs[1]), np.maximum(cs[0], cs[1])
            num = np.minimum(c2, 1) - np.maximum(c1, 0)
            den = np.maximum(c2, 1) - np.minimum(c1, 0)
            delta = num - dt(TH_IOU) * den
            margin = (1 + TH_IOU) * (errs[0] + errs[1]) + 1e-3 * (1 + np.abs(c1) + np.abs(c2))
            return delta, margin, well, num / den

        d64, _, _, iou = gate(np.float64, 0.0)
        # rays / plane normals (world)
        def rays(sn, R):
            r = np.stack([sn[:, 0], sn[:, 1], np.ones(len(sn))], 1); r /= np.linalg.norm(r, axis=1, keepdims=True)
            return r @ R  # R^T r
        rs, re = rays(a1[:, 0:2], R1), rays(a1[:, 2:4], R1)
        one = np.ones(len(b2))
        n2 = np.cross(np.stack([b2[:, 0], b2[:, 1], one], 1), np.stack([b2[:, 2], b2[:, 3], one], 1))
        n2 /= np.linalg.norm(n2, axis=1, keepdims=True)
        n2 = n2 @ R2
        as_, ae_ = np.abs((n2 * rs).sum(1)), np.abs((n2 * re).sum(1))
        ref_rej = (as_ < SIN_LO) | (ae_ < SIN_LO) | (iou < TH_IOU)
        d32, m32, w32, _ = gate(np.float32, 4e-6)
        ang32 = (as_.astype(np.float32) < np.float32(SIN_LO - 1e-6)) | (ae_.astype(np.float32) < np.float32(SIN_LO - 1e-6))
        pre_rej = ang32 | (w32 & (d32 < -m32))
        tot["rows"] += len(rows); tot["ref_rej"] += int(ref_rej.sum()); tot["pre_rej"] += int(pre_rej.sum())
        tot["pre_rej_wrong"] += int((pre_rej & ~ref_rej).sum()); tot["not_well"] += int((~w32).sum()); tot["ang_rej"] += int(ang32.sum())
print(tot)
print("reference rejects %.1f %%, pre-gate rejects %.1f %% (%.1f %% of what the reference rejects), wrongly: %d, not well-conditioned %.2f %%"
      % (100 * tot["ref_rej"] / tot["rows"], 100 * tot["pre_rej"] / tot["rows"], 100 * tot["pre_rej"] / max(tot["ref_rej"], 1),
         tot["pre_rej_wrong"], 100 * tot["not_well"] / tot["rows"]))
