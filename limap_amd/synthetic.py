"""Seeded synthetic line-triangulation workloads.

A scene is a row of box rooms (10 x 8 x 3 m each) holding GT 3D segments (70 % Manhattan on the
walls / floor / ceiling, 30 % random), observed by pinhole cameras with the Hypersim intrinsics
(f = 692.82, cx = 400, cy = 300, 800 x 600; runners/hypersim/Hypersim.py:71-106 of the
reference) moving on a closed smooth trajectory.  Every image holds exactly ``n_segs`` 2D
segments: noisy, partially overlapping observations of the visible GT segments padded with
random clutter, sorted by length like ``take_longest_k`` (line2d/base_detector.py:185-195).
Neighbours are the ``n_neighbors`` nearest views with optical-axis angle < 60 deg; matches are
the true correspondence plus random distractors up to ``topk`` rows per (line, neighbour), the
on-disk format of ``matches_{id}.npy`` (dict ng_img_id -> (K, 2) int32).

Everything is a pure function of (parameters, seed): ranks of a multi-GPU job regenerate the
same scene independently and keep only their own shard.
"""
from dataclasses import dataclass, field

import numpy as np

F_HYPERSIM = 692.8203230275509  # 400 / tan(30 deg)
W_IMG, H_IMG = 800, 600


@dataclass
class Scene:
    img_ids: np.ndarray  # (N,) int32, ascending
    kvec: np.ndarray  # (N,4) fx, fy, cx, cy
    qvec: np.ndarray  # (N,4) w,x,y,z world->cam
    tvec: np.ndarray  # (N,3)
    seg_off: np.ndarray  # (N+1,) int64
    segs: np.ndarray  # (sum M,4) x1,y1,x2,y2
    gt_ids: np.ndarray  # (sum M,) GT segment id or -1 for clutter
    gt_lines: np.ndarray  # (G,6)
    neighbors: dict  # img_id -> list[int] (nearest first)
    ranges: tuple  # (lo[3], hi[3])
    seed: int = 0
    topk: int = 10
    params: dict = field(default_factory=dict)

    @property
    def n_images(self):
        return len(self.img_ids)

    def segs_of(self, idx):
        return self.segs[self.seg_off[idx]:self.seg_off[idx + 1]]

    def all_2d_segs(self):
        return {int(i): self.segs_of(k) for k, i in enumerate(self.img_ids)}

    def matches_of(self, img_id, topk=None):
        return gen_matches(self, int(img_id), self.topk if topk is None else topk)

    def cam11(self, idx):
        return np.concatenate([self.kvec[idx], self.qvec[idx], self.tvec[idx]])


def translate_scene(scene, shift):
    """The same scene moved by `shift` in world coordinates (x_cam = R x + t  ->  t' = t - R shift): identical
    2D data, 3D results far from the origin -- exercises code that works on origin-relative coordinates."""
    import dataclasses
    shift = np.asarray(shift, float).reshape(3)
    tvec = np.array([scene.tvec[n] - quat_to_rot(scene.qvec[n]) @ shift for n in range(scene.n_images)])
    gt = scene.gt_lines.copy()
    gt[:, :3] += shift
    gt[:, 3:] += shift
    ranges = None if scene.ranges is None else (scene.ranges[0] + shift, scene.ranges[1] + shift)
    return dataclasses.replace(scene, tvec=tvec, gt_lines=gt, ranges=ranges)


def _rot_to_quat(R):
    """Rotation matrix -> (w,x,y,z), w >= 0."""
    tr = np.trace(R)
    if tr > 0:
        s = np.sqrt(tr + 1.0) * 2
        q = np.array([0.25 * s, (R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s])
    elif R[0, 0] > R[1, 1] and R[0, 0] > R[2, 2]:
        s = np.sqrt(1.0 + R[0, 0] - R[1, 1] - R[2, 2]) * 2
        q = np.array([(R[2, 1] - R[1, 2]) / s, 0.25 * s, (R[0, 1] + R[1, 0]) / s, (R[0, 2] + R[2, 0]) / s])
    elif R[1, 1] > R[2, 2]:
        s = np.sqrt(1.0 + R[1, 1] - R[0, 0] - R[2, 2]) * 2
        q = np.array([(R[0, 2] - R[2, 0]) / s, (R[0, 1] + R[1, 0]) / s, 0.25 * s, (R[1, 2] + R[2, 1]) / s])
    else:
        s = np.sqrt(1.0 + R[2, 2] - R[0, 0] - R[1, 1]) * 2
        q = np.array([(R[1, 0] - R[0, 1]) / s, (R[0, 2] + R[2, 0]) / s, (R[1, 2] + R[2, 1]) / s, 0.25 * s])
    if q[0] < 0:
        q = -q
    return q / np.linalg.norm(q)


def quat_to_rot(q):
    w, x, y, z = np.asarray(q, float) / np.linalg.norm(q)
    return np.array([
        [1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
        [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
        [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)],
    ])


def _gt_segments(rng, n_gt, size):
    """70 % Manhattan segments on the 6 faces, 30 % free segments inside the box."""
    sx, sy, sz = size
    out = np.zeros((n_gt, 6))
    n_plane = int(round(0.7 * n_gt))
    for g in range(n_gt):
        L = rng.uniform(0.3, 2.5)
        if g < n_plane:
            face = rng.integers(0, 6)
            axis = face // 2  # normal axis
            side = face % 2
            in_axes = [a for a in range(3) if a != axis]
            d_axis = in_axes[rng.integers(0, 2)]
            dims = np.array([sx, sy, sz])
            L = min(L, 0.9 * dims[d_axis])
            p = np.array([rng.uniform(0, sx), rng.uniform(0, sy), rng.uniform(0, sz)])
            p[axis] = 0.0 if side == 0 else dims[axis]
            p[d_axis] = rng.uniform(0, dims[d_axis] - L)
            q = p.copy()
            q[d_axis] += L
        else:
            p = np.array([rng.uniform(0.2, sx - 0.2), rng.uniform(0.2, sy - 0.2), rng.uniform(0.1, sz - 0.1)])
            d = rng.normal(size=3)
            d /= np.linalg.norm(d)
            q = np.clip(p + L * d, [0.05, 0.05, 0.05], [sx - 0.05, sy - 0.05, sz - 0.05])
        out[g, :3], out[g, 3:] = p, q
    return out


def _cameras(rng, n_views, size):
    sx, sy, sz = size
    cx, cy = sx / 2, sy / 2
    ax, ay = sx / 2 - 1.6, sy / 2 - 1.6
    s = 2 * np.pi * (np.arange(n_views) + rng.uniform(-0.2, 0.2, n_views)) / n_views
    # smooth closed trajectory: super-ellipse hugging the long room, slowly varying height
    ce, se = np.cos(s), np.sin(s)
    px = cx + ax * np.sign(ce) * np.abs(ce) ** 0.7
    py = cy + ay * np.sign(se) * np.abs(se) ** 0.7
    pz = 1.5 + 0.35 * np.sin(3 * s + 0.4) + rng.normal(0, 0.03, n_views)
    qv = np.zeros((n_views, 4))
    tv = np.zeros((n_views, 3))
    centers = np.stack([px, py, pz], 1)
    for i in range(n_views):
        tang = np.array([-ax * se[i], ay * ce[i], 0.0])
        tang /= np.linalg.norm(tang) + 1e-12
        inward = np.array([cx - px[i], cy - py[i], 0.0])
        inward /= np.linalg.norm(inward) + 1e-12
        yaw_j = np.deg2rad(rng.normal(0, 12.0))
        fwd = np.cos(np.deg2rad(40)) * tang + np.sin(np.deg2rad(40)) * inward
        c, s_ = np.cos(yaw_j), np.sin(yaw_j)
        fwd = np.array([c * fwd[0] - s_ * fwd[1], s_ * fwd[0] + c * fwd[1], np.tan(np.deg2rad(rng.normal(0, 6.0)))])
        fwd /= np.linalg.norm(fwd)
        up = np.array([0, 0, 1.0])
        right = np.cross(fwd, up)
        right /= np.linalg.norm(right)
        down = np.cross(fwd, right)
        roll = np.deg2rad(rng.normal(0, 2.0))
        r2 = np.cos(roll) * right + np.sin(roll) * down
        d2 = np.cross(fwd, r2)
        R = np.stack([r2, d2, fwd], 0)  # world -> cam
        qv[i] = _rot_to_quat(R)
        tv[i] = -quat_to_rot(qv[i]) @ centers[i]
    return qv, tv, centers


def _clip_segments(p, q, w, h):
    """Liang-Barsky clip of 2D segments (n,2) to [0,w]x[0,h]; returns clipped endpoints + mask."""
    d = q - p
    t0 = np.zeros(len(p))
    t1 = np.ones(len(p))
    ok = np.ones(len(p), bool)
    for pk, qk in ((-d[:, 0], p[:, 0]), (d[:, 0], w - p[:, 0]), (-d[:, 1], p[:, 1]), (d[:, 1], h - p[:, 1])):
        par = pk == 0
        ok &= ~(par & (qk < 0))
        with np.errstate(divide="ignore", invalid="ignore"):
            r = np.where(par, 0.0, qk / np.where(par, 1.0, pk))
        ent = (pk < 0) & ~par
        ext = (pk > 0) & ~par
        t0 = np.where(ent, np.maximum(t0, r), t0)
        t1 = np.where(ext, np.minimum(t1, r), t1)
    ok &= t0 < t1
    return p + d * t0[:, None], p + d * t1[:, None], ok


def _observe(rng, gt, K4, q, t, n_segs):
    """Project GT segments into one view -> (n_segs,4) segments + GT ids (-1 = clutter)."""
    R = quat_to_rot(q)
    fx, fy, cx, cy = K4
    P = gt[:, :3] @ R.T + t
    Q = gt[:, 3:] @ R.T + t
    znear = 0.2
    # clip against the near plane
    dz = Q[:, 2] - P[:, 2]
    both_behind = (P[:, 2] < znear) & (Q[:, 2] < znear)
    with np.errstate(divide="ignore", invalid="ignore"):
        tc = np.where(dz != 0, (znear - P[:, 2]) / np.where(dz != 0, dz, 1.0), 0.0)
    Pn = np.where((P[:, 2] < znear)[:, None], P + (Q - P) * tc[:, None], P)
    Qn = np.where((Q[:, 2] < znear)[:, None], P + (Q - P) * tc[:, None], Q)
    zp = np.maximum(Pn[:, 2], 1e-6)
    zq = np.maximum(Qn[:, 2], 1e-6)
    p2 = np.stack([fx * Pn[:, 0] / zp + cx, fy * Pn[:, 1] / zp + cy], 1)
    q2 = np.stack([fx * Qn[:, 0] / zq + cx, fy * Qn[:, 1] / zq + cy], 1)
    a, b, ok = _clip_segments(p2, q2, W_IMG, H_IMG)
    ok &= ~both_behind
    length = np.linalg.norm(b - a, axis=1)
    ok &= length >= 15.0
    ids = np.nonzero(ok)[0]
    a, b, length = a[ids], b[ids], length[ids]
    d = (b - a) / length[:, None]
    n = np.stack([-d[:, 1], d[:, 0]], 1)
    a = a + d * (rng.uniform(-0.15, 0.15, len(ids)) * length)[:, None] + n * rng.normal(0, 0.5, len(ids))[:, None]
    b = b + d * (rng.uniform(-0.15, 0.15, len(ids)) * length)[:, None] + n * rng.normal(0, 0.5, len(ids))[:, None]
    # random endpoint order, like a detector
    flip = rng.random(len(ids)) < 0.5
    a2 = np.where(flip[:, None], b, a)
    b2 = np.where(flip[:, None], a, b)
    segs = np.concatenate([a2, b2], 1)
    gids = ids.astype(np.int64)
    if len(segs) > n_segs:
        keep = np.argsort(-np.linalg.norm(segs[:, 2:] - segs[:, :2], axis=1), kind="stable")[:n_segs]
        segs, gids = segs[keep], gids[keep]
    n_cl = n_segs - len(segs)
    if n_cl > 0:
        c = np.stack([rng.uniform(0, W_IMG, n_cl), rng.uniform(0, H_IMG, n_cl)], 1)
        th = rng.uniform(0, np.pi, n_cl)
        L = rng.uniform(20, 150, n_cl)
        dd = np.stack([np.cos(th), np.sin(th)], 1) * (L / 2)[:, None]
        cl = np.concatenate([np.clip(c - dd, 0, [W_IMG, H_IMG]), np.clip(c + dd, 0, [W_IMG, H_IMG])], 1)
        segs = np.concatenate([segs, cl], 0)
        gids = np.concatenate([gids, -np.ones(n_cl, np.int64)])
    order = np.argsort(-np.linalg.norm(segs[:, 2:] - segs[:, :2], axis=1), kind="stable")
    return segs[order], gids[order]


def make_scene(n_views=100, n_segs=500, n_neighbors=20, n_rooms=1, n_gt=None, seed=0, topk=10,
               img_id_offset=0):
    """Build the cameras, 2D segments, neighbours and ranges of a synthetic scene."""
    rng = np.random.default_rng(seed)
    size = (10.0 * n_rooms, 8.0, 3.0)
    if n_gt is None:
        n_gt = 600 * n_rooms
    gt = _gt_segments(rng, n_gt, size)
    qv, tv, centers = _cameras(rng, n_views, size)
    K4 = np.array([F_HYPERSIM, F_HYPERSIM, 400.0, 300.0])
    # K passes through float32 in the Hypersim loader (runners/hypersim/loader.py:39,43)
    K4 = K4.astype(np.float32).astype(np.float64)
    kv = np.tile(K4, (n_views, 1))
    segs_list, gid_list = [], []
    for i in range(n_views):
        s, g = _observe(np.random.default_rng([seed, 1000 + i]), gt, K4, qv[i], tv[i], n_segs)
        segs_list.append(s)
        gid_list.append(g)
    seg_off = np.zeros(n_views + 1, np.int64)
    seg_off[1:] = np.cumsum([len(s) for s in segs_list])
    img_ids = (np.arange(n_views) + img_id_offset).astype(np.int32)
    # neighbours: nearest centres among views with optical-axis angle < 60 deg
    axes = np.stack([quat_to_rot(q)[2] for q in qv], 0)
    neighbors = {}
    for i in range(n_views):
        dist = np.linalg.norm(centers - centers[i], axis=1)
        ang_ok = axes @ axes[i] > np.cos(np.deg2rad(60.0))
        dist[~ang_ok] = np.inf
        dist[i] = np.inf
        order = np.argsort(dist, kind="stable")
        order = order[np.isfinite(dist[order])][:n_neighbors]
        neighbors[int(img_ids[i])] = [int(img_ids[j]) for j in order]
    c = np.array(size) / 2
    half = np.array(size) / 2
    ranges = (c - 1.25 * half, c + 1.25 * half)
    return Scene(img_ids=img_ids, kvec=kv, qvec=qv, tvec=tv, seg_off=seg_off,
                 segs=np.concatenate(segs_list, 0), gt_ids=np.concatenate(gid_list, 0), gt_lines=gt,
                 neighbors=neighbors, ranges=ranges, seed=seed, topk=topk,
                 params=dict(n_views=n_views, n_segs=n_segs, n_neighbors=n_neighbors, n_rooms=n_rooms,
                             n_gt=n_gt, seed=seed, topk=topk))


def make_vp_results(scene, seed=0, drop=0.15, wrong=0.05):
    """Synthetic vanishing-point detections (the content of limap.vplib.VPResult per image): the three
    Manhattan directions of the box as vanishing points vp = K R e_axis, every segment of an
    axis-parallel GT line labelled with its axis -- except `drop` of them unlabelled (-1) and `wrong`
    of them given another axis; free segments mostly unlabelled.  Returns dict img_id -> (labels, vps)."""
    out = {}
    gt_dir = scene.gt_lines[:, 3:] - scene.gt_lines[:, :3]
    gt_dir /= np.maximum(np.linalg.norm(gt_dir, axis=1, keepdims=True), 1e-300)
    axis_of = np.full(len(gt_dir), -1)
    for a in range(3):
        axis_of[np.abs(gt_dir[:, a]) > 1.0 - 1e-9] = a
    for n, img_id in enumerate(scene.img_ids):
        rng = np.random.default_rng([scene.seed, seed, 555, int(img_id)])
        K = np.array([[scene.kvec[n, 0], 0, scene.kvec[n, 2]], [0, scene.kvec[n, 1], scene.kvec[n, 3]], [0, 0, 1.0]])
        R = quat_to_rot(scene.qvec[n])
        vps = (K @ R).T.copy()            # row a = K R e_a
        g = scene.gt_ids[scene.seg_off[n]:scene.seg_off[n + 1]]
        lab = np.where(g >= 0, axis_of[np.maximum(g, 0)], -1).astype(np.int32)
        u = rng.random(len(lab))
        lab[(lab >= 0) & (u < drop)] = -1
        flip = (lab >= 0) & (u > 1.0 - wrong)
        lab[flip] = (lab[flip] + 1 + rng.integers(0, 2, size=int(flip.sum()))) % 3
        stray = (lab < 0) & (rng.random(len(lab)) < 0.05)
        lab[stray] = rng.integers(0, 3, size=int(stray.sum()))
        out[int(img_id)] = (lab, vps)
    return out


def make_bipartites(scene, seed=0, pts_per_line=3, noise_px=0.3, noise_3d=0.002):
    """Synthetic point-line bipartites (the content of limap.structures.PL_Bipartite2d per image) and SfM
    points: every GT segment carries `pts_per_line` 3D points (id = pts_per_line * g + k, slightly off the
    line); an image's segment that observes GT segment g is connected to the projections of g's points
    that fall in front of the camera and inside the image (pixel noise `noise_px`).
    Returns (bipartites: dict img_id -> dict(point_ids, xy, point3D_ids, line_points), sfm_points: dict)."""
    rng0 = np.random.default_rng([scene.seed, seed, 909])
    G = len(scene.gt_lines)
    ts = (np.arange(pts_per_line) + 0.5) / pts_per_line
    P3 = {}
    for g in range(G):
        a, b = scene.gt_lines[g, :3], scene.gt_lines[g, 3:]
        for k, t in enumerate(ts):
            P3[pts_per_line * g + k] = a + t * (b - a) + rng0.normal(0, noise_3d, 3)
    bpts = {}
    for n, img_id in enumerate(scene.img_ids):
        rng = np.random.default_rng([scene.seed, seed, 910, int(img_id)])
        R = quat_to_rot(scene.qvec[n])
        fx, fy, cx, cy = scene.kvec[n]
        gids = scene.gt_ids[scene.seg_off[n]:scene.seg_off[n + 1]]
        ids, xy, p3d, line_points = [], [], [], []
        cache = {}
        for g in gids:
            mine = []
            if g >= 0:
                for k in range(pts_per_line):
                    pid3 = pts_per_line * int(g) + k
                    if pid3 not in cache:
                        Xc = R @ P3[pid3] + scene.tvec[n]
                        if Xc[2] <= 0.2:
                            cache[pid3] = -1
                        else:
                            u = np.array([fx * Xc[0] / Xc[2] + cx, fy * Xc[1] / Xc[2] + cy]) + rng.normal(0, noise_px, 2)
                            if 0 <= u[0] <= W_IMG and 0 <= u[1] <= H_IMG:
                                cache[pid3] = len(ids)
                                ids.append(len(ids)); xy.append(u); p3d.append(pid3)
                            else:
                                cache[pid3] = -1
                    if cache[pid3] >= 0:
                        mine.append(cache[pid3])
            line_points.append(mine)
        bpts[int(img_id)] = dict(point_ids=np.array(ids, np.int32), xy=np.array(xy, float).reshape(-1, 2),
                                 point3D_ids=np.array(p3d, np.int32), line_points=line_points)
    return bpts, {int(k): v for k, v in P3.items()}


def gen_matches(scene, img_id, topk=10):
    """matches_{img_id}: dict ng_img_id -> (K,2) int32, rows grouped by line id, <= topk rows per
    line: the true GT correspondence (when the GT segment is visible in the neighbour) at a random
    rank, the rest random distractors."""
    idx = int(np.searchsorted(scene.img_ids, img_id))
    rng = np.random.default_rng([scene.seed, 77, int(img_id)])
    g1 = scene.gt_ids[scene.seg_off[idx]:scene.seg_off[idx + 1]]
    M = len(g1)
    out = {}
    for nb in scene.neighbors[int(img_id)]:
        j = int(np.searchsorted(scene.img_ids, nb))
        g2 = scene.gt_ids[scene.seg_off[j]:scene.seg_off[j + 1]]
        M2 = len(g2)
        if M == 0 or M2 == 0 or topk == 0:
            out[int(nb)] = np.zeros((0, 2), np.int32)
            continue
        k = min(topk, M2)
        cand = rng.integers(0, M2, size=(M, k))
        lut = -np.ones(int(max(g1.max(), g2.max())) + 2, np.int64)
        valid2 = g2 >= 0
        lut[g2[valid2]] = np.nonzero(valid2)[0]
        true_j = np.where(g1 >= 0, lut[np.maximum(g1, 0)], -1)
        has = true_j >= 0
        col = rng.integers(0, k, size=M)
        cand[has, col[has]] = true_j[has]
        rows = np.stack([np.repeat(np.arange(M), k), cand.reshape(-1)], 1)
        out[int(nb)] = rows.astype(np.int32)
    return out


def default_triangulation_cfg(var2d=2.0, **over):
    """cfg["triangulation"] of cfgs/triangulation/default.yaml:70-100 with var2d resolved for LSD
    (line_triangulation.py:39-40)."""
    cfg = dict(
        use_exhaustive_matcher=False, use_endpoints_triangulation=False, add_halfpix=False,
        min_length_2d=0.0, var2d=var2d, line_tri_angle_threshold=1.0, IoU_threshold=0.1,
        sensitivity_threshold=70.0, fullscore_th=1.0, max_valid_conns=1000, min_num_outer_edges=0,
        merging_strategy="greedy", num_outliers_aggregator=2, debug_mode=False,
        linker2d_config=dict(score_th=0.5, th_angle=5.0, th_perp=2.0, th_overlap=0.05),
        linker3d_config=dict(score_th=0.5, th_angle=10.0, th_overlap=0.05, th_smartoverlap=0.1,
                             th_smartangle=2.0, th_perp=1.0, th_innerseg=1.0, th_scaleinv=0.015),
        use_vp=False,
    )
    cfg.update(over)
    return cfg
