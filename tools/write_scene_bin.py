"""Writes a synthetic scene as the flat binary file examples/rccl_host.c reads (format: the comment at the top of that
file): `python tools/write_scene_bin.py out.bin [--views 100 --segs 500 --neighbors 20 --seed 0]`."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from limap_amd import synthetic as syn  # noqa: E402


def write_scene_bin(path, sc, topk=None):
    nb_off, nb_ids, m_off, rows = [0], [], [0], []
    for i in sc.img_ids:
        m = sc.matches_of(int(i), topk)
        for nb in sorted(m):  # ascending neighbour ids: what the reference's std::map iterates
            nb_ids.append(int(nb))
            rows.append(np.ascontiguousarray(m[nb], np.int32).reshape(-1, 2))
            m_off.append(m_off[-1] + len(rows[-1]))
        nb_off.append(len(nb_ids))
    rows = np.concatenate(rows, 0) if rows else np.zeros((0, 2), np.int32)
    with open(path, "wb") as f:
        np.array([sc.n_images, int(sc.seg_off[-1]), len(nb_ids), len(rows)], np.int64).tofile(f)
        np.asarray(sc.img_ids, np.int32).tofile(f)
        np.asarray(sc.seg_off, np.int64).tofile(f)
        for a in (sc.kvec, sc.qvec, sc.tvec, sc.segs):
            np.ascontiguousarray(a, np.float64).tofile(f)
        np.concatenate([np.asarray(sc.ranges[0], np.float64), np.asarray(sc.ranges[1], np.float64)]).tofile(f)
        np.asarray(nb_off, np.int64).tofile(f)
        np.asarray(nb_ids, np.int32).tofile(f)
        np.asarray(m_off, np.int64).tofile(f)
        rows.tofile(f)


def fnv_members(off, image_ids, line_ids):
    """the checksum rccl_host prints: FNV-1a 64 over off (int64) | image ids (int32) | line ids (int32)"""
    h = 1469598103934665603
    for a in (np.asarray(off, np.int64), np.asarray(image_ids, np.int32), np.asarray(line_ids, np.int32)):
        for b in a.tobytes():
            h = ((h ^ b) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("out")
    ap.add_argument("--views", type=int, default=100)
    ap.add_argument("--segs", type=int, default=500)
    ap.add_argument("--neighbors", type=int, default=20)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    write_scene_bin(a.out, syn.make_scene(n_views=a.views, n_segs=a.segs, n_neighbors=a.neighbors, seed=a.seed))
