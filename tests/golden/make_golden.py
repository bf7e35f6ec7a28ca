#!/usr/bin/env python
"""Generates tests/golden/*.npz: seeded inputs + the CPU oracle's outputs for them.

The reference's own tests hold no golden vectors for this path and the reference cannot be imported
here, so these fixtures pin the ORACLE (regression / cross-compiler stability) and
give the GPU tests a fixed target that does not depend on the generator's RNG stream.
Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from limap_amd import synthetic as syn  # noqa: E402
from oracle import oracle as ora  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def make(name, mode, seed, n_views, n_segs, nn, topk, cfg_over):
    sc = syn.make_scene(n_views=n_views, n_segs=n_segs, n_neighbors=nn, seed=seed, topk=topk)
    cfg = syn.default_triangulation_cfg(debug_mode=True, **cfg_over)
    O = ora.OracleTriangulator(cfg, faithful=True)
    O.SetRanges(sc.ranges)
    O.Init(sc.img_ids, sc.kvec, sc.qvec, sc.tvec, sc.seg_off, sc.segs)
    data = dict(img_ids=sc.img_ids, kvec=sc.kvec, qvec=sc.qvec, tvec=sc.tvec, seg_off=sc.seg_off, segs=sc.segs,
                ranges=np.stack(sc.ranges), mode=np.array(mode), cfg_over=np.array(repr(cfg_over)))
    if cfg_over.get("use_vp"):  # vplib.VPResult content per image: labels of all segments, 3 VPs per image
        vps = syn.make_vp_results(sc, seed=seed)
        O.InitVPResults(vps)
        data.update(vp_labels=np.concatenate([vps[int(i)][0] for i in sc.img_ids]).astype(np.int32),
                    vp_vps=np.stack([vps[int(i)][1] for i in sc.img_ids]))
    nb_flat, nb_off = [], [0]
    m_img, m_nb, m_off, m_rows = [], [], [0], []
    for i in sc.img_ids:
        nbs = sc.neighbors[int(i)]
        nb_flat += nbs
        nb_off.append(len(nb_flat))
        if mode == "matched":
            m = sc.matches_of(int(i), topk)
            for k in m:
                m_img.append(int(i)); m_nb.append(int(k)); m_rows.append(m[k]); m_off.append(m_off[-1] + len(m[k]))
            O.TriangulateImage(int(i), m)
        else:
            O.TriangulateImageExhaustiveMatch(int(i), nbs)
    data.update(nb_flat=np.array(nb_flat, np.int32), nb_off=np.array(nb_off, np.int64))
    if mode == "matched":
        data.update(m_img=np.array(m_img, np.int32), m_nb=np.array(m_nb, np.int32), m_off=np.array(m_off, np.int64),
                    m_rows=np.concatenate(m_rows, 0).astype(np.int32))
    b = O.get_best()
    eoff, edges = O.get_valid_edges()
    # store the valid edges sorted per node (their order is not observable)
    edges_sorted = np.concatenate([np.array(sorted(map(tuple, edges[eoff[g]:eoff[g + 1]].tolist())), np.int32).reshape(-1, 2)
                                   for g in range(len(eoff) - 1)], 0) if len(edges) else edges
    t = O.ComputeLineTracks()
    data.update(n_tris=O.get_num_tris(), best_line=b["line"], best_score=b["score"], best_src=b["src"],
                has_best=b["has_best"], edge_off=eoff, edges=edges_sorted, track_line=t["line"], track_off=t["off"],
                track_img=t["image_ids"], track_lid=t["line_ids"], track_node=t["node_ids"], track_score=t["scores"])
    st = O.stats()
    data["stats"] = np.array([st[k] for k in ("connections", "candidates", "pairs", "valid_edges", "graph_nodes",
                                              "graph_edges", "tracks")], np.int64)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **data)
    print(name, st)


if __name__ == "__main__":
    make("matched_s11", "matched", 11, 16, 100, 8, 6, {})
    make("exhaustive_s12", "exhaustive", 12, 14, 60, 8, 0, {})
    make("matched_outer2_halfpix_s13", "matched", 13, 14, 80, 8, 6, dict(add_halfpix=True, min_num_outer_edges=2))
    make("matched_endpoints_s14", "matched", 14, 10, 60, 6, 5, dict(use_endpoints_triangulation=True))
    make("matched_vp_s15", "matched", 15, 12, 70, 6, 5, dict(use_vp=True))
