"""SHA-256 digests of a triangulator's results, the same on the oracle's and on the product's side, so that runs too
long for the oracle on the GPU box (whole-scene exhaustive 100 x 500: minutes; config 3 with every image) are compared
through committed digests: tests/golden/digests.json, written by tests/golden/make_digests.py from the ORACLE in the build
container, checked against the HIP backend by tests/test_gpu_digests.py.

Only bit-exact quantities are hashed: which candidate is the best one of every node (source image / line), the best
candidate's geometry, the valid-edge SETS (the order inside a node has no observable effect: it feeds a std::set), the
track members (image ids, line ids, node ids, sizes) and the track lines.  Scores differ by libm (1e-12) and stay out."""
import hashlib

import numpy as np


def _h(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        a = np.ascontiguousarray(a)
        h.update(str(a.dtype).encode()); h.update(str(a.shape).encode()); h.update(a.tobytes())
    return h.hexdigest()


def result_digests(best, valid_edges, tracks, stats):
    """best: dict(has_best, src, line, score); valid_edges: (off, edges[E, 2]); tracks: dict(off, image_ids, line_ids,
    node_ids, line, ...); stats: dict."""
    off, edges = valid_edges
    off = np.asarray(off, np.int64)
    edges = np.asarray(edges, np.int64).reshape(-1, 2)
    # canonical order inside every node: by (image, line)
    node = np.repeat(np.arange(len(off) - 1, dtype=np.int64), np.diff(off))
    order = np.lexsort((edges[:, 1], edges[:, 0], node)) if len(edges) else np.zeros(0, np.int64)
    return {
        "best_src": _h(np.asarray(best["has_best"], np.uint8), np.asarray(best["src"], np.int64)),
        "best_line": _h(np.asarray(best["line"], np.float64)),
        "valid_edges": _h(off, edges[order]),
        "track_members": _h(np.asarray(tracks["off"], np.int64), np.asarray(tracks["image_ids"], np.int64),
                            np.asarray(tracks["line_ids"], np.int64), np.asarray(tracks["node_ids"], np.int64)),
        "track_lines": _h(np.asarray(tracks["line"], np.float64)[:, :6]),
        "counts": {k: int(stats[k]) for k in ("connections", "candidates", "valid_edges", "graph_nodes", "graph_edges", "tracks")},
    }


CASES = {
    # BASELINE configs[1] in the reference's exhaustive mode, EVERY image (5e8 connections, 2.2e7 candidates)
    "config2_exhaustive_all": dict(scene=dict(n_views=100, n_segs=500, n_neighbors=20, seed=0), exhaustive=True),
    # BASELINE configs[2]: 1000 views x 1000 segs over 4 rooms, matched top-10, EVERY image (1e8 connections)
    "config3_matched_all": dict(scene=dict(n_views=1000, n_segs=1000, n_neighbors=20, n_rooms=4, n_gt=3000, seed=1),
                                exhaustive=False),
    # a fifth of BASELINE configs[4]'s stand-in (5000 x 600 streamed): 1000 views x 600 segs over 10 rooms, matched top-10,
    # rome16k.yaml's add_halfpix -- what tests/test_gpu_stream.py streams in chunks through limap_amd.stream
    "stream_1000x600": dict(scene=dict(n_views=1000, n_segs=600, n_neighbors=20, n_rooms=10, seed=2), exhaustive=False,
                            cfg=dict(add_halfpix=True)),
}
