"""limap_amd -- MI355X-native backend for the line-triangulation hot path of cvg/limap.

    from limap_amd import triangulation
    tri = triangulation.GlobalLineTriangulator(cfg["triangulation"])   # same API as limap.triangulation

The HIP extension (limap_amd/liblimap_amd.so) must be built (``limap_amd.build.build_extension()``)
and a HIP device must be visible; there is no CPU fallback.
"""
from . import base, synthetic  # noqa: F401

__all__ = ["base", "synthetic", "triangulation", "build", "warmup"]


def warmup(n_views=100, n_segs=500, n_neighbors=20, topk=10, device=None):
    """Pay the one-time costs of a process up front instead of inside its first scene.  A synthetic scene of the EXPECTED
    shape (default: 100 views x 500 segments, 20 neighbours, top-10 matches = BASELINE config 2) goes once through the
    whole call sequence -- HIP runtime and code objects loaded (~0.2 s of a fresh process), every kernel of the pipeline,
    of the tail and of the post-triangulation chain launched, the host thread team started, and the stream / event set,
    the page-locked staging and the device blocks of that size created and handed back to the library's process-wide
    caches -- plus a toy scene through the exhaustive mode for its kernels.  The first real scene of about that size then
    runs at the warm time (tools/cold_probe.py: 4 ms instead of 260 ms in a fresh process; bench.py reports
    ``e2e_after_warmup_ms`` beside ``e2e_cold_ms``).  A service calls this once at start-up.  Returns the seconds it took."""
    import time
    from . import merging, synthetic as syn, triangulation as tri
    t0 = time.perf_counter()
    cfg = syn.default_triangulation_cfg()
    kw = {} if device is None else {"device": device}
    jobs = [(syn.make_scene(n_views=n_views, n_segs=n_segs, n_neighbors=min(n_neighbors, max(n_views - 1, 1)), topk=topk,
                            seed=12345), False),
            (syn.make_scene(n_views=6, n_segs=40, n_neighbors=3, seed=12345), True)]
    for sc, exhaustive in jobs:
        T = tri.GlobalLineTriangulator(cfg, **kw)
        T.SetRanges(sc.ranges)
        T.InitArrays(sc.img_ids, sc.kvec, sc.qvec, sc.tvec, [sc.segs_of(j) for j in range(sc.n_images)])
        for i in sc.img_ids:
            if exhaustive:
                T.TriangulateImageExhaustiveMatch(int(i), sc.neighbors[int(i)])
            else:
                T.TriangulateImage(int(i), sc.matches_of(int(i), topk))
        T.ComputeLineTracks()
        if not exhaustive:  # the post-triangulation chain of the runner (remerge runs k_track_connect)
            ts = merging.TrackSet.from_triangulator(T)
            ts.filter_by_reprojection(8.0, 5.0).remerge(dict(
                score_th=0.5, th_angle=5.0, th_overlap=0.001, th_smartoverlap=0.1, th_smartangle=1.0, th_perp=1.0,
                th_innerseg=1.0))  # cfgs/triangulation/default.yaml:102-115
            del ts
        del T
    return time.perf_counter() - t0


def __getattr__(name):  # lazy: importing the package must not require the GPU library
    if name in ("triangulation", "build", "_capi", "dist", "io"):
        import importlib
        return importlib.import_module("." + name, __name__)
    raise AttributeError(name)
