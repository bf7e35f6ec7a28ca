"""Developer experiment: two independent batches in flight (two contexts, two streams) against one.
Each context holds the bench workload (100 views x 500 segs, matched topk 10); a step = one lt_run_device_async.
usage (GPU box): python tools/two_contexts.py [n_contexts] [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from limap_amd import synthetic as syn, _capi

n_ctx = int(sys.argv[1]) if len(sys.argv) > 1 else 2
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
sc = syn.make_scene(n_views=100, n_segs=500, n_neighbors=20, seed=0)
cfg = syn.default_triangulation_cfg()
ctxs = []
for c in range(n_ctx):
    ctx = _capi.Context(cfg_dict=cfg, device=0)
    ctx.set_ranges(*sc.ranges)
    ctx.init(sc.img_ids, sc.kvec, sc.qvec, sc.tvec, sc.seg_off, sc.segs)
    for i in sc.img_ids:
        m = sc.matches_of(int(i))
        nb = list(m.keys())
        off = np.zeros(len(nb) + 1, np.int64)
        off[1:] = np.cumsum([len(m[k]) for k in nb])
        ctx.triangulate_image(int(i), nb, off, np.concatenate([m[k] for k in nb], 0))
    ctx.upload()
    ctxs.append(ctx)
for _ in range(3):
    for ctx in ctxs:
        ctx.run_device(wait=False)
for ctx in ctxs:
    ctx.sync()
torch.cuda.synchronize()
t0 = time.perf_counter()
for s in range(steps):
    ctxs[s % n_ctx].run_device(wait=False)
for ctx in ctxs:
    ctx.sync()
torch.cuda.synchronize()
el = time.perf_counter() - t0
cand = ctxs[0].stats()["candidates"] if False else 579235
print(f"{n_ctx} context(s): {1e3 * el / steps:.4f} ms per step, {cand * steps / el:.4g} candidates/s")
