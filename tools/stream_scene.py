"""A large scene streamed through the library path (limap_amd/stream.py; BASELINE configs[4]) on one GPU -- a thin caller:
`bench.py --stream` is the measured form, this prints the per-chunk records.

    python tools/stream_scene.py [--views 5000] [--segs 600] [--chunk 250]
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from limap_amd import stream, synthetic as syn  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--views", type=int, default=5000)
ap.add_argument("--segs", type=int, default=600)
ap.add_argument("--neighbors", type=int, default=20)
ap.add_argument("--chunk", type=int, default=250)
a = ap.parse_args()
sc = syn.make_scene(n_views=a.views, n_segs=a.segs, n_neighbors=a.neighbors, n_rooms=max(1, a.views // 100), seed=2)
cfg = dict(syn.default_triangulation_cfg(), add_halfpix=True)  # cfgs/triangulation/rome16k.yaml
st = stream.StreamedTriangulation(cfg, sc.img_ids, sc.kvec, sc.qvec, sc.tvec, sc.seg_off, sc.segs, sc.neighbors, sc.ranges,
                                  chunk_images=a.chunk)
for ch in st.my_chunks():
    print(json.dumps({k: (round(v, 3) if isinstance(v, float) else v) for k, v in st.run_chunk(ch, sc.matches_of).items()}))
A = st.finish()
print(json.dumps({"tracks": A.stats()["tracks"], "chunks": len(st.chunks)}))
