"""Developer tool (GPU box): the end-to-end call sequence through ctypes on two builds of the library, alternating, same box.
   python tools/ab_e2e.py [--config3] label:lib.so label:lib.so ...   (one subprocess per entry and round)"""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(config3):
    sys.path.insert(0, ROOT)
    import gc
    import numpy as np
    from limap_amd import synthetic as syn, triangulation as tri
    tri._pb = None  # the pybind shim links the in-tree library
    if config3:
        sc = syn.make_scene(n_views=1000, n_segs=1000, n_neighbors=20, n_rooms=4, n_gt=3000, seed=1, topk=10)
    else:
        sc = syn.make_scene(n_views=100, n_segs=500, n_neighbors=20, seed=0)
    cfg = syn.default_triangulation_cfg()
    matches = {int(i): sc.matches_of(int(i)) for i in sc.img_ids}
    segs_list = [sc.segs_of(j) for j in range(sc.n_images)]
    res = []
    for rep in range(6):
        gc.collect(); gc.disable()
        t0 = time.perf_counter()
        T = tri.GlobalLineTriangulator(cfg)
        T.SetRanges(sc.ranges)
        T.InitArrays(sc.img_ids, sc.kvec, sc.qvec, sc.tvec, segs_list)
        t1 = time.perf_counter()
        for i in sc.img_ids:
            T.TriangulateImage(int(i), matches[int(i)])
        t2 = time.perf_counter()
        T.context().compute_tracks()
        t3 = time.perf_counter()
        gc.enable()
        res.append((t1 - t0, t2 - t1, t3 - t2, t3 - t0))
        del T
    r = 1e3 * np.median(np.array(res[2:]), axis=0)
    print("RESULT ctor_init %.2f buffer %.2f compute_tracks %.2f total %.2f" % tuple(r), flush=True)


if "--child" in sys.argv:
    child("--config3" in sys.argv)
else:
    entries = [a for a in sys.argv[1:] if ":" in a]
    for rnd in range(2):
        for e in entries:
            label, lib = e.split(":")
            env = dict(os.environ, LIMAP_AMD_LIB=os.path.join(ROOT, lib))
            p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"] + (["--config3"] if "--config3" in sys.argv else []),
                               env=env, capture_output=True, text=True, timeout=600)
            line = [l for l in p.stdout.splitlines() if l.startswith("RESULT")]
            print(f"{label:8s}", line[-1][7:] if line else "FAILED " + p.stderr[-300:], flush=True)
