"""Developer tool (GPU box): lean A/B of library builds / environment switches on the bench scene, without torch.
   python tools/ab_score.py [--mode matched|exhaustive] [--runs N] label:lib.so[:K=V,K=V...] ...
One subprocess per entry (LIMAP_AMD_LIB + environment), each prints the median per-kernel times of N device runs and a
SHA-1 over every candidate's support score, the per-node best candidate and the valid edges -- entries whose hashes
agree produced bit-identical results."""
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(mode, runs):
    sys.path.insert(0, ROOT)
    import numpy as np
    from limap_amd import synthetic as syn, triangulation as tri
    ex = mode == "exhaustive"
    tri._pb = None  # the pybind shim links the in-tree library; everything here goes through ctypes into LIMAP_AMD_LIB
    if os.environ.get("AB_CONFIG3"):  # BASELINE config 3 (bench.py --config3)
        sc = syn.make_scene(n_views=1000, n_segs=1000, n_neighbors=20, n_rooms=4, n_gt=3000, seed=1, topk=10)
    else:
        sc = syn.make_scene(n_views=100, n_segs=500, n_neighbors=20, seed=0)
    T = tri.GlobalLineTriangulator(syn.default_triangulation_cfg(debug_mode=True))
    T.SetRanges(sc.ranges)
    T.InitArrays(sc.img_ids, sc.kvec, sc.qvec, sc.tvec, [sc.segs_of(j) for j in range(sc.n_images)])
    for i in sc.img_ids:
        if ex:
            T.TriangulateImageExhaustiveMatch(int(i), sc.neighbors[int(i)])
        else:
            T.TriangulateImage(int(i), sc.matches_of(int(i)))
    ctx = T.context()
    ctx.upload()
    keys = ("run", "k_gates", "k_tri_rows", "compact", "k_score3", "select", "gen", "score")
    rows = []
    for r in range(runs + 2):
        ctx.run_device()
        if r >= 2:
            t = ctx.timers()
            rows.append([t[k] for k in keys])
    med = np.median(np.array(rows), axis=0)
    ctx.download()
    h = hashlib.sha1()
    if not ex:
        allt = ctx.get_all_tris()
        h.update(allt["off"].tobytes()); h.update(allt["score"].tobytes()); h.update(allt["src"].tobytes())
    best = ctx.get_best()
    h.update(best["score"].tobytes()); h.update(best["line"].tobytes()); h.update(best["src"].tobytes())
    off, edges = ctx.get_valid_edges()
    h.update(off.tobytes()); h.update(edges.tobytes())
    out = {k: round(float(v), 4) for k, v in zip(keys, med)}
    out["pairs_eval"] = int(ctx.timers()["pairs_eval"])
    out["sha1"] = h.hexdigest()[:12]
    print("RESULT " + json.dumps(out), flush=True)


def main():
    args = sys.argv[1:]
    mode, runs = "matched", 15
    if "--child" in args:
        return child(args[args.index("--child") + 1], int(args[args.index("--child") + 2]))
    entries = []
    it = iter(args)
    for a in it:
        if a == "--mode":
            mode = next(it)
        elif a == "--runs":
            runs = int(next(it))
        else:
            entries.append(a)
    for e in entries:
        parts = e.split(":")
        label, lib = parts[0], parts[1]
        env = dict(os.environ)
        env["LIMAP_AMD_LIB"] = os.path.join(ROOT, lib)
        env["LT_ENABLE_TEST_SWITCHES"] = "1"  # the developer switches of an entry (LT_GEN_ROW_SLOTS, LT_SCORE_FUSED, ...)
        env["LT_FINE_TIMERS"] = env.get("LT_FINE_TIMERS", "2")
        if len(parts) > 2 and parts[2]:
            for kv in parts[2].split(","):
                k, v = kv.split("=", 1)
                env[k] = v
        try:
            p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", mode, str(runs)], env=env,
                               capture_output=True, text=True, timeout=180)
            line = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")]
            print(f"{label:24s}", line[-1][7:] if line else "FAILED rc=%d %s" % (p.returncode, p.stderr[-600:]), flush=True)
        except subprocess.TimeoutExpired:
            print(f"{label:24s} TIMEOUT", flush=True)


if __name__ == "__main__":
    main()
