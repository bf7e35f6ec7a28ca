"""A/B timing of extension builds: runs bench.py once per library given on the command line
(LIMAP_AMD_LIB override) and prints the per-stage kernel times side by side.
usage: python tools/ab_bench.py [--env K=V ...] lib1.so lib2.so ...   (run on the GPU box)"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    args = sys.argv[1:]
    runs = []
    env_extra = {}
    for a in args:
        if a.startswith("--env="):
            k, v = a[len("--env="):].split("=", 1)
            env_extra = dict(env_extra)
            env_extra[k] = v
        elif a == "--clear-env":
            env_extra = {}
        else:
            runs.append((a, dict(env_extra)))
    for lib, extra in runs:
        env = dict(os.environ)
        env.update(extra)
        env["LIMAP_AMD_LIB"] = os.path.abspath(lib)
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "3",
                              "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=600)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")]
        if not line:
            print(lib, extra, "FAILED", out.stderr[-400:])
            continue
        d = json.loads(line[-1])
        k = d["kernel_ms"]
        print(f"{os.path.basename(lib):28s} {extra} step={d['ms_per_step']:.4f} gen={k['gen']:.4f} "
              f"compact={k['compact']:.4f} score={k['score']:.4f} select={k['select']:.4f} gather={k['gather']:.4f}",
              flush=True)


if __name__ == "__main__":
    main()
