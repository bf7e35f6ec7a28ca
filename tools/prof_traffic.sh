#!/bin/bash
# HBM traffic per kernel launch of the bench step, as MI355X_MICROARCH.md prescribes: FETCH_SIZE and
# WRITE_SIZE in SEPARATE rocprofv3 --pmc passes (kernel dispatch only); FETCH_SIZE x2 (gfx950), both in KiB.
# usage (on the GPU box): bash tools/prof_traffic.sh <out.json>
out=${1:-gpurun_out/pmc_traffic.json}
repo=$(pwd)
export TMPDIR=/tmp
mkdir -p $repo/gpurun_out
for c in FETCH_SIZE WRITE_SIZE; do
  cd /tmp && rm -rf /tmp/traffic_$c
  timeout 300 rocprofv3 --pmc $c -d /tmp/traffic_$c -- python $repo/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > /dev/null 2> /tmp/traffic_$c.err
  db=$(find /tmp/traffic_$c -name "*.db" | head -1)
  python $repo/tools/rocpd_pmc.py $db 2>/dev/null > $repo/gpurun_out/traffic_$c.csv
done
cd $repo && python - "$out" <<'PY'
import csv, json, re, sys
out = sys.argv[1]
vals = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for row in csv.reader(open(f"gpurun_out/traffic_{c}.csv")):
        if len(row) != 5 or row[1] != c:
            continue
        m = re.search(r"(k_[a-z_0-9]+)", row[0])
        name = m.group(1) if m else row[0][:40]
        vals.setdefault(name, {})[c] = float(row[3])   # average per dispatch, KiB
kern = {}
for name, v in vals.items():
    f = 2.0 * 1024.0 * v.get("FETCH_SIZE", 0.0)
    w = 1024.0 * v.get("WRITE_SIZE", 0.0)
    kern[name] = {"fetch_bytes": f, "write_bytes": w, "hbm_bytes": f + w}
json.dump({"workload": "synthetic 100 views x 500 segs, nn=20, matched topk=10 (bench.py default)",
           "unit": "bytes per launch",
           "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (tools/prof_traffic.sh); "
                   "FETCH_SIZE x2 per the gfx950 correction of MI355X_MICROARCH.md; counters are in KiB",
           "kernels": kern}, open(out, "w"), indent=1)
for k in sorted(kern, key=lambda k: -kern[k]["hbm_bytes"])[:10]:
    print(k, {a: round(b / 1e6, 1) for a, b in kern[k].items()})
PY
