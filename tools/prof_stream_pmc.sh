#!/bin/bash
# HBM traffic of the streamed leg's kernels (bench.py --stream, limap_amd/stream.py) per CHUNK -> gpurun_out/r06_stream_pmc.json
# (copy to profiles/).  Same recipe as tools/prof_pmc_json.sh: one rocprofv3 --pmc pass per counter group, kernel dispatch
# only.  The scene is a fifth of the leg's (1000 x 600 in four chunks of 250 images: the chunk -- what a launch works on -- is
# the same size, the passes stay short).
repo=$(pwd)
export TMPDIR=/tmp
mkdir -p $repo/gpurun_out
groups=("FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum")
: > $repo/gpurun_out/pmcj_stream.csv
i=0
for g in "${groups[@]}"; do
  i=$((i+1))
  cd /tmp && rm -rf /tmp/pmcs_$i
  LT_BENCH_STREAM_SCENE=1000,600,20,250 timeout 600 rocprofv3 --pmc $g -d /tmp/pmcs_$i -- python $repo/bench.py --stream > /dev/null 2> /tmp/pmcs_$i.err || { echo "# group $i failed: $g"; tail -3 /tmp/pmcs_$i.err; continue; }
  db=$(find /tmp/pmcs_$i -name "*.db" | head -1)
  python $repo/tools/rocpd_pmc.py $db 2>/dev/null >> $repo/gpurun_out/pmcj_stream.csv
done
cd $repo && python - <<'PY'
import csv, json, re, sys
sys.path.insert(0, ".")
import bench
vals = {}
for row in csv.reader(open("gpurun_out/pmcj_stream.csv")):
    if len(row) != 5 or row[0] == "kernel":
        continue
    m = re.search(r"(k_[a-z_0-9]+)", row[0])
    if m:
        vals.setdefault(m.group(1), {})[row[1]] = float(row[3])
def hbm(v):
    rd = 32.0 * v.get("TCC_EA0_RDREQ_32B_sum", 0.0) + 64.0 * v.get("TCC_EA0_RDREQ_64B_sum", 0.0) + 128.0 * v.get("TCC_EA0_RDREQ_128B_sum", 0.0)
    return rd + 1024.0 * v.get("WRITE_SIZE", 0.0), rd, 2048.0 * v.get("FETCH_SIZE", 0.0) + 1024.0 * v.get("WRITE_SIZE", 0.0)
out = {"device_source_hash": bench.device_source_hash(),
       "workload": "bench.py --stream on 1000 x 600 in four chunks of 250 images (LT_BENCH_STREAM_SCENE=1000,600,20,250): per-launch = per-chunk averages",
       "how": "tools/prof_stream_pmc.sh; units as in r06_pmc.json", "kernels": {}, "raw": vals}
names = {"k_gates": ["k_gates_ln"], "k_tri_rows": ["k_tri_rounds"], "k_score3": ["k_score3", "k_dense8", "k_score_q"]}
for key, ks in names.items():
    tot = [0.0, 0.0, 0.0]
    for k in ks:
        if k in vals:
            h = hbm(vals[k])
            tot = [a + b for a, b in zip(tot, h)]
    out["kernels"][key] = {"hbm_bytes_per_chunk": tot[0], "hbm_read_bytes_per_chunk": tot[1], "hbm_bytes_fetchsize_per_chunk": tot[2]}
json.dump(out, open("gpurun_out/r06_stream_pmc.json", "w"), indent=1)
print({k: {a: round(b / 1e6, 1) for a, b in v.items()} for k, v in out["kernels"].items()})
PY
