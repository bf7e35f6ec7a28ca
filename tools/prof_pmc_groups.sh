#!/bin/bash
# Counter table for a kernel regex with arbitrary counter groups (one rocprofv3 --pmc pass per group, dispatch only):
#   bash tools/prof_pmc_groups.sh <kernel-regex> "<group 1 counters>" "<group 2 counters>" ...   (on the GPU box)
pat=$1; shift
repo=$(pwd)
export TMPDIR=/tmp
: > /tmp/pmcg.csv
i=0
for g in "$@"; do
  i=$((i+1))
  cd /tmp && rm -rf /tmp/pmcg_$i
  timeout 300 rocprofv3 --pmc $g -d /tmp/pmcg_$i -- python $repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2> /tmp/pmcg_$i.err || { echo "# group $i failed: $g"; tail -3 /tmp/pmcg_$i.err; continue; }
  db=$(find /tmp/pmcg_$i -name "*.db" | head -1)
  python $repo/tools/rocpd_pmc.py $db 2>/dev/null >> /tmp/pmcg.csv
done
cd $repo && python - "$pat" <<'PY'
import csv, re, sys
pat = re.compile(sys.argv[1])
vals = {}
for row in csv.reader(open("/tmp/pmcg.csv")):
    if len(row) != 5 or row[0] == "kernel":
        continue
    m = re.search(r"(k_[a-z_0-9]+)(I[A-Za-z0-9]*E)?", row[0])
    if not m or not pat.search(m.group(0)):
        continue
    vals.setdefault(m.group(0), {})[row[1]] = float(row[3])
for k, v in sorted(vals.items()):
    print(k)
    for c, x in sorted(v.items()):
        print(f"   {c:40s} {x:16.0f}")
PY
