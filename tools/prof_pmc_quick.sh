#!/bin/bash
# Quick counter table for a few kernels: bash tools/prof_pmc_quick.sh <mode> <kernel-regex>   (on the GPU box)
# One rocprofv3 --pmc pass per counter group (kernel dispatch only), per-launch averages side by side.
mode=${1:-matched}; pat=${2:-k_}
repo=$(pwd)
export TMPDIR=/tmp
groups=(
"SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES"
"SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY"
"SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM"
)
: > /tmp/pmcq.csv
i=0
for g in "${groups[@]}"; do
  i=$((i+1))
  cd /tmp && rm -rf /tmp/pmcq_$i
  timeout 300 rocprofv3 --pmc $g -d /tmp/pmcq_$i -- python $repo/bench.py --steps 2 --warmup 1 --mode $mode --no-cpu-baseline --no-extras > /dev/null 2> /tmp/pmcq_$i.err || { echo "# group $i failed: $g"; tail -3 /tmp/pmcq_$i.err; continue; }
  db=$(find /tmp/pmcq_$i -name "*.db" | head -1)
  python $repo/tools/rocpd_pmc.py $db 2>/dev/null >> /tmp/pmcq.csv
done
cd $repo && python - "$pat" <<'PY'
import csv, re, sys
pat = re.compile(sys.argv[1])
vals = {}
for row in csv.reader(open("/tmp/pmcq.csv")):
    if len(row) != 5 or row[0] == "kernel":
        continue
    m = re.search(r"(k_[a-z_0-9]+)(I[A-Za-z0-9]*E)?", row[0])
    if not m or not pat.search(m.group(0)):
        continue
    vals.setdefault(m.group(0), {})[row[1]] = float(row[3])
for k, v in sorted(vals.items()):
    print(k)
    for c, x in sorted(v.items()):
        print(f"   {c:28s} {x:16.0f}")
PY
