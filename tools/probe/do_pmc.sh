#!/bin/bash
# counters of one kernel of the exhaustive run (default k_depth_order): bash tools/probe/do_pmc.sh [kernel-substring]
repo=$(pwd); k=${1:-depth_order}
export TMPDIR=/tmp LT_ENABLE_TEST_SWITCHES=1 LT_FINE_TIMERS=0
groups=(
"FETCH_SIZE"
"WRITE_SIZE"
"SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT"
"SQ_INSTS_VALU SQ_WAVES SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
"SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_WAIT_INST_LDS SQ_IFETCH"
)
i=0
for g in "${groups[@]}"; do
  i=$((i+1)); cd /tmp && rm -rf /tmp/dp_$i
  timeout 200 rocprofv3 --pmc $g -d /tmp/dp_$i -- python $repo/tools/ab_score.py --child exhaustive 1 > /dev/null 2> /tmp/dp_$i.err || { echo "# group $i failed: $g"; tail -3 /tmp/dp_$i.err; continue; }
  db=$(find /tmp/dp_$i -name "*.db" | head -1)
  python $repo/tools/rocpd_pmc.py $db 2>/dev/null | grep -i "$k"
done
