#!/bin/bash
# PMC passes over the bench step (one rocprofv3 --pmc run per counter group, kernel dispatch only).
# usage (on the GPU box): bash tools/prof_pmc.sh <tag> ; output gpurun_out/<tag>_pmc.csv
tag=${1:-pmc}; shift
repo=$(pwd)
export TMPDIR=/tmp
mkdir -p $repo/gpurun_out
out=$repo/gpurun_out/${tag}_pmc.csv
: > $out
groups=(
"SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_LDS"
"SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA SQ_INSTS_SALU"
"SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM"
"SQ_INST_LEVEL_SMEM SQ_INSTS_SMEM"
"TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum"
"TCP_PENDING_STALL_CYCLES_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"
"TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"
"TA_BUSY_avr TA_TA_BUSY_sum TCC_TAG_STALL_sum GRBM_GUI_ACTIVE"
)
i=0
for g in "${groups[@]}"; do
  i=$((i+1))
  cd /tmp && rm -rf /tmp/pmc_${tag}_$i
  timeout 300 rocprofv3 --pmc $g -d /tmp/pmc_${tag}_$i -- python $repo/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras "$@" > /dev/null 2> /tmp/pmc_${tag}_$i.err || { echo "# group $i failed: $g" >> $out; tail -3 /tmp/pmc_${tag}_$i.err >> $out; continue; }
  db=$(find /tmp/pmc_${tag}_$i -name "*.db" | head -1)
  python $repo/tools/rocpd_pmc.py $db 2>/dev/null | grep -E "k_gates|k_tri_rows|k_place|k_score3|k_gen_rows|k_select|k_gather" >> $out
done
sed 's/"_ZN2lt[0-9]*\(k_[a-z_0-9]*\)[^"]*"/\1/' $out
