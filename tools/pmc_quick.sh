#!/bin/bash
# Developer tool (GPU box): SQ counters of the matched step's kernels, three rocprofv3 --pmc passes -> stdout (csv rows)
#   bash tools/pmc_quick.sh [kernel name pattern]
repo=$(pwd); export TMPDIR=/tmp; pat=${1:-k_}
groups=(
"SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU SQ_WAVES"
"SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT"
"SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
)
i=0
for g in "${groups[@]}"; do
  i=$((i+1)); cd /tmp && rm -rf /tmp/pmcq_$i
  timeout 150 rocprofv3 --pmc $g -d /tmp/pmcq_$i -- python $repo/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2> /tmp/pmcq_$i.err || { echo "# group $i failed"; tail -3 /tmp/pmcq_$i.err; continue; }
  python $repo/tools/rocpd_pmc.py $(find /tmp/pmcq_$i -name "*.db" | head -1) 2>/dev/null | grep -E "$pat" | sed 's/_ZN2lt[0-9]*\(k_[a-z_0-9]*\)[^,]*/\1/' | cut -c1-120
done
