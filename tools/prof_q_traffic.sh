#!/bin/bash
# Developer tool (GPU box, repo root): L2 memory-side traffic and VALU counters of the scoring stage in its one-kernel form
# (k_score_q) and in the two-kernel form (the default; one-kernel: LT_SCORE_ONE_KERNEL=1) -> gpurun_out/${TAG:-q}_traffic.csv
repo=$(pwd)
export TMPDIR=/tmp
mkdir -p $repo/gpurun_out
out=$repo/gpurun_out/${TAG:-q}_traffic.csv
: > $out
groups=(
"TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum"
"WRITE_SIZE"
"SQ_INSTS_VALU SQ_WAVES SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES"
)
for form in one two; do
  i=0
  for g in "${groups[@]}"; do
    i=$((i+1))
    cd /tmp && rm -rf /tmp/pq_${form}_$i
    if [ $form = two ]; then export LT_ENABLE_TEST_SWITCHES=1 LT_SCORE_ONE_KERNEL=0; unset LT_SCORE_ONE_KERNEL; else export LT_ENABLE_TEST_SWITCHES=1 LT_SCORE_ONE_KERNEL=1; fi
    timeout 150 rocprofv3 --pmc $g -d /tmp/pq_${form}_$i -- python $repo/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2> /tmp/pq_${form}_$i.err || { echo "# $form group $i failed"; tail -3 /tmp/pq_${form}_$i.err; continue; }
    db=$(find /tmp/pq_${form}_$i -name "*.db" | head -1)
    python $repo/tools/rocpd_pmc.py $db 2>/dev/null | grep -E "k_score|k_dense" | sed "s/^/$form,/" >> $out
  done
done
cat $out
