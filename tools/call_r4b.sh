#!/bin/bash
mkdir -p gpurun_out
timeout 600 python tools/ab_score.py --runs 9 new:limap_amd/liblimap_amd.so rabl1_nodense:limap_amd/variants/librabl1.so rabl2_nodense_nosweep:limap_amd/variants/librabl2.so > gpurun_out/r4b_ab.log 2>&1
cat gpurun_out/r4b_ab.log
echo "=== trace new"; timeout 300 python tools/trace_score.py > gpurun_out/r4b_trace_new.log 2>&1; head -40 gpurun_out/r4b_trace_new.log
echo "=== trace old"; LT_SCORE_OLD=1 timeout 300 python tools/trace_score.py > gpurun_out/r4b_trace_old.log 2>&1; head -40 gpurun_out/r4b_trace_old.log
