#!/bin/bash
mkdir -p gpurun_out
( timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/r4c_pytest.log
tail -5 gpurun_out/r4c_pytest.log
timeout 600 python tools/ab_score.py --runs 11 split:limap_amd/liblimap_amd.so fused:limap_amd/liblimap_amd.so:LT_SCORE_FUSED=1 old:limap_amd/liblimap_amd.so:LT_SCORE_OLD=1 \
   split_b:limap_amd/liblimap_amd.so split_d8:limap_amd/liblimap_amd.so:LT_DENSE_RESIDENT=8 split_s8:limap_amd/liblimap_amd.so:LT_SWEEP_RESIDENT=8 > gpurun_out/r4c_ab.log 2>&1
cat gpurun_out/r4c_ab.log
echo "=== trace split"; timeout 300 python tools/trace_score.py > gpurun_out/r4c_trace_split.log 2>&1; head -30 gpurun_out/r4c_trace_split.log
