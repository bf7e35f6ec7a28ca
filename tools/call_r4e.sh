#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -X faulthandler -m pytest tests/test_gpu_guards.py tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r4e_pytest.log 2>&1
grep -n "passed\|failed\|Fatal\|Error\|error" gpurun_out/r4e_pytest.log | head -20
timeout 600 python tools/ab_score.py --runs 11 split:limap_amd/liblimap_amd.so fused:limap_amd/liblimap_amd.so:LT_SCORE_FUSED=1 \
   split_b:limap_amd/liblimap_amd.so split_d8:limap_amd/liblimap_amd.so:LT_DENSE_RESIDENT=8 nosplit:limap_amd/liblimap_amd.so:LT_SCORE_NO_SPLIT=1 > gpurun_out/r4e_ab.log 2>&1
cat gpurun_out/r4e_ab.log
echo "=== trace split"; timeout 300 python tools/trace_score.py > gpurun_out/r4e_trace_split.log 2>&1; head -22 gpurun_out/r4e_trace_split.log
