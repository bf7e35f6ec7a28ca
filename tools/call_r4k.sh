#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -X faulthandler -m pytest tests -m gpu -x -q > gpurun_out/r4k_pytest.log 2>&1
grep -n "passed\|failed\|Fatal\|Error\|error" gpurun_out/r4k_pytest.log | head -20
timeout 600 python tools/ab_score.py --runs 11 pre:limap_amd/liblimap_amd.so old:limap_amd/liblimap_amd.so:LT_TEST_NO_PREGATE=1 pre_w6:limap_amd/variants/libg32w6.so \
   pre_b:limap_amd/liblimap_amd.so old_b:limap_amd/liblimap_amd.so:LT_TEST_NO_PREGATE=1 > gpurun_out/r4k_ab.log 2>&1
cat gpurun_out/r4k_ab.log
