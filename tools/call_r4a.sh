#!/bin/bash
# round 4, GPU call A: full GPU suite + A/B of the sweep-record scoring against the round-3 path
mkdir -p gpurun_out
( timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/r4a_pytest.log
tail -5 gpurun_out/r4a_pytest.log
timeout 900 python tools/ab_score.py --runs 11 new:limap_amd/liblimap_amd.so old:limap_amd/liblimap_amd.so:LT_SCORE_OLD=1 \
  new_licmoff:limap_amd/variants/libslicm.so new_b:limap_amd/liblimap_amd.so old_b:limap_amd/liblimap_amd.so:LT_SCORE_OLD=1 \
  > gpurun_out/r4a_ab.log 2>&1
cat gpurun_out/r4a_ab.log
timeout 300 python bench.py --no-cpu-baseline --no-extras > gpurun_out/r4a_bench.json 2> gpurun_out/r4a_bench.err; tail -c 1500 gpurun_out/r4a_bench.json
