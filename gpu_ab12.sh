#!/bin/bash
mkdir -p gpurun_out
timeout 900 python tools/ab_score.py --mode exhaustive --runs 9 main:limap_amd/liblimap_amd.so k1licm:limap_amd/variants/libK1.so main_b:limap_amd/liblimap_amd.so k1licm_b:limap_amd/variants/libK1.so 2>&1 | tee gpurun_out/ab12.log
