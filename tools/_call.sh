mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -q 2>&1 | tail -25 > gpurun_out/t6.log
timeout 200 bash tools/ab_lib.sh 2 limap_amd/liblimap_amd.so > gpurun_out/ab6.log 2>&1
cat gpurun_out/t6.log gpurun_out/ab6.log
