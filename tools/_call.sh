mkdir -p gpurun_out
timeout 300 python tools/trace_score5.py > gpurun_out/trace5.log 2>&1
cat gpurun_out/trace5.log
