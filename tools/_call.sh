mkdir -p gpurun_out
timeout 400 python -m pytest tests -m gpu -q 2>&1 | tail -6 > gpurun_out/t5.log
timeout 200 bash tools/ab_lib.sh 2 limap_amd/liblimap_amd.so > gpurun_out/ab5.log 2>&1
timeout 200 python bench.py --mode exhaustive --steps 5 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('exhaustive', d['ms_per_step'], d['kernel_ms'])" >> gpurun_out/ab5.log 2>&1
cat gpurun_out/t5.log gpurun_out/ab5.log
