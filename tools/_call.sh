mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/t_gpu.log 2>&1; tail -3 gpurun_out/t_gpu.log
timeout 300 python bench.py --mode exhaustive --no-cpu-baseline --no-extras --steps 10 > gpurun_out/bx.json 2> gpurun_out/bx.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bx.json').read().strip().splitlines()[-1])
print(d["ms_per_step"], d.get("kernel_ms"))
PY
