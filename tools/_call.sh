mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -q 2>&1 | tail -6 > gpurun_out/t10.log
timeout 200 bash tools/ab_lib.sh 2 limap_amd/liblimap_amd.so > gpurun_out/ab10.log 2>&1
timeout 500 python bench.py --no-cpu-baseline --no-extras > gpurun_out/bench_nocpu.json 2> gpurun_out/bench_nocpu.err
cat gpurun_out/t10.log gpurun_out/ab10.log; python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_nocpu.json').read().strip().splitlines()[-1])
for k in ("ms_per_step","e2e_wall_ms","e2e_cold_ms","e2e_breakdown_ms","e2e_batched_ms","e2e_batched_breakdown_ms","e2e_batched_reps_ms"):
    print(k, d.get(k))
PY
