timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
timeout 120 python tools/profile_e2e_batched.py 2>&1 | grep -E "^rep[123]" | tail -12
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/b1.json 2> gpurun_out/b1.err; wc -l gpurun_out/b1.json
python - <<'PY'
import json
b=json.loads(open('gpurun_out/b1.json').read())
for k in ("ms_per_step","e2e_wall_ms","e2e_cold_ms","e2e_reps_ms","e2e_batched_ms","e2e_batched_reps_ms","e2e_breakdown_ms","e2e_batched_breakdown_ms"): print(k, b.get(k))
PY
