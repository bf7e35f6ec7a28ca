mkdir -p gpurun_out
( time timeout 900 python bench.py > gpurun_out/bench_r03.json 2> gpurun_out/bench_r03.err ) 2>&1 | grep real
timeout 300 python bench.py --config3 --no-cpu-baseline --no-extras --steps 5 > gpurun_out/bench_r03_config3_n1.json 2> gpurun_out/bench_r03_config3.err
LT_BENCH_FORCE_DIST=1 timeout 300 python bench.py --no-cpu-baseline --no-extras > gpurun_out/bench_r03_forcedist.json 2> gpurun_out/bench_r03_forcedist.err
wc -l gpurun_out/bench_r03.json gpurun_out/bench_r03_config3_n1.json gpurun_out/bench_r03_forcedist.json
for f in bench_r03 bench_r03_config3_n1 bench_r03_forcedist; do python - $f <<'PY'
import json,sys
d=json.loads(open('gpurun_out/%s.json'%sys.argv[1]).read())
print(sys.argv[1], {k:d.get(k) for k in ("value","ms_per_step","step_with_merge_and_tail_ms","e2e_wall_ms","e2e_batched_ms","e2e_cold_ms","e2e_speedup_vs_cpu")}, d["roofline"].get("frac"), d["roofline"].get("traffic"), d.get("ranks",{}).get("n_ranks_rccl"), d.get("cpu_parity",{}).get("ok"))
PY
done
