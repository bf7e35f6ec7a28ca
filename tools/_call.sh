mkdir -p gpurun_out
bash tools/prof_kernels.sh r03 > gpurun_out/prof_r03.log 2>&1
bash tools/prof_kernels.sh r03_exhaustive --mode exhaustive > gpurun_out/prof_r03x.log 2>&1
timeout 900 bash tools/prof_pmc_json.sh > gpurun_out/pmc_r03.log 2>&1
cp gpurun_out/r03_pmc.json profiles/r03_pmc.json
timeout 600 python bench.py > gpurun_out/bench_r03.json 2> gpurun_out/bench_r03.err
timeout 300 python bench.py --config3 --no-cpu-baseline --no-extras --steps 5 > gpurun_out/bench_r03_config3_n1.json 2> gpurun_out/bench_r03_config3.err
LT_BENCH_FORCE_DIST=1 timeout 300 python bench.py --no-cpu-baseline --no-extras > gpurun_out/bench_r03_forcedist.json 2> gpurun_out/bench_r03_forcedist.err
(timeout 600 python tools/fuzz_parity.py 120 7000; timeout 600 python tools/fuzz_parity.py 40 9000 big) > gpurun_out/fuzz_r03.log 2>&1
tail -3 gpurun_out/pmc_r03.log; tail -4 gpurun_out/fuzz_r03.log
for f in bench_r03 bench_r03_config3_n1 bench_r03_forcedist; do python - $f <<'PY'
import json,sys
try:
    d=json.loads(open('gpurun_out/%s.json'%sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1], {k:d.get(k) for k in ("value","ms_per_step","step_with_merge_and_tail_ms","e2e_wall_ms","e2e_batched_ms","n_gpus")}, d["roofline"].get("frac"), d["roofline"].get("traffic"), d.get("ranks",{}).get("n_ranks_rccl"))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
