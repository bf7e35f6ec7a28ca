#!/bin/bash
# round 4 measurement set (GPU box): bench line, config 3, forced-dist, kernel traces matched / exhaustive, PMC json
mkdir -p gpurun_out
# counters first: bench.py quotes profiles/r04_pmc.json only while its source hash matches the code that runs
if [ "$1" = "pmc" ]; then bash tools/prof_pmc_json.sh 2>&1 | tail -12; cp gpurun_out/r04_pmc.json profiles/r04_pmc.json; fi
( time timeout 900 python bench.py > gpurun_out/bench_r04.json 2> gpurun_out/bench_r04.err ) 2>&1 | grep real
timeout 300 python bench.py --config3 --no-cpu-baseline --no-extras --steps 5 > gpurun_out/bench_r04_config3_n1.json 2> gpurun_out/bench_r04_config3.err
LT_BENCH_FORCE_DIST=1 timeout 300 python bench.py --no-cpu-baseline --no-extras > gpurun_out/bench_r04_forcedist.json 2> gpurun_out/bench_r04_forcedist.err
for f in bench_r04 bench_r04_config3_n1 bench_r04_forcedist; do python - $f <<'PY'
import json,sys
d=json.loads(open('gpurun_out/%s.json'%sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], {k:d.get(k) for k in ("value","ms_per_step","step_with_merge_and_tail_ms","e2e_wall_ms","e2e_with_postprocess_ms","e2e_batched_ms","e2e_cold_ms","e2e_speedup_vs_cpu")}, d["roofline"].get("frac"), d["roofline"].get("traffic"), d.get("ranks",{}).get("n_ranks_rccl"), d.get("cpu_parity",{}).get("ok"), d.get("postprocess",{}).get("ms"))
PY
done
bash tools/prof_kernels.sh r04 > gpurun_out/r04_prof_kernels.log 2>&1; head -16 gpurun_out/r04_prof_kernels.log
bash tools/prof_kernels.sh r04_exhaustive --mode exhaustive > gpurun_out/r04_prof_kernels_ex.log 2>&1; head -12 gpurun_out/r04_prof_kernels_ex.log
db=$(find /tmp/prof_r04 -name "*.db" | head -1); python tools/rocpd_step_timeline.py $db > gpurun_out/r04_step_timeline.txt; cat gpurun_out/r04_step_timeline.txt
