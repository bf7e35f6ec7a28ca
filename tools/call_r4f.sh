#!/bin/bash
mkdir -p gpurun_out
repo=$(pwd); export TMPDIR=/tmp
cd /tmp && rm -rf /tmp/prof_f && timeout 600 rocprofv3 --kernel-trace --memory-copy-trace -d /tmp/prof_f -- python $repo/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $repo/gpurun_out/r4f_bench.json 2> $repo/gpurun_out/r4f_prof.err
cd $repo
db=$(find /tmp/prof_f -name "*.db" | head -1)
python tools/rocpd_step_timeline.py $db | tee gpurun_out/r4f_step_timeline.txt
python tools/rocpd_kernel_stats.py $db gpurun_out/r4f_kernel_stats.csv > /dev/null; head -30 gpurun_out/r4f_kernel_stats.csv | cut -c1-150
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4f_bench.json').read().strip().splitlines()[-1])
print({k:d.get(k) for k in ("ms_per_step","value","e2e_wall_ms","e2e_batched_ms","postprocess")})
PY
