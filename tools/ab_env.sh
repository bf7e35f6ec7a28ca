#!/bin/bash
# A/B of an environment switch on the bench step (on the GPU box): bash tools/ab_env.sh VAR [reps]
var=$1; reps=${2:-2}
pick='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d["kernel_ms"]; print(sys.argv[1], round(d["ms_per_step"],4), "run", round(k["run"],4), "gates", round(k["k_gates"],4), "tri", round(k["k_tri_rows"],4), "compact", round(k["compact"],4), "score", round(k["k_score3"],4))'
for i in $(seq $reps); do
  python bench.py --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "$pick" base
  env $var=1 python bench.py --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "$pick" $var
done
