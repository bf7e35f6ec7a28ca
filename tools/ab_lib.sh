#!/bin/bash
# A/B of library variants on the bench step (on the GPU box): bash tools/ab_lib.sh [reps] libA.so libB.so ...
reps=$1; shift
pick='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d["kernel_ms"]; print(sys.argv[1], round(d["ms_per_step"],4), "run", round(k["run"],4), "gates", round(k["k_gates"],4), "tri", round(k["k_tri_rows"],4), "compact", round(k["compact"],4), "score", round(k["k_score3"],4), "pairs_eval", k["pairs_eval"])'
for i in $(seq $reps); do
  for lib in "$@"; do
    LIMAP_AMD_LIB=$(pwd)/$lib python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python -c "$pick" $lib
  done
done
