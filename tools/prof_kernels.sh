#!/bin/bash
# Per-kernel durations of the bench step: rocprofv3 kernel trace -> gpurun_out/<tag>_kernel_stats.csv
# usage (on the GPU box): bash tools/prof_kernels.sh <tag> [extra bench args]
tag=${1:-run}; shift
repo=$(pwd)
export TMPDIR=/tmp
mkdir -p $repo/gpurun_out
cd /tmp && rm -rf /tmp/prof_$tag && timeout 600 rocprofv3 --kernel-trace -d /tmp/prof_$tag -- python $repo/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-extras "$@" > $repo/gpurun_out/${tag}_bench.json 2> $repo/gpurun_out/${tag}_prof.err
db=$(find /tmp/prof_$tag -name "*.db" | head -1)
python $repo/tools/rocpd_kernel_stats.py $db $repo/gpurun_out/${tag}_kernel_stats.csv > /dev/null
head -14 $repo/gpurun_out/${tag}_kernel_stats.csv | sed 's/"_ZN2lt[0-9]*\(k_[a-z_0-9]*\)[^"]*"/\1/' | cut -c1-120
python $repo/tools/rocpd_timeline.py $db > $repo/gpurun_out/${tag}_timeline.txt 2>&1
