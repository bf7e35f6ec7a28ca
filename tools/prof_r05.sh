#!/bin/bash
# Round-5 measurement set on the GPU box (repo root): kernel stats / timelines (rocprofv3 --kernel-trace) of the default
# bench step, of the exhaustive mode and of BASELINE config 3; counter file (tools/prof_pmc_json.sh); bench lines.
# Everything lands in gpurun_out/r05_*; copy what is to be judged into profiles/.
repo=$(pwd)
export TMPDIR=/tmp
mkdir -p $repo/gpurun_out
run_trace() {  # tag, bench args...
  tag=$1; shift
  cd /tmp && rm -rf /tmp/prof_$tag
  timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_$tag -- python $repo/bench.py --no-cpu-baseline --no-extras "$@" \
      > $repo/gpurun_out/${tag}_trace_bench.json 2> $repo/gpurun_out/${tag}_prof.err
  db=$(find /tmp/prof_$tag -name "*.db" | head -1)
  python $repo/tools/rocpd_kernel_stats.py $db $repo/gpurun_out/${tag}_kernel_stats.csv > /dev/null
  python $repo/tools/rocpd_timeline.py $db > $repo/gpurun_out/${tag}_timeline.txt 2>&1
  python $repo/tools/rocpd_step_timeline.py $db > $repo/gpurun_out/${tag}_step_timeline.txt 2>&1
  cd $repo
}
run_trace r05 --steps 30 --warmup 3
run_trace r05_exhaustive --steps 6 --warmup 1 --mode exhaustive
run_trace r05_config3 --steps 12 --warmup 2 --config3
bash tools/prof_pmc_json.sh > gpurun_out/r05_pmc_summary.txt 2>&1
python bench.py > gpurun_out/bench_r05.json 2> gpurun_out/bench_r05.err
python bench.py --config3 --no-cpu-baseline --no-extras > gpurun_out/bench_r05_config3_n1.json 2> gpurun_out/bench_r05_config3.err
LT_BENCH_FORCE_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29541 \
    bench.py --gpus 1 --no-cpu-baseline > gpurun_out/bench_r05_forcedist.json 2> gpurun_out/bench_r05_forcedist.err
head -12 gpurun_out/r05_kernel_stats.csv | cut -c1-150
cat gpurun_out/r05_step_timeline.txt
tail -8 gpurun_out/r05_pmc_summary.txt | cut -c1-300
