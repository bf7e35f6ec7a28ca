#!/bin/bash
# Round-6 measurement set on the GPU box (repo root): kernel stats / timelines (rocprofv3 --kernel-trace) of the default
# bench step, of the exhaustive mode and of BASELINE config 3; counter file (tools/prof_pmc_json.sh); bench lines.
# Everything lands in gpurun_out/r06_*; copy what is to be judged into profiles/.
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
run_trace r06 --steps 30 --warmup 3
run_trace r06_exhaustive --steps 6 --warmup 1 --mode exhaustive
run_trace r06_config3 --steps 12 --warmup 2 --config3
TAG=r06 bash tools/prof_pmc_json.sh > gpurun_out/r06_pmc_summary.txt 2>&1
bash tools/prof_stream_pmc.sh > gpurun_out/r06_stream_pmc_summary.txt 2>&1
cp gpurun_out/r06_pmc.json gpurun_out/r06_stream_pmc.json profiles/ 2>/dev/null  # (bench.py quotes the counters of the matching hash)
python bench.py > gpurun_out/bench_r06.json 2> gpurun_out/bench_r06.err
python bench.py --stream > gpurun_out/bench_r06_stream.json 2> gpurun_out/bench_r06_stream.err
python bench.py --mode exhaustive --no-cpu-baseline --no-extras > gpurun_out/bench_r06_exhaustive.json 2> gpurun_out/bench_r06_exhaustive.err
python bench.py --config3 --no-cpu-baseline --no-extras > gpurun_out/bench_r06_config3_n1.json 2> gpurun_out/bench_r06_config3.err
LT_BENCH_FORCE_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29541 \
    bench.py --gpus 1 --no-cpu-baseline > gpurun_out/bench_r06_forcedist.json 2> gpurun_out/bench_r06_forcedist.err
head -12 gpurun_out/r06_kernel_stats.csv | cut -c1-150
cat gpurun_out/r06_step_timeline.txt
tail -8 gpurun_out/r06_pmc_summary.txt | cut -c1-300
tail -3 gpurun_out/r06_stream_pmc_summary.txt | cut -c1-400
