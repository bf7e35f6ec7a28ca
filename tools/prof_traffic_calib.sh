#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of tools/probe/traffic_probe's kernels (known byte counts) -> gpurun_out/r03_traffic_calib.txt
# (on the GPU box, repo root; separate --pmc passes, kernel dispatch only)
repo=$(pwd)
export TMPDIR=/tmp
out=$repo/gpurun_out/r03_traffic_calib.txt
mkdir -p $repo/gpurun_out
cd /tmp
$repo/tools/probe/traffic_probe > $out 2>&1
for c in FETCH_SIZE WRITE_SIZE "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum"; do
  tag=$(echo $c | cut -d" " -f1)
  rm -rf /tmp/calib_$tag
  timeout 90 rocprofv3 --pmc $c -d /tmp/calib_$tag -- $repo/tools/probe/traffic_probe > /dev/null 2> /tmp/calib_$tag.err || { echo "# $c failed" >> $out; tail -3 /tmp/calib_$tag.err >> $out; continue; }
  db=$(find /tmp/calib_$tag -name "*.db" | head -1)
  echo "# counter $c (KiB per dispatch, instances summed; kernels in launch order, two repetitions averaged)" >> $out
  python $repo/tools/rocpd_pmc.py $db 2>/dev/null | sed 's/"\([^"(]*\).*",/\1,/' >> $out
done
cat $out
