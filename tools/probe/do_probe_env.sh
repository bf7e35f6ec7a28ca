#!/bin/bash
# like do_probe.sh, for the in-tree library under environment settings: bash tools/probe/do_probe_env.sh "K=V K=V" ...
repo=$(pwd)
export TMPDIR=/tmp LT_ENABLE_TEST_SWITCHES=1 LT_FINE_TIMERS=0
i=0
for e in "$@"; do
  i=$((i+1)); cd /tmp && rm -rf /tmp/pe_$i
  env $e timeout 300 rocprofv3 --kernel-trace -d /tmp/pe_$i -- python $repo/tools/ab_score.py --child ${MODE:-exhaustive} 6 > /tmp/pe_$i.out 2>&1
  db=$(find /tmp/pe_$i -name "*.db" | head -1)
  python $repo/tools/rocpd_kernel_stats.py $db /tmp/pe_$i.csv > /dev/null
  echo "== $e"; grep RESULT /tmp/pe_$i.out | cut -c1-200
  head -9 /tmp/pe_$i.csv | sed 's/"_ZN2lt[0-9]*\(k_[a-z_0-9]*\)[^"]*"/\1/' | cut -c1-100
done
