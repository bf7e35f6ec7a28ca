#!/bin/bash
# k_depth_order and its neighbours under rocprofv3 for the libraries given (default: the in-tree one)
repo=$(pwd)
export TMPDIR=/tmp LT_ENABLE_TEST_SWITCHES=1 LT_FINE_TIMERS=0
mkdir -p $repo/gpurun_out
for v in ${@:-liblimap_amd.so}; do
  tag=$(basename $v .so)
  cd /tmp && rm -rf /tmp/p_$tag
  LIMAP_AMD_LIB=$repo/limap_amd/$v timeout 300 rocprofv3 --kernel-trace -d /tmp/p_$tag -- python $repo/tools/ab_score.py --child exhaustive 6 > $repo/gpurun_out/do_$tag.out 2>&1
  db=$(find /tmp/p_$tag -name "*.db" | head -1)
  python $repo/tools/rocpd_kernel_stats.py $db $repo/gpurun_out/do_${tag}_stats.csv > /dev/null
  echo "== $tag"; grep RESULT $repo/gpurun_out/do_$tag.out | cut -c1-250
  grep -i "depth_order\|k_score3\|k_place_ex\|k_cand_meta\|k_tri_ex\|k_gates_ex" $repo/gpurun_out/do_${tag}_stats.csv | sed 's/"_ZN2lt[0-9]*\(k_[a-z_0-9]*\)[^"]*"/\1/' | cut -c1-110
done
