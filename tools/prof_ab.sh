#!/bin/bash
# Developer tool (GPU box): per-kernel durations of tools/ab_score.py's run loop for library builds, by rocprofv3
#   bash tools/prof_ab.sh label:lib.so[:K=V,...] ...
repo=$(pwd); export TMPDIR=/tmp
for ent in "$@"; do
  IFS=: read -r label lib envs <<< "$ent"
  rm -rf /tmp/pab_$label
  ( cd /tmp; IFS=,; for kv in $envs; do export "$kv"; done; LIMAP_AMD_LIB=$repo/$lib timeout 300 rocprofv3 --kernel-trace -d /tmp/pab_$label -- python $repo/tools/ab_score.py --child matched 15 > /tmp/pab_$label.out 2>&1 )
  db=$(find /tmp/pab_$label -name "*.db" | head -1)
  python $repo/tools/rocpd_kernel_stats.py $db /tmp/pab_$label.csv > /dev/null
  echo "== $label"; grep -E "k_score3|k_dense8|k_gates|k_tri_rows" /tmp/pab_$label.csv | sed 's/"_ZN2lt[0-9]*\(k_[a-z_0-9]*\)[^"]*"/\1/' | cut -c1-80
done
