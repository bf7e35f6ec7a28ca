#!/bin/bash
# tools/kres.sh FILE.hip [extra flags] -- registers / spills / LDS / occupancy of every kernel of a translation unit
# (hipcc -Rpass-analysis=kernel-resource-usage, gfx950; run from limap_amd/csrc)
f=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wno-pass-failed "$@" \
  -Rpass-analysis=kernel-resource-usage -c "$f" -o /tmp/kres_tmp.o 2>&1 |
  awk '/Function Name:/ {name=$(NF-1)} / VGPRs:/ {v=$(NF-1)} /AGPRs:/ {a=$(NF-1)} /ScratchSize/ {s=$(NF-1)} /Occupancy/ {o=$(NF-1)} /SGPRs Spill/ {ss=$(NF-1)} /VGPRs Spill/ {vs=$(NF-1)} /LDS Size/ {printf "%-90s VGPR %3s AGPR %3s scratch %4s occ %2s spillS %s spillV %s LDS %s\n", name, v, a, s, o, ss, vs, $(NF-1)}' | c++filt | cut -c1-200
