#!/bin/bash
# Build a variant of liblimap_amd.so that differs in the device-compile flags of lt_kernels_v2.hip (stage kernels;
# V2FLAGS, default as in the Makefile) and lt_kernels_score.hip (scoring) only:
#   bash tools/build_variant.sh NAME "-DLT_SCORE4_WAVES_PER_EU=3 ..."   ->  limap_amd/variants/libNAME.so
# (run in the container; the variants travel to the GPU box with the snapshot, see tools/ab_lib.sh)
set -e
name=$1; flags=$2
cd "$(dirname "$0")/../limap_amd/csrc"
mkdir -p ../variants build
make -s all
V2FLAGS=${V2FLAGS--mllvm -disable-machine-licm}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wno-pass-failed $V2FLAGS $flags -c lt_kernels_v2.hip -o build/v2_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wno-pass-failed ${SCFLAGS--mllvm -disable-machine-licm} $flags -c lt_kernels_score.hip -o build/score_$name.o
g++ -shared -o ../variants/lib$name.so build/lt_kernels.o build/v2_$name.o build/score_$name.o build/lt_kernels_tail.o build/lt_api.o build/lt_api_rows.o \
  build/lt_api_run.o build/lt_api_tail.o build/lt_api_query.o build/lt_tracks.o \
  -L/opt/rocm/lib -lamdhip64 -ldl -fopenmp -Wl,-rpath,/opt/rocm/lib
echo "built limap_amd/variants/lib$name.so ($flags)"
