#!/bin/bash
# Kernel-tuning experiments: build libmimosa_hip with extra -D flags for icp_kernels.hip into mimosa_amd/lib/variants/<tag>.so
#   usage: tools/variant.sh <tag> [--patch tools/variants/<name>.patch] -DMH_PIPE=8 ...
#          then  MH_LIB_OVERRIDE=mimosa_amd/lib/variants/<tag>.so python tools/k3_cold_probe.py
# Knobs that give CORRECT results are macros of icp_kernels.hip (MH_PIPE, MH_PIPE_SMALL, MH_PRUNE_TRIPS, MH_XCD_PIECES,
# MH_QL2_MAX / MH_QL4_MAX, MH_LOC_BLOCKS_512).  Timing-only bound experiments that give WRONG results are NOT in the product
# source: they live as patches under tools/variants/ (fake_bounds.patch: MH_FAKE_SCAN_ADDR, MH_FAKE_SCAN_MASK, MH_FAKE_TRIP_CAP,
# MH_FAKE_NO_EIGEN, MH_FAKE_NO_TAIL) and are applied to a COPY of the source here.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
tag=$1; shift
src=$R/mimosa_amd/csrc/icp_kernels.hip
python -m mimosa_amd.build > /dev/null
mkdir -p $R/mimosa_amd/lib/variants $R/mimosa_amd/build/variants
if [ "$1" = "--patch" ]; then
  cp $src $R/mimosa_amd/csrc/icp_kernels_variant_$tag.hip
  patch -s $R/mimosa_amd/csrc/icp_kernels_variant_$tag.hip < $2
  src=$R/mimosa_amd/csrc/icp_kernels_variant_$tag.hip
  shift; shift
fi
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wextra -Wno-unused-parameter "$@" -c $src -o $R/mimosa_amd/build/variants/icp_$tag.o
rm -f $R/mimosa_amd/csrc/icp_kernels_variant_$tag.hip
objs=$(ls $R/mimosa_amd/build/*.o | grep -v icp_kernels.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/mimosa_amd/lib/variants/$tag.so $objs $R/mimosa_amd/build/variants/icp_$tag.o -ldl
echo $R/mimosa_amd/lib/variants/$tag.so
