#!/bin/bash
# Kernel-tuning experiments: build libmimosa_hip with extra -D flags for icp_kernels.hip into mimosa_amd/lib/variants/<tag>.so
#   usage: tools/variant.sh <tag> -DMH_PIPE=8 ...      then  MH_LIB_OVERRIDE=mimosa_amd/lib/variants/<tag>.so python bench.py
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
tag=$1; shift
python -m mimosa_amd.build > /dev/null
mkdir -p $R/mimosa_amd/lib/variants $R/mimosa_amd/build/variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wextra -Wno-unused-parameter "$@" -c $R/mimosa_amd/csrc/icp_kernels.hip -o $R/mimosa_amd/build/variants/icp_$tag.o
objs=$(ls $R/mimosa_amd/build/*.o | grep -v icp_kernels.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/mimosa_amd/lib/variants/$tag.so $objs $R/mimosa_amd/build/variants/icp_$tag.o -ldl
echo $R/mimosa_amd/lib/variants/$tag.so
