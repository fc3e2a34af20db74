#!/bin/bash
# K3 / K4 rocprofv3 averages of the default bench command for the library and for every variant under mimosa_amd/lib/variants/
# (tools/variant.sh builds them).  usage (GPU box): tools/variant_time.sh [tag ...]   -> gpurun_out/variant_time.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
tags=${@:-$(ls mimosa_amd/lib/variants/*.so 2>/dev/null | xargs -n1 basename | sed 's/\.so$//')}
: > gpurun_out/variant_time.txt
for t in base $tags; do
  if [ $t = base ]; then unset MH_LIB_OVERRIDE; else export MH_LIB_OVERRIDE=$R/mimosa_amd/lib/variants/$t.so; fi
  tools/k3_prof.sh v_$t > /tmp/v_$t.txt 2>&1
  k3=$(grep "icp_linearize_kernel" /tmp/v_$t.txt | head -1 | awk -F, '{printf "%.2f", $(NF-4)/1000}')
  k4=$(grep "icp_localizability_kernel" /tmp/v_$t.txt | head -1 | awk -F, '{printf "%.2f", $(NF-4)/1000}')
  echo "$t K3 $k3 us K4 $k4 us" | tee -a gpurun_out/variant_time.txt
  rm -f gpurun_out/k3prof_v_$t.csv
done
