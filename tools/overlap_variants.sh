#!/bin/bash
# Pipelined step (bench.py --profile-mode: the timed region only) and K3 / server figures by rocprofv3 for the library and the
# variants under mimosa_amd/lib/variants/.  usage (GPU box): tools/overlap_variants.sh [tag ...] -> gpurun_out/overlap_variants.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
tags=${@:-$(ls mimosa_amd/lib/variants/*.so 2>/dev/null | xargs -n1 basename | sed 's/\.so$//')}
: > gpurun_out/overlap_variants.txt
for t in base $tags; do
  if [ $t = base ]; then unset MH_LIB_OVERRIDE; else export MH_LIB_OVERRIDE=$R/mimosa_amd/lib/variants/$t.so; fi
  s=""
  for rep in 1 2 3; do
    v=$(timeout 200 python bench.py --profile-mode --steps 640 --warmup 64 --no-measure-traffic 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['kernel_ms_avg'])")
    s="$s | $v"
  done
  tools/k3_prof.sh ov_$t > /tmp/ov_$t.txt 2>&1
  k3=$(grep "icp_linearize_kernel" /tmp/ov_$t.txt | head -1 | awk -F, '{printf "%.2f", $(NF-4)/1000}')
  echo "$t step_ms,k3_events_ms $s | K3 rocprof $k3 us" | tee -a gpurun_out/overlap_variants.txt
  rm -f gpurun_out/k3prof_ov_$t.csv
done
