#!/bin/bash
# Round profile: rocprofv3 kernel trace + stats of the default bench command, then the HBM-traffic
# PMC passes (each in its own run, kernel-trace only, each under its own timeout).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/prof
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 200 --warmup 20 --profile-mode"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $CMD > $OUT/stats.log 2>&1
for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "TCP_TOTAL_ACCESSES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum"; do
  tag=$(echo $pass | tr ' ' '_' | cut -c1-32)
  timeout 300 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $OUT/pmc_$tag -- $CMD > $OUT/pmc_$tag.log 2>&1
done
python3 - <<PY
import csv,glob,collections,json
out={}
for d in sorted(glob.glob('$OUT/pmc_*/')):
    f=glob.glob(d+'*/*_counter_collection.csv')
    if not f: continue
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        acc[r['Kernel_Name'].split('(')[0][-40:]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in acc.items():
        if 'icp_' in k:
            out.setdefault(k,{}).update({c:sum(x)/len(x) for c,x in v.items()})
json.dump(out,open('$OUT/pmc_summary.json','w'),indent=1)
print(json.dumps(out,indent=1))
PY
cat $OUT/stats/*/*_kernel_stats.csv
