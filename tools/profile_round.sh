#!/bin/bash
# Round profile: rocprofv3 kernel trace + stats of the default bench command, then the PMC passes — each in its own run,
# kernel-trace only (gpurun refuses --pmc combined with the sys / hip / memory-copy trace domains), each under its own
# timeout.  Usage on a GPU box:  MH_ROUND=r02 MH_COMMIT=<sha> bash tools/profile_round.sh
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${MH_ROUND:-rXX}
OUT=$R/gpurun_out/prof
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 200 --warmup 20 --profile-mode --no-measure-traffic"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $CMD > $OUT/stats.log 2>&1
i=0
for pass in \
  "FETCH_SIZE" \
  "WRITE_SIZE" \
  "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
  "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" \
  "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64" \
  "SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_TRANS_F64 SQ_THREAD_CYCLES_VALU" \
  "TCP_TOTAL_ACCESSES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" \
  "TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
  "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $OUT/pmc_$i -- $CMD > $OUT/pmc_$i.log 2>&1
done
python3 - <<PY
import csv,glob,collections,json,os,re
def short(n):
    n=n.split('(')[0]
    m=re.search(r'(icp_\w+<[^>]*>|icp_\w+)',n)
    return m.group(1) if m else n[-60:]
out={}
for d in sorted(glob.glob('$OUT/pmc_*/')):
    f=glob.glob(d+'*/*_counter_collection.csv')
    if not f: continue
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        acc[short(r['Kernel_Name'])][r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in acc.items():
        if 'icp_' in k:
            out.setdefault(k,{}).update({c:sum(x)/len(x) for c,x in v.items()})
# per-dispatch register / LDS figures from the kernel trace
tr=glob.glob('$OUT/stats/*/*_kernel_trace.csv')
res={}
if tr:
    for r in csv.DictReader(open(tr[0])):
        n=short(r['Kernel_Name'])
        if 'icp_' in n and n not in res:
            res[n]={k:r.get(k) for k in ('VGPR_Count','Accum_VGPR_Count','SGPR_Count','LDS_Block_Size','Scratch_Size','Workgroup_Size','Grid_Size')}
json.dump({'counters':out,'resources':res,'commit':os.environ.get('MH_COMMIT','?'),'round':'$TAG'},open('$OUT/pmc_summary.json','w'),indent=1)
k3=[k for k in out if 'linearize_kernel' in k and 'batch' not in k]
if k3:
    c=out[k3[0]]
    import hashlib
    ksha=hashlib.sha256(open('$R/mimosa_amd/csrc/icp_kernels.hip','rb').read()).hexdigest()[:16]
    json.dump({'kernel':k3[0],'kernel_source_sha16':ksha,'FETCH_SIZE_KB':c.get('FETCH_SIZE'),'WRITE_SIZE_KB':c.get('WRITE_SIZE'),'TCC_EA0_RDREQ':c.get('TCC_EA0_RDREQ_sum'),
               'commit':os.environ.get('MH_COMMIT','?'),'round':'$TAG',
               'source':'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, bench.py --steps 200 --warmup 20 --profile-mode (tools/profile_round.sh)',
               'note':'FETCH_SIZE = TCC_EA0_RDREQ x 64 B on gfx950 (MI355X_MICROARCH.md HBM section): doubled as the guide prescribes for 128-B requests tallied at 64 B; uncalibrated for 16-B scattered gathers, so the read side is an upper bound'},
              open('$OUT/latest_pmc.json','w'),indent=1)
print(json.dumps(out,indent=1)[:6000])
PY
cat $OUT/stats/*/*_kernel_stats.csv | cut -c1-200
# gpurun merges at most 64 MiB back: keep the summaries, the logs and the stats table, drop the raw per-dispatch tables
mkdir -p $OUT/keep && cp $OUT/stats/*/*_kernel_stats.csv $OUT/keep/kernel_stats.csv 2>/dev/null
rm -rf $OUT/pmc_*/ $OUT/stats
