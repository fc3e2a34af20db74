#!/bin/bash
# usage: scripts_pmc.sh <tag> ; PMC passes for the bench command, each in its own run (kernel-trace only)
R=$GRAFT_REPO_ROOT
TAG=${1:-pmc}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline"
i=0
while IFS= read -r pass; do
  [ -z "$pass" ] && continue
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $OUT/p$i -- $CMD > $OUT/p$i.log 2>&1
done <<'PASSES'
SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT
SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS
TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCP_LATENCY_sum TCP_TOTAL_ACCESSES_sum
TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_STALL_MULTI_MISS_sum
TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TOTAL_WAVEFRONTS_sum
TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_GATE_EN1_sum
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum
FETCH_SIZE
WRITE_SIZE
GRBM_GUI_ACTIVE
PASSES
python3 - <<PY
import csv,glob,collections
for d in sorted(glob.glob('$OUT/p*/')):
    f=glob.glob(d+'*/*_counter_collection.csv')
    if not f: print(d,'no csv'); continue
    rows=list(csv.DictReader(open(f[0])))
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        acc[r['Kernel_Name'][:34]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in acc.items():
        if 'icp_' not in k: continue
        print(k, {c:round(sum(x)/len(x),1) for c,x in v.items()})
PY
