#!/bin/bash
# round-5 GPU call 10: pipelined step by library variant (component server grid / K4 chunking / no server / round-4 library), sync latency vs kernarg placement
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/c10
mkdir -p $O
for v in base new srv128 loc4; do
  if [ $v = new ]; then unset MH_LIB_OVERRIDE; else export MH_LIB_OVERRIDE=$PWD/mimosa_amd/lib/variants/$v.so; fi
  timeout 300 python bench.py --headline-only --no-measure-traffic --no-cpu-baseline --steps 200 --warmup 20 > $O/head_$v.json 2> $O/head_$v.err
done
unset MH_LIB_OVERRIDE
MH_OVERLAP=0 timeout 300 python bench.py --headline-only --no-measure-traffic --no-cpu-baseline --steps 200 --warmup 20 > $O/head_nooverlap.json 2> $O/head_nooverlap.err
for kv in 0 1; do
  HIP_FORCE_DEV_KERNARG=$kv SYNC_PROBE_CALLS=100 timeout 300 python tools/sync_probe.py > $O/sync_kernarg$kv.json 2> $O/sync_kernarg$kv.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/c10/head_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d['value'], d['ms_per_step_blocks'], d['roofline']['kernel_ms_avg'])
    except Exception as e: print(f, 'ERR', e)
for f in sorted(glob.glob('gpurun_out/c10/sync_kernarg*.json')):
    d=json.loads(open(f).read()); print(f, {k:v['cold_p50'] for k,v in d.items() if isinstance(v,dict)})
PY
