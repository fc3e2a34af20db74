#!/bin/bash
# round-5 GPU call 24: is the one-off stall under torch.distributed a full collection of CPython's garbage collector?
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/c24
mkdir -p $O
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29571 tools/torchrun_stall.py > $O/stall_gc.jsonl 2> $O/stall_gc.err
GC_FREEZE=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29572 tools/torchrun_stall.py > $O/stall_gc_freeze.jsonl 2> $O/stall_gc_freeze.err
PLAIN=1 timeout 600 python tools/torchrun_stall.py > $O/stall_gc_plain.jsonl 2> $O/stall_gc_plain.err
python3 - <<'PY'
import json
for f in ("stall_gc", "stall_gc_freeze", "stall_gc_plain"):
    print("==", f)
    for l in open(f"gpurun_out/c24/{f}.jsonl"):
        try:
            d = json.loads(l)
        except Exception:
            continue
        print(d["block"], d["burst_ms"], d["slowest_call"]["call"], d["slowest_call"]["index"], d["slowest_call"]["wall_ms"], "gc:", d["gc_collections_in_burst"], d["gc_objects_tracked"])
PY
