#!/bin/bash
# round-5 GPU call 40: the driver's launch line at one rank with EVERY leg (process group + the library's own RCCL communicator in one process)
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/c40
mkdir -p $O
timeout 1500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29591 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_torchrun_full.json 2> $O/bench_torchrun_full.err
echo "rc $?"
tail -n 1 $O/bench_torchrun_full.json | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print({k:d.get(k) for k in ('value','ms_per_step','n_gpus','value_sync','sync_latency_ms','value_sharded')})
print(d.get('sharded_rccl')); print({k:d['sharded'].get(k) for k in ('n_ranks','backend','rccl_ranks','value','ms_per_cold_linearize')} if isinstance(d.get('sharded'),dict) else d.get('sharded'))
print(d['config'])"
tail -n 5 $O/bench_torchrun_full.err | cut -c1-300
