#!/bin/bash
# round-5 GPU call 28: bench.py under the driver's torchrun line at one rank (the garbage collector frozen out of the timed region)
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/c28
mkdir -p $O
for k in 1 2; do
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2958$k bench.py --gpus 1 --steps 20 --warmup 5 --headline-only > $O/bench_torchrun_$k.json 2> $O/bench_torchrun_$k.err
tail -n 1 $O/bench_torchrun_$k.json | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print({k:d.get(k) for k in ('value','ms_per_step','ms_per_step_blocks','n_gpus','scaling')}, d['config'].get('mode'), d.get('sharded_rccl'))"
done
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29585 bench.py --gpus 1 --steps 200 --warmup 20 --headline-only > $O/bench_torchrun_200.json 2> $O/bench_torchrun_200.err
tail -n 1 $O/bench_torchrun_200.json | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print({k:d.get(k) for k in ('value','ms_per_step','ms_per_step_blocks')})"
tail -n 3 $O/bench_torchrun_1.err
