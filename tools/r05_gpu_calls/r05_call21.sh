#!/bin/bash
# round-5 GPU call 21: the one-off stall under torch.distributed, per-call host-side accounting
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/c21
mkdir -p $O
# K4 with the record entries requested inside the rows' round trip: tests that cover it, kernel time, timeline
timeout 1800 python -m pytest tests -q -m gpu -x -k "parity or batch or overlap or golden or fuzz or configs1 or reported" > $O/pytest_k4.log 2>&1; echo "rc $?" >> $O/pytest_k4.log; tail -n 3 $O/pytest_k4.log
R=$PWD
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/coldstats -- python $R/tools/k3_cold_probe.py > $R/$O/cold_probe.log 2>&1)
python3 tools/save_stats.py /tmp/coldstats $O/cold_kernel_stats.csv; grep icp_ $O/cold_kernel_stats.csv
timeout 300 python tools/timeline.py > $O/timeline_131k.txt 2>&1; grep "^K4" $O/timeline_131k.txt
SYNC_PROBE_CALLS=200 timeout 600 python tools/sync_probe.py > $O/sync_probe.json 2> $O/sync_probe.err; cat $O/sync_probe.json
for k in 1 2; do
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2954$k tools/torchrun_stall.py > $O/stall_torchrun_$k.jsonl 2> $O/stall_torchrun_$k.err
  cat $O/stall_torchrun_$k.jsonl | cut -c1-700
done
MH_OVERLAP=0 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29547 tools/torchrun_stall.py > $O/stall_torchrun_nooverlap.jsonl 2> $O/stall_torchrun_nooverlap.err
cut -c1-700 $O/stall_torchrun_nooverlap.jsonl
PLAIN=1 timeout 600 python tools/torchrun_stall.py > $O/stall_plain.jsonl 2> $O/stall_plain.err
cut -c1-400 $O/stall_plain.jsonl
