#!/bin/bash
# round-5 GPU call 22: (a) the torch.distributed stall: which part of the slow mh_icp_linearize_async (MH_WAIT_TRACE reports it);
# (b) work sharing inside a workgroup (-DMH_SHARE=1), re-measured now that every workgroup holds a mix of light and heavy waves
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD
O=gpurun_out/c22
mkdir -p $O
export TMPDIR=/tmp
MH_WAIT_TRACE=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29551 tools/torchrun_stall.py > $O/stall_trace.jsonl 2> $O/stall_trace.err
grep "MH_WAIT_TRACE" $O/stall_trace.err | head -5
# the same under rocprofv3 --hip-trace: the HIP calls of the slow enqueue
(cd /tmp && MH_WAIT_TRACE=1 timeout 600 rocprofv3 --hip-trace --output-format csv -d /tmp/stall_hip -- python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29552 $R/tools/torchrun_stall.py > $R/$O/stall_hiptrace.jsonl 2> $R/$O/stall_hiptrace.err)
python3 - <<'PY'
import csv, glob
rows = []
for f in glob.glob("/tmp/stall_hip/**/*hip_api_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        try:
            rows.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Function"], r.get("Process_Id", ""), r.get("Thread_Id", ""), int(r["Start_Timestamp"])))
        except (KeyError, ValueError):
            pass
rows.sort(reverse=True)
print("longest HIP API calls (ns, function, pid, tid):")
for r in rows[:12]:
    print("  ", r[:4])
PY
grep "MH_WAIT_TRACE" $O/stall_hiptrace.err | head -5
for v in base share share_t2 share_h1 share_t8; do
  if [ $v = base ]; then unset MH_LIB_OVERRIDE; else export MH_LIB_OVERRIDE=$R/mimosa_amd/lib/variants/$v.so; fi
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/cold_$v -- python $R/tools/k3_cold_probe.py > $R/$O/cold_probe_$v.log 2>&1)
  python3 tools/save_stats.py /tmp/cold_$v $O/cold_kernel_stats_$v.csv
  echo "== $v"; grep icp_linearize $O/cold_kernel_stats_$v.csv
done
export MH_LIB_OVERRIDE=$R/mimosa_amd/lib/variants/share.so
timeout 1200 python -m pytest tests -q -m gpu -x -k "parity or batch or golden or fuzz or configs1" > $O/pytest_share.log 2>&1; echo "rc $?" >> $O/pytest_share.log; tail -n 3 $O/pytest_share.log
unset MH_LIB_OVERRIDE
