#!/bin/bash
# round-5 GPU call 1: baseline — sync latency probe, K3/K4 event times, per-phase timelines (K3 + K4) at 131 072 / 24 576 points, API trace
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/c1
mkdir -p $O
timeout 600 python tools/sync_probe.py > $O/sync_probe.json 2> $O/sync_probe.err
timeout 300 python tools/k3_time.py > $O/k3_time_base.json 2> $O/k3_time_base.err
MH_LIB_OVERRIDE=$PWD/mimosa_amd/lib/variants/interleave.so timeout 300 python tools/k3_time.py > $O/k3_time_interleave.json 2> $O/k3_time_interleave.err
timeout 300 python tools/timeline.py > $O/timeline_131k.txt 2>&1
MH_TL_POINTS=24576 timeout 300 python tools/timeline.py > $O/timeline_24k.txt 2>&1
R=$PWD
cd /tmp && export TMPDIR=/tmp
SYNC_PROBE_CALLS=40 timeout 600 rocprofv3 --hip-trace --kernel-trace --output-format csv -d /tmp/trace1 -- python $R/tools/sync_probe.py > $R/$O/trace.log 2>&1
for f in $(ls /tmp/trace1/*/*_hip_api_trace.csv /tmp/trace1/*/*_kernel_trace.csv 2>/dev/null); do cp $f $R/$O/; done
ls -la $R/$O
