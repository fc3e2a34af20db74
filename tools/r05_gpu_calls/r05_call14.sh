#!/bin/bash
# round-5 GPU call 14: the one-off 35-48 ms stall of a pipelined burst under torch.distributed — kernel + HIP API trace of tools/torchrun_probe.py
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD
O=gpurun_out/c14
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29571 $R/tools/torchrun_probe.py > $R/$O/probe_plain.json 2> $R/$O/probe_plain.err
cat $R/$O/probe_plain.json
MH_OVERLAP=0 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29572 $R/tools/torchrun_probe.py > $R/$O/probe_nooverlap.json 2> $R/$O/probe_nooverlap.err
cat $R/$O/probe_nooverlap.json
timeout 900 rocprofv3 --hip-trace --kernel-trace --output-format csv -d /tmp/trp -- python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29573 $R/tools/torchrun_probe.py > $R/$O/probe_traced.json 2> $R/$O/probe_traced.err
cat $R/$O/probe_traced.json
python3 - <<'PY'
import csv,glob,os
ks=[]
for f in glob.glob('/tmp/trp/**/*_kernel_trace.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        ks.append((int(r['Start_Timestamp']),int(r['End_Timestamp']),r['Kernel_Name'][:50],r.get('Stream_Id'),r.get('Queue_Id')))
ks.sort()
print('kernels',len(ks))
# largest gaps between consecutive kernel ends/starts after the first linearize kernel
first=[i for i,k in enumerate(ks) if 'icp_linearize' in k[2]]
if first:
    i0=first[0]
    gaps=sorted(((ks[i+1][0]-max(k[1] for k in ks[max(i0,i-3):i+1]),i) for i in range(i0,len(ks)-1)),reverse=True)[:6]
    for g,i in gaps:
        print('gap %.2f ms after'%(g/1e6), ks[i][2], 'stream',ks[i][3],'queue',ks[i][4], '-> next', ks[i+1][2], 'stream',ks[i+1][3],'queue',ks[i+1][4])
    longk=sorted(ks[i0:],key=lambda k:k[1]-k[0],reverse=True)[:6]
    for k in longk: print('long kernel %.2f ms'%((k[1]-k[0])/1e6), k[2], 'stream',k[3],'queue',k[4])
api=[]
for f in glob.glob('/tmp/trp/**/*_hip_api_trace.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        d=int(r['End_Timestamp'])-int(r['Start_Timestamp'])
        if d>5e6: api.append((d/1e6,r['Function'],int(r['Start_Timestamp'])))
api.sort(reverse=True)
t_first=ks[first[0]][0] if first else 0
for a in api[:12]: print('api %.1f ms'%a[0], a[1], 'at %.1f ms after the first linearize kernel'%((a[2]-t_first)/1e6))
PY
