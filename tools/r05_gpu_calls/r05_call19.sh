#!/bin/bash
# round-5 GPU call 19: photometric chain with the frame copy and the write-back folded in; K3 wave-interleave variant re-measured
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$(pwd)
O=gpurun_out/c19
mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -q -m gpu -x -k "photo or replay or frontend or scan" > $O/pytest_photo.log 2>&1; echo "rc $?" >> $O/pytest_photo.log; tail -n 3 $O/pytest_photo.log
MH_PHOTO_UNFUSED=1 timeout 600 python -m pytest tests/test_gpu_photo.py -q -m gpu -x > $O/pytest_photo_unfused.log 2>&1; echo "rc $?" >> $O/pytest_photo_unfused.log; tail -n 2 $O/pytest_photo_unfused.log
timeout 300 python tools/photo_resident_time.py > $O/photo_resident.json 2> $O/photo_resident.err; cat $O/photo_resident.json
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_photo -- python $R/tools/photo_resident_time.py > /dev/null 2>&1)
python tools/save_stats.py /tmp/prof_photo $O/photo_kernel_stats.csv
head -12 $O/photo_kernel_stats.csv | cut -c1-160
# K3: base against the interleave variant, HIP events then rocprofv3 on cold calls
for v in base interleave; do
  if [ $v = base ]; then unset MH_LIB_OVERRIDE; else export MH_LIB_OVERRIDE=$R/mimosa_amd/lib/variants/$v.so; fi
  timeout 600 python tools/k3_time.py > $O/k3_time_$v.json 2> $O/k3_time_$v.err; cat $O/k3_time_$v.json
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/cold_$v -- python $R/tools/k3_cold_probe.py > $R/$O/cold_probe_$v.log 2>&1)
  python tools/save_stats.py /tmp/cold_$v $O/cold_kernel_stats_$v.csv
  grep icp_ $O/cold_kernel_stats_$v.csv
done
unset MH_LIB_OVERRIDE
