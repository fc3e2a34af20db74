#!/bin/bash
# round-5 GPU call 33: full GPU suite after the ABI pruning; round profile (kernel stats with / without the component server, PMC);
# cold-call kernel stats at 131 072 / 24 576 points; timelines; bench with the driver's arguments and with the defaults
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD
O=gpurun_out/c33
mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -n 4 $O/pytest.log
MH_ROUND=r05 MH_COMMIT=28f4b24 timeout 3000 bash tools/profile_round.sh > $O/profile_round.log 2>&1
cp -r gpurun_out/prof $O/prof 2>/dev/null
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/coldstats -- python $R/tools/k3_cold_probe.py > $R/$O/cold_probe.log 2>&1
python3 $R/tools/save_stats.py /tmp/coldstats $R/$O/cold_kernel_stats.csv
cd $R
timeout 300 python tools/timeline.py > $O/timeline_131k.txt 2>&1
MH_TL_POINTS=24576 timeout 300 python tools/timeline.py > $O/timeline_24k.txt 2>&1
SYNC_PROBE_CALLS=200 timeout 600 python tools/sync_probe.py > $O/sync_probe.json 2> $O/sync_probe.err
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_driver_args.json 2> $O/bench_driver_args.err
timeout 1200 python bench.py > $O/bench_default.json 2> $O/bench_default.err
grep icp_ $O/cold_kernel_stats.csv
ls $O
# photometric: resident preprocess
timeout 300 python tools/photo_resident_time.py > $O/photo_resident.json 2> $O/photo_resident.err; cat $O/photo_resident.json
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_photo -- python $R/tools/photo_resident_time.py > /dev/null 2>&1)
python3 tools/save_stats.py /tmp/prof_photo $O/photo_kernel_stats.csv
python3 - <<'PY'
import json
for f in ("bench_driver_args", "bench_default"):
    try:
        d = json.loads(open(f"gpurun_out/c33/{f}.json").read().strip().splitlines()[-1])
        print(f, {k: d.get(k) for k in ("value", "ms_per_step", "value_sync", "sync_latency_ms", "sync_latency_without_components_ms", "sync_latency_through_ctypes_ms")})
        print("   photometric.resident", d.get("photometric", {}).get("resident"))
    except Exception as e:
        print(f, "unreadable:", e)
PY
cat $O/sync_probe.json
for p in 8 5; do
(cd /tmp && rm -rf /tmp/ptrace && MH_PATCH=$p timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ptrace -- python $R/tools/photo_trace.py > $R/$O/photo_trace_$p.log 2>&1)
python3 tools/save_stats.py /tmp/ptrace $O/photo_trace_kernel_stats_$p.csv; grep "photo_linearize" $O/photo_trace_kernel_stats_$p.csv
done
