#!/bin/bash
# round-5 GPU call 17: closing soaks under the allocation checker
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/c17
mkdir -p $O
# first: the bench lines with the sync call timed around the foreign call alone (arguments marshalled once)
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_driver_args.json 2> $O/bench_driver_args.err; tail -c 400 $O/bench_driver_args.err
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
python - <<'PY'
import json
for f in ("bench_driver_args", "bench_default"):
    try:
        d = json.loads(open(f"gpurun_out/c17/{f}.json").read().strip().splitlines()[-1])
        print(f, {k: d.get(k) for k in ("value", "ms_per_step", "value_sync", "sync_latency_ms", "sync_latency_without_components_ms")})
    except Exception as e:
        print(f, "unreadable:", e)
PY
export MH_ALLOC_CHECK=1
for i in 1 2; do
  timeout 2400 python -m pytest tests -q -m gpu > $O/full_$i.log 2>&1; echo "rc $?" >> $O/full_$i.log; tail -n 2 $O/full_$i.log
done
for i in 1 2 3 4; do
  MH_FUZZ_EXTRA=300 timeout 2400 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_batch.py tests/test_gpu_shard_native.py tests/test_gpu_overlap.py tests/test_zz_alloc_check.py -q -m gpu > $O/fuzz_$i.log 2>&1; echo "rc $?" >> $O/fuzz_$i.log; tail -n 2 $O/fuzz_$i.log
done
timeout 1200 python tools/replay_soak.py 12 > $O/replay_soak.log 2>&1; tail -n 3 $O/replay_soak.log
