#!/bin/bash
# round-5 GPU call 39: the two bench lines at the final build
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/c39
mkdir -p $O
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_driver_args.json 2> $O/bench_driver_args.err
timeout 1200 python bench.py > $O/bench_default.json 2> $O/bench_default.err
python3 - <<'PY'
import json
for f in ("bench_driver_args", "bench_default"):
    try:
        d = json.loads(open(f"gpurun_out/c39/{f}.json").read().strip().splitlines()[-1])
        print(f, {k: d.get(k) for k in ("value", "ms_per_step", "value_sync", "sync_latency_ms", "sync_latency_without_components_ms")})
        ph = d.get("photometric", {})
        print("   photometric.resident", ph.get("resident", {}).get("preprocess_scan_ms"), {k: (v.get("factor_kernel_ms"), v.get("factor_linearize_sync_ms"), v.get("preprocess_ms_host_buffers")) for k, v in ph.items() if isinstance(v, dict) and "factor_kernel_ms" in v})
        print("   roofline traffic", d["roofline"].get("traffic"), d["roofline"].get("frac"), d["roofline"].get("valu_issue_frac"))
    except Exception as e:
        print(f, "unreadable:", e)
PY
