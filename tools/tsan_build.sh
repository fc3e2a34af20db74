#!/bin/bash
# ThreadSanitizer pass over the HOST side of the library: every .hip file's host code compiled with
# -fsanitize=thread (hipcc ignores the flag for the device code), the drivers of the host mirror with the same clang.
#   tools/tsan_build.sh          (here, no GPU)   -> mimosa_amd/lib/tsan/{libmimosa_hip.so, replay_native, sharded_pipeline}
#   tools/tsan_build.sh run      (GPU box)        -> gpurun_out/tsan_*.log   (take mimosa_amd/lib/tsan/ out of .gpurunignore first)
# What TSAN cannot see: the HIP runtime is not instrumented, so ordering established ONLY through it (a stream
# synchronisation between two host threads) is invisible, and device writes into mapped host memory are not events at all.
set -u
R=$(cd "$(dirname "$0")/.." && pwd)
D=$R/mimosa_amd/lib/tsan
CXX=/opt/rocm/lib/llvm/bin/clang++
if [ "${1:-build}" = build ]; then
  mkdir -p $D $R/mimosa_amd/build_tsan
  objs=""
  for f in icp_kernels deskew_kernels order_kernels scan_kernels mh_api map_kernels map_api shard_kernels shard_api photo_kernels photo_api; do
    extra=""
    case $f in deskew_kernels|scan_kernels|map_kernels|photo_kernels) extra="-ffp-contract=off";; esac
    o=$R/mimosa_amd/build_tsan/$f.o
    if [ ! -f $o ] || [ $R/mimosa_amd/csrc/$f.hip -nt $o ]; then
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -DMH_BUILDING_LIBRARY -Wno-unused-parameter -Wno-option-ignored -fsanitize=thread $extra -c $R/mimosa_amd/csrc/$f.hip -o $o &
    fi
    objs="$objs $o"
  done
  wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fsanitize=thread -o $D/libmimosa_hip.so $objs -ldl || exit 1
  H=$R/mimosa_amd/host
  $CXX -std=c++17 -O1 -g -fsanitize=thread -I $R/mimosa_amd -I $H -I $H/gtsam_sig $H/replay_main.cpp -o $D/replay_native -L $D -lmimosa_hip -lpthread -Wl,-rpath,'$ORIGIN' || exit 1
  $CXX -std=c++17 -O1 -g -fsanitize=thread -I $R -I $H/gtsam_sig $R/tests/cpp/sharded_pipeline.cpp -o $D/sharded_pipeline -L $D -lmimosa_hip -lpthread -Wl,-rpath,'$ORIGIN' || exit 1
  ls -la $D
else
  cd $R
  export TSAN_OPTIONS="halt_on_error=0 second_deadlock_stack=1 history_size=4 report_signal_unsafe=0"
  python - <<'PY'
import json, os, re, sys, tempfile, subprocess
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from mimosa_amd import replay
D = os.path.join(os.getcwd(), "mimosa_amd", "lib", "tsan")
SUPP = os.path.join(os.getcwd(), "tools", "tsan.supp")
RUNTIME = ("libamdhip64", "libhsa-runtime", "librocprofiler", "libdrm", "libamd_comgr")


def classify(stderr):
    """reports, and those with at least one racing access performed by OUR code (the first frame below the sanitizer's
    own interceptor is not inside the uninstrumented HIP/HSA runtime)"""
    reps = [r for r in stderr.split("==================") if "WARNING: ThreadSanitizer" in r]
    ours = []
    for r in reps:
        for b in re.split(r"\n\s*\n", r):
            if not re.match(r"\s*(WARNING|Write|Read|Previous|Atomic)", b):
                continue
            for fr in re.findall(r"#\d+ (.*)", b):
                if "compiler-rt/lib/tsan" in fr:
                    continue
                if not any(k in fr for k in RUNTIME):
                    ours.append(r)
                break
    return len(reps), ours


def run(tag, cmd):
    res = {}
    for mode in ("raw", "suppressed"):
        env = dict(os.environ)
        if mode == "suppressed":
            env["TSAN_OPTIONS"] = env["TSAN_OPTIONS"] + " suppressions=" + SUPP
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=1500, env=env)
        n, ours = classify(out.stderr)
        res[mode] = {"rc": out.returncode, "reports": n, "reports_with_an_access_in_our_code": len(ours)}
        if mode == "raw":
            open(f"gpurun_out/tsan_{tag}.log", "w").write(out.stderr[-150000:])
        if ours:
            open(f"gpurun_out/tsan_{tag}_{mode}_ours.log", "w").write("\n==================\n".join(ours)[:400000])
        res[mode]["stdout"] = out.stdout.strip()[-200:]
    print(tag, json.dumps(res), flush=True)
    return res


summary = {}
cfg = replay.ReplayConfig(n_scans=8, rows=128)
scans = replay.make_scans(cfg)
with tempfile.TemporaryDirectory() as td:
    path = os.path.join(td, "replay_input.bin")
    replay.write_native_input(path, cfg, scans, 7)
    summary["replay_pipelined"] = run("replay_pipelined", [os.path.join(D, "replay_native"), path, "1"])
    summary["replay_sequential"] = run("replay_seq", [os.path.join(D, "replay_native"), path, "1", "sequential"])
    import test_gpu_host_cpp as T
    inp = os.path.join(td, "sh.bin")
    T._write_sharded_input(inp)
    for w in ("2", "3"):
        summary["sharded_local" + w] = run("sharded_local" + w, [os.path.join(D, "sharded_pipeline"), inp, "local", w])
json.dump(summary, open("gpurun_out/tsan_summary.json", "w"), indent=1)
PY
fi
