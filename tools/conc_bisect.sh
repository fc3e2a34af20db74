#!/bin/bash
# Bisect of bench.py's value_concurrent across the history.
#   tools/conc_bisect.sh export <sha>...   (here, no GPU) check the commits out under bisect_trees/<sha>/ and build their libraries
#   tools/conc_bisect.sh run [opts]        (GPU box)      run tools/conc_probe.py against every exported tree + the working tree
# bisect_trees/ is git-ignored but travels with the gpurun snapshot.
set -u
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
cd "$ROOT"
mode="$1"; shift
if [ "$mode" = export ]; then
  mkdir -p bisect_trees
  for sha in "$@"; do
    d="bisect_trees/$sha"
    if [ ! -d "$d" ]; then
      mkdir -p "$d"
      git archive "$sha" | tar -x -C "$d" --exclude='tests/golden' --exclude='profiles' --exclude='*.md' --exclude='BENCH_*' || exit 1
    fi
    ( cd "$d" && python -m mimosa_amd.build > build.log 2>&1 && ls -la mimosa_amd/lib/libmimosa_hip.so ) || { echo "build failed: $sha"; tail -5 "$d/build.log"; }
  done
elif [ "$mode" = run ]; then
  mkdir -p gpurun_out
  out="gpurun_out/conc_bisect.jsonl"
  : > "$out"
  for d in bisect_trees/*/; do
    sha="$(basename "$d")"
    timeout 300 python tools/conc_probe.py --tree "$d" --label "$sha" "$@" >> "$out" 2>> gpurun_out/conc_bisect.err || echo "{\"label\": \"$sha\", \"failed\": true}" >> "$out"
  done
  timeout 300 python tools/conc_probe.py --label HEAD "$@" >> "$out" 2>> gpurun_out/conc_bisect.err
  timeout 300 python tools/conc_probe.py --label HEAD-nocomp --no-components "$@" >> "$out" 2>> gpurun_out/conc_bisect.err
  cat "$out"
fi
