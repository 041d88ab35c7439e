#!/usr/bin/env bash
# timelines + anatomy of the prebuilt PM_CARVE_PROF library under environment settings: tools/r05_trace.sh <out> "<ENV=a>" ...
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
out=gpurun_out/${1:-r05t}; shift
mkdir -p "$out"
i=0
for envs in "$@"; do
  i=$((i+1))
  echo "=== $envs" > "$out/trace_$i.txt"
  env $envs PM_PROF_NO_BUILD=1 timeout 200 python tools/stream_trace.py 100000 10000 >> "$out/trace_$i.txt" 2>&1
  env $envs PM_PROF_NO_BUILD=1 timeout 200 python tools/stream_prof.py 100000 10000 >> "$out/trace_$i.txt" 2>&1
  head -60 "$out/trace_$i.txt"
done
