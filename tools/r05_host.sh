#!/usr/bin/env bash
# host marks + kernel timeline of a cold match (configs[1]): tools/r05_host.sh <out-subdir>
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
out=$PWD/gpurun_out/${1:-r05h}
mkdir -p "$out"
PM_TRACE_HOST=1 timeout 120 python tools/host_trace.py 1 > "$out/host_marks.txt" 2>&1
cd /tmp && export TMPDIR=/tmp
timeout -k 5 200 rocprofv3 --kernel-trace -d "$out/trace" -o t -- python "$GRAFT_REPO_ROOT/tools/host_trace.py" 1 > "$out/trace.log" 2>&1
python "$GRAFT_REPO_ROOT/tools/tick_timeline.py" "$out/trace/"*"/t_results.db" --all > "$out/timeline.txt" 2>&1 || python "$GRAFT_REPO_ROOT/tools/tick_timeline.py" "$out/trace/t_results.db" --all > "$out/timeline.txt" 2>&1
rm -rf "$out/trace"
tail -30 "$out/host_marks.txt"; cat "$out/timeline.txt"
