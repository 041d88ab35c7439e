#!/usr/bin/env bash
# anatomy of the streaming carve's rows (prebuilt PM_CARVE_PROF library): tools/r05_anat.sh <out-subdir>
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
out=gpurun_out/${1:-r05an}
mkdir -p "$out"
PM_PROF_NO_BUILD=1 timeout 200 python tools/stream_prof.py 100000 10000 > "$out/anatomy_10k.txt" 2>&1
PM_PROF_NO_BUILD=1 timeout 200 python tools/stream_prof.py 1000000 100000 > "$out/anatomy_100k.txt" 2>&1
grep "T=\|a row\|proposer rows\|chain anatomy\|compute\|networks\|bitmap sweeps" "$out/anatomy_10k.txt" "$out/anatomy_100k.txt"
