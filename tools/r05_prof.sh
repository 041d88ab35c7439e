#!/usr/bin/env bash
# anatomy of the streaming carve with the prebuilt PM_CARVE_PROF library: tools/r05_prof.sh <out-subdir> [suite]
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
out=gpurun_out/${1:-r05p}
mkdir -p "$out"
if [ "${2:-}" = "suite" ]; then
  timeout 400 python -m pytest tests -m gpu -q -p no:cacheprovider -rfE -x > "$out/1_suite.log" 2>&1; echo "suite rc=$?" | tee -a "$out/1_suite.log"; tail -3 "$out/1_suite.log"
fi
PM_PROF_NO_BUILD=1 timeout 200 python tools/stream_prof.py 100000 10000 > "$out/3_anatomy_10k.txt" 2>&1
PM_PROF_NO_BUILD=1 timeout 200 python tools/stream_prof.py 1000000 100000 > "$out/3_anatomy_100k.txt" 2>&1
timeout 120 python tools/variant_bench.py 1 16 2>&1 | grep -v "^  " > "$out/2_bench.txt"
timeout 120 python tools/variant_bench.py 2 6 2>&1 | grep -v "^  " >> "$out/2_bench.txt"
cat "$out/3_anatomy_10k.txt" "$out/3_anatomy_100k.txt" "$out/2_bench.txt"
