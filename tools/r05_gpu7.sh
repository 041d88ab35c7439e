#!/usr/bin/env bash
# the whole GPU suite on the product build, then its timings (cold matches of configs[1] / [2], churn ticks): tools/r05_gpu7.sh <out-subdir>
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
out=gpurun_out/${1:-r05s}
mkdir -p "$out"
timeout 500 python -m pytest tests -m gpu -q -p no:cacheprovider -rfE -x > "$out/1_suite.log" 2>&1; echo "suite rc=$?" | tee -a "$out/1_suite.log"
tail -5 "$out/1_suite.log"
for r in 1 2; do
  timeout 120 python tools/variant_bench.py 1 20 2>&1 | grep -v "^  " | tee -a "$out/2_timing.log"
  timeout 120 python tools/variant_bench.py 2 8 2>&1 | grep -v "^  " | tee -a "$out/2_timing.log"
done
timeout 120 python tools/churn_probe.py 8 2>&1 | tail -12 | tee -a "$out/2_timing.log"
