#!/usr/bin/env bash
# suite, then A/B of an environment knob on the product build: tools/r05_gpu4.sh <out-subdir> "<ENV=a>" "<ENV=b>" ...
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
out=gpurun_out/${1:-r05e}; shift
mkdir -p "$out"
timeout 500 python -m pytest tests -m gpu -q -p no:cacheprovider -rfE -x > "$out/1_suite.log" 2>&1; echo "suite rc=$?" | tee -a "$out/1_suite.log"
for envs in "$@"; do
  echo "=== $envs" >> "$out/2_ab.log"
  env $envs timeout 120 python tools/variant_bench.py 1 16 >> "$out/2_ab.log" 2>&1
  env $envs timeout 120 python tools/variant_bench.py 2 6 >> "$out/2_ab.log" 2>&1
  env $envs timeout 120 python tools/churn_probe.py 8 >> "$out/2_ab.log" 2>&1
done
tail -5 "$out/1_suite.log"; grep -v "^  " "$out/2_ab.log"
