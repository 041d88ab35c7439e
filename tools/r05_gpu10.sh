#!/usr/bin/env bash
# parity subset on the product build, then tools/r05_ab_libs.sh (configs[1], configs[2], churn ticks; product first and last):
# tools/r05_gpu10.sh <out-subdir> [variant...]
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
out=gpurun_out/${1:-r05p}
mkdir -p "$out"
timeout 600 python -m pytest tests/test_gpu_row_networks.py tests/test_gpu_scale.py tests/test_gpu_golden_churn.py tests/test_gpu_parity.py -q -p no:cacheprovider -rfE -x > "$out/1_parity.log" 2>&1; echo "parity rc=$?" | tee -a "$out/1_parity.log"
tail -4 "$out/1_parity.log"
bash tools/r05_ab_libs.sh "$@"
