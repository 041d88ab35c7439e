#!/usr/bin/env bash
# Round 5, first GPU call: everything in the tree gets GPU evidence, then the experiments.
#   1 the GPU suite as the driver runs it (with the C++ plugin's leg ungated and the forced-abort tests);
#   2 the same under the shipped runtime configuration (16 hardware queues);
#   3 the bench line;
#   4 the PM_EXP_* row-supply experiments: timing (cold match configs[1], churn ticks), then the parity suite on all three together.
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
out=gpurun_out/r05a
mkdir -p "$out"
timeout 400 python -m pytest tests -m gpu -q -p no:cacheprovider -rfEs -x > "$out/1_suite.log" 2>&1; echo "suite rc=$?" | tee -a "$out/1_suite.log"
PM_TEST_HW_QUEUES=16 timeout 400 python -m pytest tests -m gpu -q -p no:cacheprovider -rfE > "$out/2_suite_q16.log" 2>&1; echo "suite q16 rc=$?" | tee -a "$out/2_suite_q16.log"
timeout 300 python bench.py > "$out/3_bench.json" 2> "$out/3_bench.err"; echo "bench rc=$?"
for v in "" ctz skip cold all3; do
  lib=""; [ -n "$v" ] && lib="protocol_amd/variants/libpm_engine_$v.so"
  echo "=== variant '${v:-product}'" >> "$out/4_variants.log"
  PM_EXP_LIB=$lib timeout 120 python tools/variant_bench.py 1 12 >> "$out/4_variants.log" 2>&1
  PM_EXP_LIB=$lib timeout 120 python tools/churn_probe.py 8 >> "$out/4_variants.log" 2>&1
done
PM_EXP_LIB=protocol_amd/variants/libpm_engine_all3.so timeout 400 python -m pytest tests -m gpu -q -p no:cacheprovider -rfE > "$out/5_suite_all3.log" 2>&1; echo "suite all3 rc=$?" | tee -a "$out/5_suite_all3.log"
tail -4 "$out/1_suite.log" "$out/2_suite_q16.log" "$out/5_suite_all3.log"; cat "$out/4_variants.log" | grep -v "^  "; tail -c 600 "$out/3_bench.json"
