#!/usr/bin/env bash
# a kernel change through the GPU in one call: parity subset on the product build, then timings of prebuilt variants
# against it (ROUNDS rounds, the product first in each), then the product-like timeline (PM_ROW_REC variant):
# tools/r05_gpu6.sh <out-subdir> [variant...]
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
out=gpurun_out/${1:-r05q}; shift
mkdir -p "$out"
timeout 600 python -m pytest tests/test_gpu_row_networks.py tests/test_gpu_scale.py tests/test_gpu_golden_churn.py tests/test_gpu_parity.py -q -p no:cacheprovider -rfE -x > "$out/1_parity.log" 2>&1; echo "parity rc=$?" | tee -a "$out/1_parity.log"
tail -4 "$out/1_parity.log"
for round in $(seq 1 ${ROUNDS:-2}); do
for v in "" "$@"; do
  lib=""; [ -n "$v" ] && lib="protocol_amd/variants/libpm_engine_$v.so"
  echo "=== variant '${v:-product}'" >> "$out/2_variants.log"
  PM_EXP_LIB=$lib timeout 120 python tools/variant_bench.py 1 20 >> "$out/2_variants.log" 2>&1
  PM_EXP_LIB=$lib timeout 120 python tools/variant_bench.py 2 8 >> "$out/2_variants.log" 2>&1
done
done
grep -v "^  " "$out/2_variants.log" | sed 's/defines .*: carve/carve/; s/, groups.*//' | paste - - -
L=protocol_amd/variants/libpm_engine_rowrec.so
if [ -f $L ]; then
  PM_EXP_LIB=$L timeout 200 python tools/row_rec.py 100000 10000 > "$out/rows_10k.txt" 2>&1
  PM_PROF_LIB=$L PM_PROF_NO_BUILD=1 timeout 200 python tools/stream_trace.py 100000 10000 --dump "$out/events_10k.txt" > "$out/timeline_10k.txt" 2>&1
  PM_PROF_LIB=$L PM_PROF_NO_BUILD=1 timeout 200 python tools/stream_trace.py 1000000 100000 > "$out/timeline_100k.txt" 2>&1
  head -3 "$out/rows_10k.txt"; head -34 "$out/timeline_10k.txt"; grep "chain waits\|totals" "$out/timeline_100k.txt"
fi
