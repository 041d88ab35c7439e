#!/usr/bin/env bash
# timings of the product (and prebuilt variants) + the product-like timeline with its event dump, no parity run:
# tools/r05_gpu8.sh <out-subdir> [variant...]
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
out=gpurun_out/${1:-r05d}; shift
mkdir -p "$out"
for round in $(seq 1 ${ROUNDS:-2}); do
for v in "" "$@"; do
  lib=""; [ -n "$v" ] && lib="protocol_amd/variants/libpm_engine_$v.so"
  echo "=== variant '${v:-product}'" >> "$out/2_variants.log"
  for cf in ${CFGS:-1 2}; do
    n=20; [ "$cf" = 2 ] && n=8
    PM_EXP_LIB=$lib timeout 120 python tools/variant_bench.py $cf $n >> "$out/2_variants.log" 2>&1
  done
done
done
grep -v "^  " "$out/2_variants.log" | sed 's/defines .*: carve/carve/; s/, groups.*//'
L=protocol_amd/variants/libpm_engine_rowrec.so
if [ -f $L ]; then
  PM_PROF_LIB=$L PM_PROF_NO_BUILD=1 timeout 200 python tools/stream_trace.py 100000 10000 --dump "$out/events_10k.txt" > "$out/timeline_10k.txt" 2>&1
  head -34 "$out/timeline_10k.txt"
fi
