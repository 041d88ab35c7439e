#!/usr/bin/env bash
# timings of variant libraries against the product library, twice round: tools/r05_var.sh <out-subdir> variant...
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
out=gpurun_out/${1:-r05v}; shift
mkdir -p "$out"
for round in 1 2; do
for v in "" "$@"; do
  lib=""; [ -n "$v" ] && lib="protocol_amd/variants/libpm_engine_$v.so"
  echo "=== variant '${v:-product}'" >> "$out/variants.log"
  PM_EXP_LIB=$lib timeout 120 python tools/variant_bench.py 1 20 >> "$out/variants.log" 2>&1
  PM_EXP_LIB=$lib timeout 120 python tools/variant_bench.py 2 8 >> "$out/variants.log" 2>&1
done
done
grep -v "^  " "$out/variants.log" | sed 's/defines .*: carve/carve/; s/, groups.*//' | paste - - -
