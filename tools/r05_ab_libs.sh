#!/usr/bin/env bash
# A/B of prebuilt variant libraries (tools/build_variants.py) against the product: tools/r05_ab_libs.sh <out> <name> [<name> ...]
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
out=gpurun_out/${1:-r05ab}; shift
mkdir -p "$out"
for v in "" "$@" ""; do
  lib=""; [ -n "$v" ] && lib="protocol_amd/variants/libpm_engine_$v.so"
  echo "=== variant '${v:-product}'" >> "$out/ab.log"
  PM_EXP_LIB=$lib timeout 120 python tools/variant_bench.py 1 20 >> "$out/ab.log" 2>&1
  PM_EXP_LIB=$lib timeout 120 python tools/variant_bench.py 2 6 >> "$out/ab.log" 2>&1
  PM_EXP_LIB=$lib timeout 120 python tools/churn_probe.py 8 2>&1 | grep "^tick" | awk '{s+=$3; n++} END {printf "churn ticks mean %.3f ms over %d\n", s/n, n}' >> "$out/ab.log"
done
grep -v "^  " "$out/ab.log"
