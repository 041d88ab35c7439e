#!/usr/bin/env bash
# carve_stream_kernel's average duration under rocprofv3 for prebuilt variant libraries and the product, same box
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
out=$PWD/gpurun_out/${1:-r05ks}; shift
mkdir -p "$out"
export TMPDIR=/tmp
for v in "$@" "" "$@" ""; do
  lib=""; [ -n "$v" ] && lib="$GRAFT_REPO_ROOT/protocol_amd/variants/libpm_engine_$v.so"
  d="$out/tr_${v:-product}"
  (cd /tmp && PM_EXP_LIB=$lib timeout -k 5 200 rocprofv3 --kernel-trace --stats -d "$d" -o t -- python "$GRAFT_REPO_ROOT/tools/variant_bench.py" 1 20 > "$out/log_${v:-product}.txt" 2>&1)
  echo "=== ${v:-product}" >> "$out/kstat.txt"
  python tools/rocpd_summary.py "$d/t_results.db" "$out/k.csv" > /dev/null 2>&1; grep "carve_stream_kernel\|carve_finish\|elig_place" "$out/k.csv" >> "$out/kstat.txt"
  rm -rf "$d"
done
cat "$out/kstat.txt"
