#!/usr/bin/env bash
# Round 5, second GPU call: the packed gather records + the reworked configuration boundaries against the build before them
# (variants/libpm_engine_base.so), the parity suite on the new build, and the anatomy of the new build.
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
out=gpurun_out/${1:-r05b}
mkdir -p "$out"
timeout 400 python -m pytest tests -m gpu -q -p no:cacheprovider -rfE -x > "$out/1_suite.log" 2>&1; echo "suite rc=$?" | tee -a "$out/1_suite.log"
for v in base ""; do
  lib=""; [ -n "$v" ] && lib="protocol_amd/variants/libpm_engine_$v.so"
  echo "=== variant '${v:-product}'" >> "$out/2_variants.log"
  PM_EXP_LIB=$lib timeout 120 python tools/variant_bench.py 1 16 >> "$out/2_variants.log" 2>&1
  PM_EXP_LIB=$lib timeout 120 python tools/variant_bench.py 2 6 >> "$out/2_variants.log" 2>&1
  PM_EXP_LIB=$lib timeout 120 python tools/churn_probe.py 8 >> "$out/2_variants.log" 2>&1
done
PM_PROF_NO_BUILD=1 timeout 200 python tools/stream_prof.py 100000 10000 > "$out/3_anatomy_10k.txt" 2>&1
PM_PROF_NO_BUILD=1 timeout 200 python tools/stream_prof.py 1000000 100000 > "$out/3_anatomy_100k.txt" 2>&1
PM_PROF_NO_BUILD=1 timeout 200 python tools/stream_trace.py 100000 10000 > "$out/4_timeline_10k.txt" 2>&1
PM_PROF_NO_BUILD=1 timeout 200 python tools/stream_trace.py churn > "$out/4_timeline_churn.txt" 2>&1
tail -3 "$out/1_suite.log"; grep -v "^  " "$out/2_variants.log"; head -16 "$out/3_anatomy_10k.txt"
