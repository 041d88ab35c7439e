#!/usr/bin/env bash
# suite + timings + host marks of the current build against variants/libpm_engine_base.so: tools/r05_gpu3.sh <out-subdir>
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
out=gpurun_out/${1:-r05d}
mkdir -p "$out"
timeout 500 python -m pytest tests -m gpu -q -p no:cacheprovider -rfE -x > "$out/1_suite.log" 2>&1; echo "suite rc=$?" | tee -a "$out/1_suite.log"
for v in base ""; do
  lib=""; [ -n "$v" ] && lib="protocol_amd/variants/libpm_engine_$v.so"
  echo "=== variant '${v:-product}'" >> "$out/2_variants.log"
  PM_EXP_LIB=$lib timeout 120 python tools/variant_bench.py 1 16 >> "$out/2_variants.log" 2>&1
  PM_EXP_LIB=$lib timeout 120 python tools/variant_bench.py 2 6 >> "$out/2_variants.log" 2>&1
  PM_EXP_LIB=$lib timeout 120 python tools/churn_probe.py 8 >> "$out/2_variants.log" 2>&1
done
PM_TRACE_HOST=1 timeout 120 python tools/host_trace.py 1 > "$out/host_marks.txt" 2>&1
timeout 200 python bench.py --no-cpu-baseline --no-extras > "$out/3_bench_line.json" 2> "$out/3_bench.err"
tail -5 "$out/1_suite.log"; grep -v "^  " "$out/2_variants.log"; tail -14 "$out/host_marks.txt"; python - <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1] if len(sys.argv)>1 else "/dev/null"))
except Exception as ex:
    d=None
PY
python -c "
import json
d=json.load(open('$out/3_bench_line.json'))
print('bench: ms_per_step', d['ms_per_step'], 'p50', d.get('p50_match_latency_ms'), d.get('phase_ms_p50'))
"
