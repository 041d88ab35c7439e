#!/usr/bin/env bash
# the whole GPU suite on the product build, then the default bench line and the churn ticks with their host marks:
# tools/r05_gpu11.sh <out-subdir>
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
out=gpurun_out/${1:-r05s}
mkdir -p "$out"
timeout 500 python -m pytest tests -m gpu -q -p no:cacheprovider -rfE -x > "$out/1_suite.log" 2>&1; echo "suite rc=$?" | tee -a "$out/1_suite.log"
tail -3 "$out/1_suite.log"
timeout 400 python bench.py > "$out/bench.json" 2> "$out/bench.err"; echo "bench rc=$?"
PM_TRACE_HOST=1 timeout 120 python tools/churn_probe.py 8 > "$out/churn.txt" 2>&1
grep "^tick\|^cold" "$out/churn.txt"
python - "$out/bench.json" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print("ms_per_step", d["ms_per_step"], "churn", d["churn"]["ms_per_tick"], d["churn"]["split_ms_p50"], "cfg2", d["configs2"]["ms_per_match"], "merge", d["merge"]["ms_p50"])
PY
