#!/usr/bin/env bash
# suite + the N = 4 plumbing tick (four processes over gloo sharing the one GPU): tools/r05_gpu5.sh <out-subdir>
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
out=gpurun_out/${1:-r05i}
mkdir -p "$out"
timeout 500 python -m pytest tests -m gpu -q -p no:cacheprovider -rfE -x > "$out/1_suite.log" 2>&1; echo "suite rc=$?" | tee -a "$out/1_suite.log"
PM_STREAM_WGS=60 PM_BENCH_BACKEND=gloo PM_BENCH_SHARE_DEVICE=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 \
  --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 4 --steps 5 --warmup 2 2> "$out/n4.err" | grep '^{' > "$out/bench_n4_gloo.json"
tail -4 "$out/1_suite.log"
python - "$out/bench_n4_gloo.json" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print("n4: ms_per_step", d["ms_per_step"], "dist", json.dumps(d.get("dist"))[:900])
except Exception as ex:
    print("n4 line missing:", ex)
PY
tail -5 "$out/n4.err"
