#!/usr/bin/env bash
# What to run on the GPU box first in a new round (through gpurun; ~4 minutes of box time):
#   1. the GPU suite as the driver runs it (the state the last round left verified);
#   2. the C++ plugin's GPU leg, which is opt-in until it has passed once (tests/test_gpu_plugin_cxx.py) — then drop its gate;
#   3. the GPU suite under 16 HIP hardware queues (tests/conftest.py: PM_TEST_HW_QUEUES) — the multi-rank and multi-pool
#      tests put up to 16 streams on the runtime's default 4 today;
#   4. the bench line.
# Results under gpurun_out/recheck/.
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
out=gpurun_out/recheck
mkdir -p "$out"
timeout 200 python -m pytest tests -m gpu -q -p no:cacheprovider -rfE > "$out/1_suite.log" 2>&1; echo "suite rc=$?" | tee -a "$out/1_suite.log"
PM_TEST_CXX_PLUGIN=1 timeout 120 python -m pytest tests/test_gpu_plugin_cxx.py -m gpu -q -p no:cacheprovider -rfE > "$out/2_cxx_plugin.log" 2>&1; echo "cxx plugin rc=$?" | tee -a "$out/2_cxx_plugin.log"
PM_TEST_HW_QUEUES=16 timeout 200 python -m pytest tests -m gpu -q -p no:cacheprovider -rfE > "$out/3_suite_q16.log" 2>&1; echo "suite q16 rc=$?" | tee -a "$out/3_suite_q16.log"
timeout 120 python bench.py > "$out/4_bench.json" 2> "$out/4_bench.err"; echo "bench rc=$?"
tail -3 "$out/1_suite.log" "$out/2_cxx_plugin.log" "$out/3_suite_q16.log"
