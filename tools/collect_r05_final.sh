#!/bin/bash
# The round's final profile set on the GPU box (from the repository root): tools/collect_profiles.py r05 (kernel statistics of
# configs[1] / configs[2] / the churn stream, PMC traffic of configs[1], the default bench line), then the streaming carve's
# anatomy (prebuilt PM_CARVE_PROF library), its product-like timelines (prebuilt PM_ROW_REC library) and the host marks.
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out/r05
timeout 900 python tools/collect_profiles.py r05 > gpurun_out/r05/collect.log 2>&1; echo "collect rc=$?"
tail -3 gpurun_out/r05/collect.log | cut -c1-600
PM_PROF_NO_BUILD=1 timeout 200 python tools/stream_prof.py 100000 10000 > gpurun_out/r05/r05_stream_anatomy_10k.txt 2>&1
PM_PROF_NO_BUILD=1 timeout 200 python tools/stream_prof.py 1000000 100000 > gpurun_out/r05/r05_stream_anatomy_100k.txt 2>&1
L=protocol_amd/variants/libpm_engine_rowrec.so
PM_PROF_LIB=$L PM_PROF_NO_BUILD=1 timeout 200 python tools/stream_trace.py 100000 10000 > gpurun_out/r05/r05_stream_timeline_10k.txt 2>&1
PM_PROF_LIB=$L PM_PROF_NO_BUILD=1 timeout 200 python tools/stream_trace.py 1000000 100000 > gpurun_out/r05/r05_stream_timeline_100k.txt 2>&1
PM_PROF_LIB=$L PM_PROF_NO_BUILD=1 timeout 200 python tools/stream_trace.py churn > gpurun_out/r05/r05_stream_timeline_churn.txt 2>&1
PM_TRACE_HOST=1 timeout 120 python tools/host_trace.py 1 2>&1 | tail -16 > gpurun_out/r05/r05_host_marks_match.txt
PM_TRACE_HOST=1 timeout 120 python tools/churn_probe.py 8 2>&1 | tail -40 > gpurun_out/r05/r05_host_marks_churn.txt
head -3 gpurun_out/r05/r05_stream_anatomy_10k.txt; head -2 gpurun_out/r05/r05_stream_timeline_churn.txt
