#!/bin/bash
# What profiles/r05_* holds beyond tools/collect_profiles.py's set (run on the GPU box from the repository root after
# `python tools/collect_profiles.py r05`; outputs under gpurun_out/r05/): anatomy and timelines of the streaming carve
# (prebuilt PM_CARVE_PROF library), host marks of a match and of churn ticks.
set -u
mkdir -p gpurun_out/r05
PM_PROF_NO_BUILD=1 timeout 200 python tools/stream_prof.py 100000 10000 > gpurun_out/r05/r05_stream_anatomy_10k.txt 2>&1
PM_PROF_NO_BUILD=1 timeout 200 python tools/stream_prof.py 1000000 100000 > gpurun_out/r05/r05_stream_anatomy_100k.txt 2>&1
PM_PROF_NO_BUILD=1 timeout 200 python tools/stream_trace.py 100000 10000 > gpurun_out/r05/r05_stream_timeline_10k.txt 2>&1
PM_PROF_NO_BUILD=1 timeout 200 python tools/stream_trace.py churn > gpurun_out/r05/r05_stream_timeline_churn.txt 2>&1
PM_TRACE_HOST=1 timeout 120 python tools/host_trace.py 1 2>&1 | tail -16 > gpurun_out/r05/r05_host_marks_match.txt
PM_TRACE_HOST=1 timeout 120 python tools/churn_probe.py 8 2>&1 | tail -40 > gpurun_out/r05/r05_host_marks_churn.txt
