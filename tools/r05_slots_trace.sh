cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r05t1
L=protocol_amd/variants/libpm_engine_rowrec.so
PM_PROF_LIB=$L PM_PROF_NO_BUILD=1 timeout 200 python tools/stream_trace.py 100000 10000 --dump gpurun_out/r05t1/events_10k.txt > gpurun_out/r05t1/timeline_10k.txt 2>&1
grep "slot" gpurun_out/r05t1/events_10k.txt | head -60
