export PM_PROF_NO_BUILD=1
mkdir -p gpurun_out/r04k
export PM_PROF_LIB=$PWD/protocol_amd/libpm_var_new.so
python tools/stream_trace.py 1000000 100000 --dump gpurun_out/r04k/trace_100k.txt > gpurun_out/r04k/tl_100k.txt 2>&1; head -40 gpurun_out/r04k/tl_100k.txt
