export PM_PROF_NO_BUILD=1
mkdir -p gpurun_out/r04k
export PM_PROF_LIB=$PWD/protocol_amd/libpm_var_new.so
python tools/stream_trace.py 100000 10000 --dump gpurun_out/r04k/trace_10k.txt > gpurun_out/r04k/tl_10k.txt 2>&1; grep "slots\|config\| run " gpurun_out/r04k/trace_10k.txt | head -60
