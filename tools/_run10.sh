export PM_PROF_NO_BUILD=1
mkdir -p gpurun_out/r04j
export PM_PROF_LIB=$PWD/protocol_amd/libpm_var_new.so
for i in 1 2; do python tools/stream_prof.py 100000 10000 | sed -n '1,2p;5,6p'; done
for i in 1 2; do python tools/stream_prof.py 1000000 100000 | sed -n '1,2p;5,6p'; done
PM_STREAM_WGS=180 python tools/stream_prof.py 1000000 100000 | sed -n '1p;5p'
python tools/stream_trace.py 100000 10000 --dump gpurun_out/r04j/trace_10k.txt > gpurun_out/r04j/tl_10k.txt 2>&1; cat gpurun_out/r04j/tl_10k.txt | head -34
python tools/stream_trace.py churn --dump gpurun_out/r04j/trace_churn.txt > gpurun_out/r04j/tl_churn.txt 2>&1; head -16 gpurun_out/r04j/tl_churn.txt
unset PM_PROF_LIB
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
