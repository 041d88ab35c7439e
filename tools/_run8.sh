export PM_PROF_NO_BUILD=1
mkdir -p gpurun_out/r04h
python tools/stream_prof.py 100000 10000 > gpurun_out/r04h/anat_10k.txt 2>&1; cat gpurun_out/r04h/anat_10k.txt
python tools/stream_prof.py 1000000 100000 > gpurun_out/r04h/anat_100k.txt 2>&1; cat gpurun_out/r04h/anat_100k.txt
python tools/stream_trace.py 100000 10000 --dump gpurun_out/r04h/trace_10k.txt > gpurun_out/r04h/tl_10k.txt 2>&1; head -40 gpurun_out/r04h/tl_10k.txt
python tools/stream_trace.py churn --dump gpurun_out/r04h/trace_churn.txt > gpurun_out/r04h/tl_churn.txt 2>&1; head -30 gpurun_out/r04h/tl_churn.txt
bash tools/_bench.sh
