mkdir -p gpurun_out/r04f; export PM_PROF_NO_BUILD=1
timeout 200 python tools/stream_trace.py 100000 10000 --dump gpurun_out/r04f/trace10k.txt > gpurun_out/r04f/t10k.txt 2>&1
timeout 300 python tools/stream_trace.py 1000000 100000 --dump gpurun_out/r04f/trace100k.txt > gpurun_out/r04f/t100k.txt 2>&1
cat gpurun_out/r04f/t10k.txt; cat gpurun_out/r04f/t100k.txt
