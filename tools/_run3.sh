mkdir -p gpurun_out/r04c; export PM_PROF_NO_BUILD=1
( echo "== default"; timeout 120 python tools/stream_prof.py 100000 10000; timeout 200 python tools/stream_prof.py 1000000 100000
  for d in 2 16; do echo "== LA_DIV=$d"; PM_STREAM_LA_DIV=$d timeout 120 python tools/stream_prof.py 100000 10000 | head -3; PM_STREAM_LA_DIV=$d timeout 200 python tools/stream_prof.py 1000000 100000 | head -3; done
  for l in 64 256; do echo "== LA=$l"; PM_STREAM_LA=$l timeout 120 python tools/stream_prof.py 100000 10000 | head -3; PM_STREAM_LA=$l timeout 200 python tools/stream_prof.py 1000000 100000 | head -3; done
  for w in 8 64 250; do echo "== WGS=$w"; PM_STREAM_WGS=$w timeout 120 python tools/stream_prof.py 100000 10000 | head -3; PM_STREAM_WGS=$w timeout 200 python tools/stream_prof.py 1000000 100000 | head -3; done
) > gpurun_out/r04c/prof.txt 2>&1
cat gpurun_out/r04c/prof.txt
