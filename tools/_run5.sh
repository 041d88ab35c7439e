mkdir -p gpurun_out/r04e; export PM_PROF_NO_BUILD=1
( for f in 128 32 8; do echo "== PRUNE_FACTOR=$f"; PM_PRUNE_FACTOR=$f timeout 120 python tools/stream_prof.py 100000 10000 | sed -n '1,2p;5,9p'; PM_PRUNE_FACTOR=$f timeout 200 python tools/stream_prof.py 1000000 100000 | sed -n '1,2p;5,9p'; done
  echo "== PRUNE_MODE=2 (always walk)"; PM_PRUNE_MODE=2 timeout 120 python tools/stream_prof.py 100000 10000 | sed -n '1,2p;5,9p'; PM_PRUNE_MODE=2 timeout 200 python tools/stream_prof.py 1000000 100000 | sed -n '1,2p;5,9p'
) > gpurun_out/r04e/prof.txt 2>&1
cat gpurun_out/r04e/prof.txt
